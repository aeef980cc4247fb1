#!/bin/bash
# kernel timelines (rocprofv3 --kernel-trace): (1) ONE zk_sumcheck_prove_batch of eight (k = 3, n = 20) proofs -- 22 launches, grid (x, 8);
# (2) the same eight proofs on eight contexts / host threads (stream concurrency): which kernels of different queues overlap
set -u
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_batch_tl
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/one_batch.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, zk_amd
from zk_amd import MultiLinearPolynomial as MLE
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
layers = [zk_amd.ProductPoly.new([MLE.random(ctx, 20, 0x6000 + 16 * l + f, 0) for f in range(3)]) for l in range(8)]
s = np.stack([zk_amd.fe_from_int(0, 7)] * 8)
for _ in range(4): zk_amd.SumcheckProver(3).prove_partial_batch(layers, s)
PY
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/batch -- python3 /tmp/one_batch.py > $OUT/batch.log 2>&1 || { echo failed; tail -5 $OUT/batch.log; exit 1; }
python3 - <<P
import csv, glob
f = glob.glob("$OUT/batch/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_store_sponge_b" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev = None
print("== (1) one zk_sumcheck_prove_batch, eight (k = 3, D = 3, n = 20) proofs: every launch of the call")
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", "")[:52]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{n:54s} grid {r.get('Grid_Size','?'):>9s}  start {(st - t0) / 1e3:8.1f}  dur {(en - st) / 1e3:7.1f} us  gap before {((st - prev) / 1e3) if prev else 0.0:5.1f}")
    prev = en
print(f"span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
P
cat > /tmp/threads8.py <<PY
import sys, threading; sys.path.insert(0, "$R")
import numpy as np, zk_amd
from zk_amd import MultiLinearPolynomial as MLE
ctxs = [zk_amd.Context(zk_amd.BN254_FR, 0) for _ in range(8)]
polys = [zk_amd.ProductPoly.new([MLE.random(c, 20, 0x6000 + 16 * l + f, 0) for f in range(3)]) for l, c in enumerate(ctxs)]
claimed = zk_amd.fe_from_int(0, 7)
for pp in polys: zk_amd.SumcheckProver(3).prove_partial(pp, claimed)
for rep in range(3):
    bar = threading.Barrier(8)
    def th(i):
        bar.wait(); zk_amd.SumcheckProver(3).prove_partial(polys[i], claimed)
    ts = [threading.Thread(target=th, args=(i,)) for i in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
PY
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/threads -- python3 /tmp/threads8.py > $OUT/threads.log 2>&1 || { echo failed; tail -5 $OUT/threads.log; exit 1; }
python3 - <<P
import csv, glob
f = glob.glob("$OUT/threads/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last repetition: the last 8 k_store_sponge launches mark its start
st_idx = [i for i, r in enumerate(rows) if "k_store_sponge" in r["Kernel_Name"]][-8]
sel = rows[st_idx:]
t0 = int(sel[0]["Start_Timestamp"]); end = max(int(r["End_Timestamp"]) for r in sel)
queues = sorted({r["Queue_Id"] for r in sel})
print("== (2) eight contexts on eight host threads, one (k = 3, n = 20) proof each: span %.1f us, %d kernels on queues %s" % ((end - t0) / 1e3, len(sel), ",".join(queues)))
# how much of the span has >= 2 kernels (of different queues) running at once
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = 0; last = t0; busy = [0, 0, 0]
for t, d in ev:
    busy[min(cur, 2)] += t - last; last = t; cur += d
print("time with 0 / 1 / >= 2 kernels in flight: %.1f / %.1f / %.1f us" % tuple(b / 1e3 for b in busy))
big = [r for r in sel if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 30000]
print("kernels longer than 30 us (they fill the machine; another queue's launches wait behind them):")
for r in big[:24]:
    n = r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", "")[:44]
    print(f"  queue {r['Queue_Id']:>2s} {n:46s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} us")
P
rm -rf $OUT/batch $OUT/threads
