#!/bin/bash
# kernel timeline of the one-rank sharded prover (RCCL): what an exchanging round consists of, with the gaps between its launches
set -u
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_shard
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/prof_shard.py 13 3 > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; exit 1; }
python3 - <<P
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
# last proof: from the last k_store_sponge on
idx = max(i for i, r in enumerate(rows) if "k_store_sponge" in r["Kernel_Name"])
prev_end = None
for r in rows[idx:idx + 70]:
    n = r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", "")[:60]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print(f"{n:62s} dur {(en - st) / 1e3:7.1f} us   gap before {gap:6.1f} us")
    prev_end = en
P
rm -rf $OUT/trace
tail -1 $OUT/run.log
