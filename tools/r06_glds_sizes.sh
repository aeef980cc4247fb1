#!/bin/bash
# per-round kernel durations of the big rounds, n = 18..23, three shapes, LDS-DMA kernels forced on at every size vs off
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_glds_sizes
rm -rf $OUT; mkdir -p $OUT
( export ZK_ROUND_GLDS=0; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/off -- python3 tools/r06_glds_sizes.py > $OUT/off.log 2>&1 ) || exit 1
( export ZK_ROUND_GLDS=1 ZK_ROUND_GLDS_MIN_PAIRS=65536; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/on -- python3 tools/r06_glds_sizes.py > $OUT/on.log 2>&1 ) || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
def cells(arm):
    f = glob.glob(f"{out}/{arm}/*/*kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    big = [r for r in rows if any(s in r["Kernel_Name"] for s in ("k_round0", "k_round_kd", "fused_glds"))]
    # a proof starts at a sums-only kernel (round 0): k_round0_* or k_round_kd<..., false, ...>
    proofs, cur = [], None
    for r in big:
        nm = r["Kernel_Name"]
        is0 = "k_round0" in nm or ", false, " in nm.split("(")[0].replace("<", ", ", 1)[:60] and "k_round_kd" in nm and nm.split("<")[1].split(",")[2].strip() == "false"
        if is0:
            cur = []
            proofs.append(cur)
        if cur is not None:
            cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return proofs
off, on = cells("off"), cells("on")
shapes = ("k=2", "k=3", "2+1")
i = 0
for n in range(18, 24):
    for s in shapes:
        a = [min(p[j] for p in off[i:i + 3] if len(p) > j) for j in range(min(4, min(len(p) for p in off[i:i + 3])))]
        b = [min(p[j] for p in on[i:i + 3] if len(p) > j) for j in range(min(4, min(len(p) for p in on[i:i + 3])))]
        print(f"n={n} {s}: off " + " ".join(f"{x:7.1f}" for x in a) + "   | on " + " ".join(f"{x:7.1f}" for x in b))
        i += 3
PY
