#!/bin/bash
# kernel timeline of one prove_partial at n = $1 (k = 2, D = 2)
set -u
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
N=${1:-12}
OUT=$R/gpurun_out/prof_trace_n
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/prof_sumcheck.py $N 3 > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; exit 1; }
python3 - <<P
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_store_sponge" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + 40]:
    n = r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", "")[:56]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{n:58s} start {(st - t0) / 1e3:7.1f}  dur {(en - st) / 1e3:6.1f} us")
P
rm -rf $OUT/trace
tail -1 $OUT/run.log
