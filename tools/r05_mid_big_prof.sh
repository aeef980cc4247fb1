#!/bin/bash
# throughput of k_round_mid at LARGE sizes (n = 24 with the middle-round kernel taking every round up to 2^20 pair indices)
set -u
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mid_big
rm -rf $OUT; mkdir -p $OUT
export ZK_PIPE_MID_MAX_PAIRS=1048576
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/prof_sumcheck.py 24 2 > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; exit 1; }
python3 - <<P
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows[-60:]:
    n = r["Kernel_Name"].split("(")[0].replace("void zk::", "")
    print(f'{n:50s} grid {r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"):>8} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us')
P
rm -rf $OUT/trace
tail -1 $OUT/run.log
