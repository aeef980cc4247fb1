#!/bin/bash
# rocprofv3 kernel stats of the k = 3, D = 3, n = 20 prover (tools/prof_k3.py)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_k3
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 tools/prof_k3.py 20 > $OUT/run.log 2>&1
f=$(find $OUT/run -name "*kernel_stats.csv" | head -1)
grep "k3 n" $OUT/run.log
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} tot {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
