#!/bin/bash
# kernel stats of the n = 20 / n = 24 prover under two environment arms: tools/prof_ab.sh "ZK_LANE_ACC=0" "ZK_LANE_ACC=1"
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_ab
rm -rf $OUT; mkdir -p $OUT
i=0
for arm in "$@"; do
  i=$((i+1))
  for n in 20 24; do
    ( export $arm; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/arm${i}_n$n -- python3 tools/prof_sumcheck.py $n 10 > $OUT/arm${i}_n$n.log 2>&1 )
    f=$(find $OUT/arm${i}_n$n -name "*kernel_stats.csv" | head -1)
    echo "== $arm n=$n: $(grep '^n ' $OUT/arm${i}_n$n.log)"
    python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} tot {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
  done
done
