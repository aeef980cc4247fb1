#!/bin/bash
# kernel stats of the k = 3 prover (n = 20, n = 22) and the GKR driver with the LDS-DMA round kernels off / on
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_glds_k3
rm -rf $OUT; mkdir -p $OUT
for arm in 0 1; do
  for n in 20 22; do
    ( export ZK_ROUND_GLDS=$arm; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/arm${arm}_n$n -- python3 tools/prof_k3.py $n > $OUT/arm${arm}_n$n.log 2>&1 )
    f=$(find $OUT/arm${arm}_n$n -name "*kernel_stats.csv" | head -1)
    echo "== ZK_ROUND_GLDS=$arm k=3 n=$n: $(grep '^k3 ' $OUT/arm${arm}_n$n.log)"
    python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
  done
  ( export ZK_ROUND_GLDS=$arm; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/arm${arm}_gkr -- python3 tools/prof_gkr.py 20 8 > $OUT/arm${arm}_gkr.log 2>&1 )
  f=$(find $OUT/arm${arm}_gkr -name "*kernel_stats.csv" | head -1)
  echo "== ZK_ROUND_GLDS=$arm gkr"
  grep "k_round0\|k_round_kd\|fused_glds" "$f" | awk -F, '{printf "  %s calls %s avg %.1f\n", substr($1,1,70), $2, $4/1000}' | head -8
done
