#!/bin/bash
# rocprofv3 kernel stats of the GKR driver (depth 8, width 2^20) for each library variant in ab_tmp/: tools/prof_gkr_stats.sh ep pair
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_gkr
rm -rf $OUT; mkdir -p $OUT
for v in "$@"; do
  cp ab_tmp/libzk_$v.so zk_amd/libzk_amd.so || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 tools/prof_gkr.py 20 8 > $OUT/$v.log 2>&1
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep -E 'prove' $OUT/$v.log | tail -1)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} tot {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
done
