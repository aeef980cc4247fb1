#!/bin/bash
# kernel stats of the n = 20 prover with the mid-size pipelined rounds on (ZK_PIPE_MID_MAX_PAIRS=65536) and off (0)
set -u
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for arm in 0 65536; do
  OUT=$R/gpurun_out/prof_mid_$arm
  rm -rf $OUT; mkdir -p $OUT
  export ZK_PIPE_MID_MAX_PAIRS=$arm
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_sumcheck.py 20 5 > $OUT/run.log 2>&1 || { echo "failed arm $arm"; tail -5 $OUT/run.log; exit 1; }
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_mid_kernel_stats_$arm.csv
  rm -rf $OUT/trace
  echo "== arm $arm"; cat $OUT/run.log | tail -1; head -12 $R/gpurun_out/r05_mid_kernel_stats_$arm.csv
done
