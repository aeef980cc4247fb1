#!/bin/bash
# rocprofv3 passes for the bench command: kernel trace + stats, then PMC passes (separate runs, counters only).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${EXTRA:---no-extra}"
cd $PWD
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python3 $ARGS > $OUT/pmc_$ctr.log 2>&1 || { echo pmc $ctr failed; tail -5 $OUT/pmc_$ctr.log; exit 1; }
done
find $OUT -name "*.csv" | head -20
