#!/bin/bash
# One GPU-box session: parity tests, then (unless the tests hung) the bench.  Logs go to gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 ${TEST_TIMEOUT:-700} python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/gpu_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/gpu_tests.log
tail -n 25 gpurun_out/gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests hung: not starting the bench"; exit $rc; fi
timeout -k 10 ${BENCH_TIMEOUT:-400} python bench.py --steps ${STEPS:-20} --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err
brc=$?
echo "bench rc=$brc"; tail -n 5 gpurun_out/bench.err; cat gpurun_out/bench.log
exit $(( rc != 0 ? rc : brc ))
