#!/bin/bash
# Round-4 evidence run: (1) the bench command under rocprofv3 --kernel-trace --stats (k_fold_msb's average must agree with the
# bench's own HIP events), (2) tools/gpu_profile_all.sh (prover n = 24 / 20, k = 3, evaluate, NTT, GKR, fold).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-extra --no-pmc --no-cpu-baseline --no-parity-gate > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { echo "bench under rocprof failed"; tail -5 $OUT/bench_under_rocprof.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_fold_kernel_stats.csv
cat $OUT/bench_fold_kernel_stats.csv | head -5
bash tools/gpu_profile_all.sh
python3 tools/summarize_prof_all.py gpurun_out/prof_all r04 > /dev/null && cp profiles/r04_prover_ntt_gkr_kernel_stats.md gpurun_out/
