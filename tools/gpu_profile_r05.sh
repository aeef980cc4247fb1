#!/bin/bash
# Round-5 evidence run, part 2 (part 1 is `pytest -m gpu`): the bench line, the bench under rocprofv3 (k_fold_msb's average must agree
# with the bench's own HIP events), tools/gpu_profile_all.sh, and the sharded leg through RCCL at one rank.  Everything lands in
# gpurun_out/; tools/gen_design_tables.py reads the copies under profiles/r05_*.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err || { echo "bench failed"; tail -5 gpurun_out/r05_bench_final.err; exit 1; }
OUT=$PWD/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-extra --no-pmc --no-cpu-baseline --no-parity-gate > gpurun_out/r05_bench_fold_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { echo "bench under rocprof failed"; tail -5 $OUT/bench_under_rocprof.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r05_bench_fold_kernel_stats.csv
rm -rf $OUT/trace
head -3 gpurun_out/r05_bench_fold_kernel_stats.csv
bash tools/gpu_profile_all.sh || { echo "profile_all failed"; exit 1; }
python3 tools/summarize_prof_all.py gpurun_out/prof_all r05 > /dev/null && cp profiles/r05_prover_ntt_gkr_kernel_stats.md gpurun_out/
find gpurun_out/prof_all -name "*.csv" -size +2M -delete
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-pmc > gpurun_out/r05_world1_rccl_final.json 2> gpurun_out/r05_world1_rccl_final.err || { echo "world-1 failed"; tail -5 gpurun_out/r05_world1_rccl_final.err; exit 1; }
tail -c 400 gpurun_out/r05_bench_final.json
