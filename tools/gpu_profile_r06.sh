#!/bin/bash
# Round-6 evidence run, part 2 (part 1 is `pytest -m gpu`): the bench line, the bench under rocprofv3 (k_fold_msb's average must agree
# with the bench's own HIP events), tools/gpu_profile_all.sh, and the sharded leg through RCCL at one rank.  Everything lands in
# gpurun_out/; tools/gen_design_tables.py reads the copies under profiles/r06_*.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err || { echo "bench failed"; tail -5 gpurun_out/r06_bench_final.err; exit 1; }
OUT=$PWD/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-extra --no-pmc --no-cpu-baseline --no-parity-gate > gpurun_out/r06_bench_fold_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { echo "bench under rocprof failed"; tail -5 $OUT/bench_under_rocprof.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r06_bench_fold_kernel_stats.csv
rm -rf $OUT/trace
head -3 gpurun_out/r06_bench_fold_kernel_stats.csv
bash tools/gpu_profile_all.sh || { echo "profile_all failed"; exit 1; }
python3 tools/summarize_prof_all.py gpurun_out/prof_all r06 > /dev/null && cp profiles/r06_prover_ntt_gkr_kernel_stats.md gpurun_out/
find gpurun_out/prof_all -name "*.csv" -size +2M -delete
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-pmc > gpurun_out/r06_world1_rccl_final.json 2> gpurun_out/r06_world1_rccl_final.err || { echo "world-1 failed"; tail -5 gpurun_out/r06_world1_rccl_final.err; exit 1; }
tail -c 400 gpurun_out/r06_bench_final.json
# round 6 extras: counters of the evaluate kernels (item 4), the serial sharded loop with an injected all-reduce latency (what W = 8 would pay per round)
bash tools/pmc_evaluate.sh > gpurun_out/r06_evaluate_kernel_stats_and_pmc.log 2>&1 || echo "pmc_evaluate failed"
for A in 0 10 15 25; do echo "A=$A us: $(ZK_SHARD_FAKE_ALLREDUCE_US=$A python3 tools/prof_shard.py 13 9 21 2>/dev/null | tail -1)"; done > gpurun_out/r06_shard_serial_fake_latency.log 2>&1
cat gpurun_out/r06_shard_serial_fake_latency.log
