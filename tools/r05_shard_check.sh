#!/bin/bash
# sharded prover with SKIP1 / LEAD round kernels: parity in child processes, then the world-1 RCCL bench (A/B on ZK_SHARD_SKIP1)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_shard.py -x -q -k "derive_behind or single_rank or match_unsharded" > gpurun_out/r05_shard_tests.log 2>&1 || { tail -30 gpurun_out/r05_shard_tests.log; exit 1; }
tail -2 gpurun_out/r05_shard_tests.log
for arm in 1 0; do
  ZK_SHARD_SKIP1=$arm timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$arm bench.py --gpus 1 --no-cpu-baseline --no-pmc > gpurun_out/r05_world1_skip$arm.json 2> gpurun_out/r05_world1_skip$arm.err || { tail -5 gpurun_out/r05_world1_skip$arm.err; exit 1; }
  python - <<P
import json
d=json.loads([l for l in open('gpurun_out/r05_world1_skip$arm.json') if l.startswith('{')][-1])
ex=d['extra']
print('ZK_SHARD_SKIP1=$arm', {k:v for k,v in ex.items() if k.startswith('sharded_sumcheck_ms')}, d['sumcheck_prover_wall_clock_ms'])
P
done
