#!/bin/bash
# the SKIP1 claim evaluated by the round kernel's extra workgroup (ZK_CLAIM_IN_ROUND=1, shipped) vs by the tails (=0): parity, then A/B
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_gpu_gkr.py -x -q -k "skip1 or derive_behind or prover or config2 or config3 or gkr_depth8 or terms or sumcheck" > gpurun_out/r05_claim_tests.log 2>&1 || { tail -30 gpurun_out/r05_claim_tests.log; exit 1; }
tail -2 gpurun_out/r05_claim_tests.log
ZK_CLAIM_IN_ROUND=0 ZK_SKIP1_MIN_PAIRS=1 ZK_LEAD_MIN_PAIRS=1 ZK_QUAD_MAX_PAIRS=0 ZK_CHECK_SIZES=3,7,11,13 timeout -k 10 300 python tests/skip1_check.py | tail -1
timeout -k 10 600 python tools/ab_prover.py ZK_CLAIM_IN_ROUND=0 ZK_CLAIM_IN_ROUND=1 > gpurun_out/r05_claim_in_round_ab.log 2>&1
cat gpurun_out/r05_claim_in_round_ab.log
for arm in 0 1; do
  ZK_CLAIM_IN_ROUND=$arm timeout -k 10 200 python tools/prof_shard.py 16 7 | tail -1
  ZK_CLAIM_IN_ROUND=$arm timeout -k 10 200 python tools/prof_shard.py 13 7 | tail -1
done
