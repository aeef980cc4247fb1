#!/bin/bash
# the big rounds on the LDS-DMA kernels (ZK_ROUND_GLDS=1: k_round0_glds<0/1>, k_round_fused_glds<3,0> / <2,1>) vs k_round0_dot29 /
# k_round_kd (=0): forced-path parity at small sizes, then the wall-clock A/B (each arm in its own process, interleaved)
set -u
mkdir -p gpurun_out
L=gpurun_out/r06_glds_ab.log
: > $L
G="ZK_ROUND_GLDS_MIN_PAIRS=64 ZK_ROUND_GLDS_NT_MIN_PAIRS=256 ZK_LEAD_MIN_PAIRS=1 ZK_SKIP1_MIN_PAIRS=1 ZK_QUAD_MAX_PAIRS=0"
env $G ZK_PIPE_MAX_PAIRS=0 ZK_CHECK_SIZES=7,8,9,11,13,15 timeout -k 10 400 python tests/skip1_check.py 2>&1 | tail -3 | tee -a $L || exit 1
env $G ZK_CHECK_SIZES=11,12,13,15 timeout -k 10 400 python tests/skip1_check.py 2>&1 | tail -3 | tee -a $L || exit 1
env ZK_CHECK_SIZES=18,19,20 ZK_CHECK_FIELDS=2 timeout -k 10 600 python tests/skip1_check.py 2>&1 | tail -3 | tee -a $L || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_gkr.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -2 | tee -a $L
timeout -k 10 900 python tools/r06_glds_fused_ab.py 2>&1 | tee -a $L
timeout -k 10 900 python tools/ab_prover.py ZK_ROUND_GLDS=0 ZK_ROUND_GLDS=1 2>&1 | tee -a $L
