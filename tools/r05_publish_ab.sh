#!/bin/bash
# the proof block published by the pipelined finisher itself (ZK_PUBLISH_IN_FINISHER=1, shipped) vs by k_publish_host (=0): parity, A/B
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gkr.py tests/test_gpu_concurrency.py tests/test_gpu_cpp_host.py -x -q -k "not skip1 and not beyond and not streaming and not evaluate" > gpurun_out/r05_publish_tests.log 2>&1 || { tail -30 gpurun_out/r05_publish_tests.log; exit 1; }
tail -2 gpurun_out/r05_publish_tests.log
timeout -k 10 600 python tools/ab_prover.py ZK_PUBLISH_IN_FINISHER=0 ZK_PUBLISH_IN_FINISHER=1 > gpurun_out/r05_publish_in_finisher_ab.log 2>&1
cat gpurun_out/r05_publish_in_finisher_ab.log
for rep in 1 2; do for arm in 0 1; do ZK_PUBLISH_IN_FINISHER=$arm python tools/prof_sumcheck.py 12 60 | python -c "
import sys,ast
l=sys.stdin.read().strip().split('ms ',1)[1]; v=sorted(ast.literal_eval(l)); print('n=12 arm $arm median %.4f min %.4f ms' % (v[len(v)//2], v[0]))" | tee -a gpurun_out/r05_publish_in_finisher_ab.log; done; done
