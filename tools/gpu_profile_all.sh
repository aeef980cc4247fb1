#!/bin/bash
# kernel-trace + stats for the prover (n=24 and n=20), the NTT and the GKR driver: summaries for profiles/ (tools/summarize_prof_all.py)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_all
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/ntt_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] if "GRAFT_REPO_ROOT" in os.environ else "/root/repo")
import zk_amd
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
x = zk_amd.MultiLinearPolynomial.random(ctx, 24, 5, 0); y = zk_amd.MultiLinearPolynomial.alloc(ctx, 24)
print("ntt ms", zk_amd.bench_ntt(ctx, x, y, False, 5), "intt ms", zk_amd.bench_ntt(ctx, x, y, True, 5))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sumcheck_n24 -- python3 tools/prof_sumcheck.py 24 5 > $OUT/sumcheck_n24.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sumcheck_n20 -- python3 tools/prof_sumcheck.py 20 5 > $OUT/sumcheck_n20.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ntt -- python3 /tmp/ntt_run.py > $OUT/ntt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gkr -- python3 tools/prof_gkr.py 20 8 > $OUT/gkr.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sumcheck_k3_n20 -- python3 tools/prof_k3.py 20 > $OUT/sumcheck_k3_n20.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/batch8_k3_n20 -- python3 tools/prof_batch.py 20 3 > $OUT/batch8_k3_n20.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/evaluate -- python3 tools/prof_evaluate.py > $OUT/evaluate.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fold -- python3 tools/pmc_fold.py 24 200 > $OUT/fold.log 2>&1 || exit 1
grep -h "^n \|^k3 \|ntt ms\|prove ms\|verify ms\|^evaluate\|^k=" $OUT/*.log
