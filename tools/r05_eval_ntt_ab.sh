#!/bin/bash
# (1) evaluate parity incl. the new L = 14 / 15 coverage, (2) eq-table depth A/B (two library builds, same box), (3) NTT inter-pass
# table A/B (time + HBM traffic per pass)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "evaluate" > gpurun_out/r05_eval_tests.log 2>&1 || { tail -30 gpurun_out/r05_eval_tests.log; exit 1; }
tail -2 gpurun_out/r05_eval_tests.log
L=gpurun_out/r05_eval_eq_depth_ab.log
echo "# evaluate, whole call (zk_bench_evaluate, std::chrono inside the library), BN254 / BLS12-381: eq tables built as a chain of three multiplications (eqchain, round 4) vs two levels (eqdepth2); same box, alternating" > $L
for r in 1 2; do for v in eqchain eqdepth2; do
  echo "== $v" >> $L
  ZK_AMD_LIB=$PWD/ab_tmp/libzk_$v.so ZK_AB_CHILD=1 timeout -k 10 300 python tools/ab_evaluate.py >> $L 2>&1 || { tail -5 $L; exit 1; }
done; done
grep -E "==|n=18|n=21|n=24" $L | grep -E "==|bn254"
N=gpurun_out/r05_ntt_table_ab.log
echo "# 2^20 / 2^22 / 2^24-point NTT: full inter-pass twiddle tables up to 2^24 entries (shipped) vs up to 2^16 (pass 0 composes) vs none" > $N
timeout -k 10 600 python tools/ab_ntt.py ZK_NTT_FULL_TABLE_MAX_LOG=24 ZK_NTT_FULL_TABLE_MAX_LOG=16 ZK_NTT_FULL_TABLE_MAX_LOG=0 >> $N 2>&1
cat $N
for arm in 24 16; do
  export ZK_NTT_FULL_TABLE_MAX_LOG=$arm
  bash tools/pmc_ntt.sh > gpurun_out/r05_ntt_pmc_$arm.log 2>&1
  echo "== PMC, ZK_NTT_FULL_TABLE_MAX_LOG=$arm" >> $N
  grep -E "k_ntt_pass|FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU " gpurun_out/r05_ntt_pmc_$arm.log >> $N
done
tail -24 $N
