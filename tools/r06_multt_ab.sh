#!/bin/bash
# A/B: the inner table x table products of the K = 3 round kernels on the carry-free core (fe_mul_tt) instead of the saturated one (fe_mul):
# ab_tmp/libzk_cur.so (shipped) vs ab_tmp/libzk_multt.so (-DZK_KD_INNER_MUL=fe_mul_tt), three alternating passes on one box
set -u
for r in 1 2 3; do
  for v in cur multt; do
    export ZK_AMD_LIB=$PWD/ab_tmp/libzk_$v.so
    echo "== $v: $(python3 tools/prof_k3.py 20 2>/dev/null | tail -1)"
    echo "== $v: $(python3 tools/prof_batch.py 20 5 2>/dev/null | grep 'k=3 D=3 n=20 B=8')"
  done
done
