#!/bin/bash
# Same-box A/B of library builds kept under ab_tmp/ (libzk_<name>.so), selected with ZK_AMD_LIB (announced on stderr by zk_amd/_lib.py):
#   bash tools/r06_ab_libs.sh head batch     -> three alternating rounds of the n = 24 / n = 20 / k = 3 provers and the GKR driver
set -u
for r in 1 2 3; do
  for v in "$@"; do
    export ZK_AMD_LIB=$PWD/ab_tmp/libzk_$v.so
    echo "== $v: $(python3 tools/prof_sumcheck.py 24 8 2>/dev/null | tail -1)"
    echo "== $v: $(python3 tools/prof_sumcheck.py 20 8 2>/dev/null | tail -1)"
    echo "== $v: $(python3 tools/prof_k3.py 20 2>/dev/null | tail -1)"
    echo "== $v: $(python3 tools/prof_gkr.py 20 8 2>/dev/null | grep prove | tail -1)"
  done
done
