#!/bin/bash
# Same-box A/B of library builds: put the variants at ab_tmp/libzk_<name>.so (ab_tmp/ travels with gpurun), then on the GPU box:
#   bash tools/ab_libs.sh old new        -> three alternating rounds of the n = 24 / n = 20 / k = 3 provers and the GKR driver
# (box-to-box spread is ~3 %, so variants are only comparable inside one call).  Restores nothing: rebuild afterwards.
set -u
for r in 1 2 3; do
  for v in "$@"; do
    cp ab_tmp/libzk_$v.so zk_amd/libzk_amd.so || exit 1
    echo "== $v: $(python3 tools/prof_sumcheck.py 24 8 | tail -1)"
    echo "== $v: $(python3 tools/prof_sumcheck.py 20 8 | tail -1)"
    echo "== $v: $(python3 tools/prof_k3.py 20 | tail -1)"
    echo "== $v: $(python3 tools/prof_gkr.py 20 8 | grep prove | tail -1)"
  done
done
