#!/bin/bash
# Same-box A/B of library builds on the GKR driver only (depth 8, width 2^20): bash tools/ab_gkr_libs.sh base new
set -u
for r in 1 2 3; do
  for v in "$@"; do
    cp ab_tmp/libzk_$v.so zk_amd/libzk_amd.so || exit 1
    echo "== $v: $(python3 tools/prof_gkr.py 20 8 | grep -E 'prove|verify' | tr '\n' ' ')"
  done
done
