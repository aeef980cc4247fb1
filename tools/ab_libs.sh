#!/bin/bash
# A/B of two builds on the same box: alternate, 3 rounds
for r in 1 2 3; do
  for v in new2 new3; do
    cp ab_tmp/libzk_$v.so zk_amd/libzk_amd.so
    echo "== $v: $(python3 tools/prof_sumcheck.py 24 8 | tail -1)"
    echo "== $v: $(python3 tools/prof_sumcheck.py 20 8 | tail -1)"
    echo "== $v: $(python3 tools/prof_k3.py 20 | tail -1)"
    echo "== $v: $(python3 tools/prof_gkr.py 20 8 | grep prove | tail -1)"
  done
done
