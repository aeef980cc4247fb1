#!/bin/bash
# same-box A/B of library builds on the NTT: ab_tmp/libzk_<name>.so, three alternating rounds (tools/ab_ntt.py prints fwd / inv ms)
set -u
cp zk_amd/libzk_amd.so /tmp/libzk_keep.so
for r in 1 2 3; do
  for v in "$@"; do
    cp ab_tmp/libzk_$v.so zk_amd/libzk_amd.so || exit 1
    echo "== $v: $(python3 tools/ab_ntt.py A=1 | head -1)"
  done
done
cp /tmp/libzk_keep.so zk_amd/libzk_amd.so
