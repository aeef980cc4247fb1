#!/usr/bin/env python3
"""Child of bench.py's traffic leg: a few launches of the headline kernel (k_fold_msb, 2^n -> 2^(n-1), BN254 Fr) and nothing
else, so that `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, MI355X_MICROARCH.md "HBM") can count its
HBM traffic.  Usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d DIR --output-format csv -- python3 tools/pmc_fold.py [n] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
t = zk_amd.MultiLinearPolynomial.random(ctx, n, 0x5EED0000 + 24, 0)
o = zk_amd.MultiLinearPolynomial.alloc(ctx, n - 1)
tr = zk_amd.Transcript()
tr.append(b"zk_amd bench challenge")
r = tr.sample_field_element(zk_amd.BN254_FR)
for _ in range(reps):
    t.fold_into(r, o)
ctx.synchronize()
print("pmc_fold done", n, reps)
