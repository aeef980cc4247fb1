#!/usr/bin/env python3
"""tools/prof_zeta.py [n_vars] [reps] [n_terms] -- to_evaluation_form (coefficient_form.rs:340-347) timed on the device and
ready for rocprofv3 (kernel trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE): prints the median wall-clock of the device-resident call
(term list of n_terms random keys; the host-side merge of the term list is inside the call and is a few hundred microseconds at
1k terms).  ZK_ZETA_GLOBAL=1 selects round 4's global passes for the A/B."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zk_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n_terms = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 10
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
rng = np.random.default_rng(0xC0EF)
keys = rng.integers(0, 1 << n, n_terms, dtype=np.uint64)
coeffs = zk_amd.fe_from_ints(field, [int(x) for x in rng.integers(1, 1 << 62, n_terms)])
cf = zk_amd.CoeffMultilinearPolynomial.new_with_coefficient(field, n, {int(k): c for k, c in zip(keys, coeffs)})
cf.to_evaluation_form(ctx).free()
ts = []
for _ in range(reps):
    ctx.synchronize()
    t1 = time.perf_counter()
    t = cf.to_evaluation_form(ctx)
    ctx.synchronize()
    ts.append(time.perf_counter() - t1)
    t.free()
ts.sort()
form = "global passes (round 4)" if os.environ.get("ZK_ZETA_GLOBAL") == "1" else "LDS-tiled passes"
print(f"to_evaluation_form n={n} terms={n_terms} {form}: median {ts[len(ts) // 2] * 1e6:.1f} us  min {ts[0] * 1e6:.1f} us")
