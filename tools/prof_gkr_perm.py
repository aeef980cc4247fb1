"""GKR phase kernels vs wiring: random inputs (row lengths Poisson(1): divergent per-row loops) against permutation wiring (every
row exactly one entry, gathers still random).  Run under rocprofv3 --kernel-trace --stats to read k_gkr_phase1/2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zk_amd
from zk_amd import gkr
w, depth = 20, 4
mode = sys.argv[1] if len(sys.argv) > 1 else "perm"
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
rng = np.random.default_rng(0x6B72)
circ = gkr.Circuit(ctx)
for _ in range(depth):
    if mode == "perm":
        left, right = rng.permutation(1 << w).astype(np.uint32), rng.permutation(1 << w).astype(np.uint32)
    else:
        left, right = rng.integers(0, 1 << w, 1 << w, dtype=np.uint32), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32)
    circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), left, right)
xin = zk_amd.MultiLinearPolynomial.random(ctx, w, 0x6B72, 0)
seed = bytes(range(32))
for _ in range(4):
    ctx.synchronize(); t = time.perf_counter(); out, proof = gkr.gkr_prove(circ, xin, seed); print(mode, "prove ms", (time.perf_counter() - t) * 1e3)
