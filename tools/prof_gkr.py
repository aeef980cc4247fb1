import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zk_amd
from zk_amd import gkr
w = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
rng = np.random.default_rng(0x6B72)
circ = gkr.Circuit(ctx)
for _ in range(depth):
    circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32),
                   rng.integers(0, 1 << w, 1 << w, dtype=np.uint32))
xin = zk_amd.MultiLinearPolynomial.random(ctx, w, 0x6B72, 0)
seed = bytes(range(32))
out, proof = gkr.gkr_prove(circ, xin, seed)
for _ in range(3):
    ctx.synchronize(); t = time.perf_counter(); out, proof = gkr.gkr_prove(circ, xin, seed); print("prove ms", (time.perf_counter() - t) * 1e3)
for _ in range(0 if os.environ.get("ZK_GKR_NOVERIFY") else 3):
    t = time.perf_counter(); ok = gkr.gkr_verify(circ, xin, out, seed, proof); print("verify ms", (time.perf_counter() - t) * 1e3, ok)
