import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, zk_amd
from zk_amd import gkr
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
for w in (4, 10, 14, 16, 18, 20):
    rng = np.random.default_rng(w)
    circ = gkr.Circuit(ctx)
    for _ in range(8):
        circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32))
    x = zk_amd.MultiLinearPolynomial.random(ctx, w, 5, 0)
    seed = bytes(32)
    out, proof = gkr.gkr_prove(circ, x, seed)
    gkr.gkr_verify(circ, x, out, seed, proof)
    tp, tv = [], []
    for _ in range(7):
        ctx.synchronize(); t = time.perf_counter(); o, p = gkr.gkr_prove(circ, x, seed); tp.append(time.perf_counter() - t)
        ctx.synchronize(); t = time.perf_counter(); ok = gkr.gkr_verify(circ, x, o, seed, p); tv.append(time.perf_counter() - t)
    print("width 2^%d depth 8: prove %.3f ms  verify %.3f ms %s" % (w, sorted(tp)[3] * 1e3, sorted(tv)[3] * 1e3, ok))
    circ.free()
