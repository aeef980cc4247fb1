"""A/B of prover wall-clock under environment switches, each arm in its own child process (the library reads its switches
once per process): python3 tools/ab_prover.py "ZK_LANE_ACC=0" "ZK_LANE_ACC=1" ...  -> median / min of prove_partial at
n = 20, 24 (k = 2, D = 2), n = 20 (k = 3, D = 3) and the GKR driver (depth 8, width 2^20), arms interleaved twice."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np
import zk_amd
from zk_amd import gkr
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
def med(ts): ts = sorted(ts); return ts[len(ts) // 2] * 1e3, ts[0] * 1e3
out = []
for n, k, D in ((20, 2, 2), (24, 2, 2), (20, 3, 3)):
    polys = [zk_amd.MultiLinearPolynomial.random(ctx, n, 77 + n, f << n) for f in range(k)]
    pp = zk_amd.ProductPoly.new(polys)
    s = pp.round_sums(1)
    claimed = zk_amd.fe_from_int(field, zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1]))
    prover = zk_amd.SumcheckProver(D)
    for _ in range(3): prover.prove_partial(pp, claimed)
    ts = []
    for _ in range(25):
        ctx.synchronize(); t = time.perf_counter(); prover.prove_partial(pp, claimed); ts.append(time.perf_counter() - t)
    out.append("n%%d_k%%d %%.4f/%%.4f" %% ((n, k) + med(ts)))
    for q in polys: q.free()
rng = np.random.default_rng(0x6B72)
w = 20
circ = gkr.Circuit(ctx)
for _ in range(8):
    circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32))
xin = zk_amd.MultiLinearPolynomial.random(ctx, w, 0x6B72, 0)
seed = bytes(range(32))
for _ in range(2): gkr.gkr_prove(circ, xin, seed)
ts = []
for _ in range(7):
    ctx.synchronize(); t = time.perf_counter(); o, pr = gkr.gkr_prove(circ, xin, seed); ts.append(time.perf_counter() - t)
out.append("gkr %%.3f/%%.3f" %% med(ts))
print(" | ".join(out))
''' % ROOT

arms = sys.argv[1:] or ["ZK_LANE_ACC=0", "ZK_LANE_ACC=1"]
for rep in range(2):
    for arm in arms:
        env = dict(os.environ)
        for kv in arm.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                env[k] = v
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        print(f"[{arm:>28}] {r.stdout.strip() or r.stderr[-400:]}", flush=True)
