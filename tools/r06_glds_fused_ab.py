"""the LDS-DMA round kernels on (ZK_ROUND_GLDS=1) vs off (=0): prove_partial at n = 20, 22, 24 for k = 2 and k = 3,
each arm in its own process, interleaved three times"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import zk_amd
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
def med(ts): ts = sorted(ts); return ts[len(ts) // 2] * 1e3, ts[0] * 1e3
out = []
for n, k, D in ((20, 2, 2), (22, 2, 2), (24, 2, 2), (20, 3, 3), (22, 3, 3), (24, 3, 3)):
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
print(" | ".join(out))
''' % ROOT
for rep in range(3):
    for arm in ("0", "1"):
        env = dict(os.environ, ZK_ROUND_GLDS=arm)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        print(f"[ZK_ROUND_GLDS={arm}] {r.stdout.strip() or r.stderr[-400:]}", flush=True)
