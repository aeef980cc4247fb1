"""A/B of the fused round kernels WITHOUT SKIP1 over three and four tables -- k_round_kd<2,2,fused,EXTRA=1> and k_round_kd<3,3,fused,EXTRA=1>,
the instantiations that ZK_KD_MIN_BLOCKS leaves at one wave per SIMD (round_kernels.cuh) -- against a build that forces two waves per
SIMD on them (52 / 292 spilled registers).  Library variants at ab_tmp/libzk_kd_<name>.so; SKIP1 is switched off so that the big rounds
take these kernels: python3 tools/ab_kd_occupancy.py default all2"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
import zk_amd
from zk_amd import gkr
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
out = []
for kk, D, n in ((2, 2, 20), (3, 3, 20), (2, 2, 22)):
    prod = [zk_amd.MultiLinearPolynomial.random(ctx, n, 900 + f, f << n) for f in range(kk)]
    extra = [zk_amd.MultiLinearPolynomial.random(ctx, n, 990, 7 << n)]
    poly = gkr.SumOfProductsPoly([prod, extra])
    zero = zk_amd.fe_from_int(field, 0)
    for _ in range(3): gkr.prove_partial_terms(poly, D, zero)
    ts = []
    for _ in range(15):
        ctx.synchronize(); t = time.perf_counter(); gkr.prove_partial_terms(poly, D, zero); ts.append(time.perf_counter() - t)
    ts.sort()
    out.append("terms {%%d,1} D=%%d n=%%d: %%.4f / %%.4f ms" %% (kk, D, n, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
    for q in prod + extra: q.free()
print(" | ".join(out))
''' % ROOT
for rep in range(3):
    for v in sys.argv[1:]:
        env = dict(os.environ, ZK_AMD_LIB=os.path.join(ROOT, "ab_tmp", f"libzk_kd_{v}.so"), ZK_SKIP1_MIN_PAIRS=str(1 << 40))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        print(f"[{v:>8}] {r.stdout.strip() or r.stderr[-500:]}", flush=True)
