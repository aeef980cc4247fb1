"""A/B of the 2^24-point NTT under environment switches (each arm in a child process):
python3 tools/ab_ntt.py ZK_NTT_MAX_LOG=8 ZK_NTT_MAX_LOG=6"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, zk_amd
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
out = []
for lg in (20, 22, 24):
    x = zk_amd.MultiLinearPolynomial.random(ctx, lg, 5, 0); y = zk_amd.MultiLinearPolynomial.alloc(ctx, lg); z = zk_amd.MultiLinearPolynomial.alloc(ctx, lg)
    zk_amd.ntt(ctx, x, y); zk_amd.ntt(ctx, y, z, inverse=True)
    ok = (z == x)
    out.append("2^%%d fwd %%.3f inv %%.3f ms rt=%%s" %% (lg, zk_amd.bench_ntt(ctx, x, y, False, 10), zk_amd.bench_ntt(ctx, x, y, True, 10), ok))
    x.free(); y.free(); z.free()
print(" | ".join(out))
''' % ROOT
for rep in range(2):
    for arm in sys.argv[1:]:
        env = dict(os.environ)
        for kv in arm.split(","):
            k, v = kv.split("=", 1); env[k] = v
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        print(f"[{arm:>24}] {r.stdout.strip() or r.stderr[-600:]}", flush=True)
