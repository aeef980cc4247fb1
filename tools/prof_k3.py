import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
pp = zk_amd.ProductPoly.new([zk_amd.MultiLinearPolynomial.random(ctx, n, 0x6000 + f, 0) for f in range(3)])
claimed = zk_amd.fe_from_int(zk_amd.BN254_FR, 7)
p3 = zk_amd.SumcheckProver(3)
p3.prove_partial(pp, claimed)
ts = []
for _ in range(5):
    ctx.synchronize(); t = time.perf_counter(); p3.prove_partial(pp, claimed); ts.append((time.perf_counter() - t) * 1e3)
print("k3 n", n, "ms", [round(x, 3) for x in ts])
