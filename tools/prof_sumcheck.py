import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zk_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
A = zk_amd.MultiLinearPolynomial.random(ctx, n, 1, 0)
B = zk_amd.MultiLinearPolynomial.random(ctx, n, 1, 1 << n)
pp = zk_amd.ProductPoly.new([A, B])
s = pp.round_sums(1)
claimed = zk_amd.fe_from_int(field, zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1]))
prover = zk_amd.SumcheckProver(2)
prover.prove_partial(pp, claimed)
ts = []
for _ in range(reps):
    ctx.synchronize(); t = time.perf_counter(); prover.prove_partial(pp, claimed); ts.append(time.perf_counter() - t)
print("n", n, "ms", [round(x * 1e3, 3) for x in ts])
