"""Child of tools/r06_glds_sizes.sh (run under rocprofv3): three proofs each of the two-table product, the three-table product and the
product-plus-term shape at n = 18..23, in a fixed order, so that the kernel trace can be cut into (shape, n, round) cells."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_amd
from zk_amd import gkr
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
claimed = zk_amd.fe_from_int(zk_amd.BN254_FR, 7)
for n in range(18, 24):
    t = [zk_amd.MultiLinearPolynomial.random(ctx, n, 0x6000 + f, 0) for f in range(3)]
    pp2 = zk_amd.ProductPoly.new(t[:2])
    pp3 = zk_amd.ProductPoly.new(t)
    sop = gkr.SumOfProductsPoly([t[:2], t[2:]])
    for _ in range(3):
        zk_amd.SumcheckProver(2).prove_partial(pp2, claimed)
    for _ in range(3):
        zk_amd.SumcheckProver(3).prove_partial(pp3, claimed)
    for _ in range(3):
        gkr.prove_partial_terms(sop, 2, claimed)
    ctx.synchronize()
    for q in t:
        q.free()
print("done")
