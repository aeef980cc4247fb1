"""Run by tests/test_gpu_parity.py::test_skip1_rounds_bit_exact in a child process with ZK_SKIP1_MIN_PAIRS=1 (and
ZK_QUAD_MAX_PAIRS=0, so that the small-round kernel does not take these sizes), so that EVERY
fused round of the shapes that have the variant leaves out the t = 1 sums and the tail derives S(1) = S_prev(r_prev) - S(0)
(k_round_kd SKIP1 / TailDerive).  The proofs must stay bit-identical to the CPU oracle's (prover.rs:44-68 computes S(1)
directly; the identity is exact in F_p), including a WRONG claimed sum: the identity is about the prover's own sums."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the caller selects the kernel path with ZK_SKIP1_MIN_PAIRS / ZK_QUAD_MAX_PAIRS / ZK_PIPE_MAX_PAIRS (read once per process by
# the library) and the table sizes with ZK_CHECK_SIZES
SIZES = tuple(int(x) for x in os.environ.get("ZK_CHECK_SIZES", "2,3,7,11,13").split(","))

import numpy as np  # noqa: E402

import zk_amd  # noqa: E402
from oracle import binding as orc  # noqa: E402
from oracle import gkr_ref  # noqa: E402
from zk_amd import MultiLinearPolynomial as MLE  # noqa: E402
from zk_amd import ProductPoly, SumcheckProver, gkr  # noqa: E402

checked = 0
FIELDS = (zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR)[: int(os.environ.get("ZK_CHECK_FIELDS", "3"))]
for field in FIELDS:
    ctx = zk_amd.Context(field, 0)
    p = zk_amd.modulus(field)
    for k, D in ((2, 2), (3, 3), (1, 1), (2, 3)):
        for n in SIZES:
            tabs = [orc.fill_random(field, 7000 + 10 * k + f, 1 << n) for f in range(k)]
            claimed = np.zeros(4, dtype=np.uint64)
            for e in orc.prod_reduce(field, n, tabs):
                claimed = orc.add(field, claimed, e)
            for wrong in (False, True):
                s = orc.add(field, claimed, orc.from_int(field, 5)) if wrong else claimed
                want_rp, want_ch = orc.sumcheck_prove(field, n, tabs, D, s, False)
                pp = ProductPoly.new([MLE.new(ctx, n, t) for t in tabs])
                proof, ch = SumcheckProver(D).prove_partial(pp, s)
                assert np.array_equal(proof.round_polys, want_rp), (field, k, D, n, wrong)
                assert np.array_equal(ch, want_ch), (field, k, D, n, wrong)
                checked += 1
    # the two-term GKR layer shape through the merged kernel
    rng = random.Random(field)
    for n in (3, 8, 12) + tuple(x for x in SIZES if x > 12):
        tabs = [[[rng.randrange(p) for _ in range(1 << n)] for _ in range(kk)] for kk in (2, 1)]
        s = sum(a * b + c for a, b, c in zip(tabs[0][0], tabs[0][1], tabs[1][0])) % p
        want = gkr_ref.prove_partial_terms(field, tabs, 2, s)
        poly = gkr.SumOfProductsPoly([[MLE.new(ctx, n, zk_amd.fe_from_ints(field, t)) for t in term] for term in tabs])
        rp, ch, fin = gkr.prove_partial_terms(poly, 2, zk_amd.fe_from_int(field, s))
        assert [zk_amd.fe_to_ints(field, r) for r in rp] == want[0] and zk_amd.fe_to_ints(field, ch) == want[1]
        assert zk_amd.fe_to_ints(field, fin) == want[2]
        checked += 1
print(f"skip1 ok: {checked} proofs bit-exact (ZK_SKIP1_MIN_PAIRS={os.environ.get('ZK_SKIP1_MIN_PAIRS')} "
      f"ZK_QUAD_MAX_PAIRS={os.environ.get('ZK_QUAD_MAX_PAIRS')} ZK_PIPE_MAX_PAIRS={os.environ.get('ZK_PIPE_MAX_PAIRS')})")
