"""Run by tests/test_gpu_parity.py::test_skip1_rounds_bit_exact in a child process with ZK_SKIP1_MIN_PAIRS=1 (and
ZK_QUAD_MAX_PAIRS=0, so that the small-round kernel does not take these sizes), so that EVERY
fused round of the shapes that have the variant leaves out the t = 1 sums and the tail derives S(1) = S_prev(r_prev) - S(0)
(k_round_kd SKIP1 / TailDerive).  The proofs must stay bit-identical to the CPU oracle's (prover.rs:44-68 computes S(1)
directly; the identity is exact in F_p), including a WRONG claimed sum: the identity is about the prover's own sums."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the caller selects the kernel path with ZK_SKIP1_MIN_PAIRS / ZK_QUAD_MAX_PAIRS / ZK_PIPE_MAX_PAIRS (read once per process by
# the library) and the table sizes with ZK_CHECK_SIZES
SIZES = tuple(int(x) for x in os.environ.get("ZK_CHECK_SIZES", "2,3,7,11,13").split(","))

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_cache  # noqa: E402  (oracle answers: read from $ZK_ORACLE_CACHE when the parent test prefilled it, else computed here)

import zk_amd  # noqa: E402
from zk_amd import MultiLinearPolynomial as MLE  # noqa: E402
from zk_amd import ProductPoly, SumcheckProver, gkr  # noqa: E402

KD = ((2, 2), (3, 3), (1, 1), (2, 3))
SEED = 7000


def spec(sizes, n_fields):
    """every oracle answer this script needs for (sizes, n_fields): the parent prefills the cache with it"""
    out = []
    for field in range(n_fields):
        for k, D in KD:
            for n in sizes:
                for wrong in (0, 5):
                    out.append(["sc", field, k, D, n, SEED + 10 * k, wrong])
        for n in (3, 8, 12) + tuple(x for x in sizes if x > 12):
            out.append(["terms", field, n])
    return out


def main():
    checked = 0
    fields = (zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR)[: int(os.environ.get("ZK_CHECK_FIELDS", "3"))]
    for field in fields:
        ctx = zk_amd.Context(field, 0)
        for k, D in KD:
            for n in SIZES:
                tabs = oracle_cache.sumcheck_tables(field, k, n, SEED + 10 * k)
                both = []
                for wrong in (0, 5):   # a WRONG claimed sum too: the SKIP1 identity is about the prover's own sums
                    _, s, want_rp, want_ch = oracle_cache.sumcheck_case(field, k, D, n, SEED + 10 * k, wrong, tabs=tabs)
                    pp = ProductPoly.new([MLE.new(ctx, n, t) for t in tabs])
                    proof, ch = SumcheckProver(D).prove_partial(pp, s)
                    assert np.array_equal(proof.round_polys, want_rp), (field, k, D, n, wrong)
                    assert np.array_equal(ch, want_ch), (field, k, D, n, wrong)
                    both.append((pp, s, want_rp, want_ch))
                    checked += 1
                # the same two proofs (+ a third on the first one's handles) side by side: zk_sumcheck_prove_batch under the same
                # forced kernel selection -- the batched twins of whatever kernels the switches picked
                got = SumcheckProver(D).prove_partial_batch([both[0][0], both[1][0], both[0][0]], [both[0][1], both[1][1], both[0][1]])
                for (proof, ch), (_, _, want_rp, want_ch) in zip(got, both + both[:1]):
                    assert np.array_equal(proof.round_polys, want_rp) and np.array_equal(ch, want_ch), ("batch", field, k, D, n)
                checked += 1
        # the two-term GKR layer shape (A.B + C) through the merged kernel, against the big-int definition (oracle/gkr_ref.py)
        for n in (3, 8, 12) + tuple(x for x in SIZES if x > 12):
            tabs, s, want_rp, want_ch, want_fin = oracle_cache.terms_case(field, n)
            poly = gkr.SumOfProductsPoly([[MLE.new(ctx, n, t) for t in term] for term in tabs])
            rp, ch, fin = gkr.prove_partial_terms(poly, 2, s)
            assert np.array_equal(rp, want_rp) and np.array_equal(ch, want_ch), (field, n)
            assert np.array_equal(fin, want_fin), (field, n)
            checked += 1
    print(f"skip1 ok: {checked} proofs bit-exact (ZK_SKIP1_MIN_PAIRS={os.environ.get('ZK_SKIP1_MIN_PAIRS')} "
          f"ZK_QUAD_MAX_PAIRS={os.environ.get('ZK_QUAD_MAX_PAIRS')} ZK_PIPE_MAX_PAIRS={os.environ.get('ZK_PIPE_MAX_PAIRS')})")


if __name__ == "__main__":
    main()
