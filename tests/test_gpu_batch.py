"""zk_sumcheck_prove_batch: B independent prove_partial calls (prover.rs:24-30, one per ProductPoly) proved side by side -- one launch
per round for all proofs (blockIdx.y = proof), every proof with its own transcript.  Each proof must equal, bit for bit, the oracle's
proof of its own tables and claimed sum (and therefore what zk_sumcheck_prove returns for it alone).  Config 4 of BASELINE.json as
SURVEY 8d words it -- "8 independent layers x (k = 3 factors, n = 20)" -- is the case at its own size."""
import numpy as np
import pytest

import zk_amd
from oracle import binding as orc
from zk_amd import MultiLinearPolynomial as MLE
from zk_amd import ProductPoly, SumcheckProver

pytestmark = pytest.mark.gpu
FIELDS = [zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR]
_ctx = {}


def ctx_for(field):
    if field not in _ctx:
        _ctx[field] = zk_amd.Context(field, 0)
    return _ctx[field]


def _cases(field, B, k, D, n, seed):
    out = []
    for b in range(B):
        tabs = [orc.fill_random(field, seed + 16 * b + f, 1 << n) for f in range(k)]
        s = orc.sum_elems(field, orc.prod_reduce(field, n, tabs))
        if b % 3 == 2:
            s = orc.add(field, s, orc.from_int(field, 11 + b))   # a wrong claimed sum is proved all the same (prover.rs never checks it)
        out.append((tabs, s, orc.sumcheck_prove(field, n, tabs, D, s, False)))
    return out


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,D", [(2, 2), (3, 3)])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 9, 12, 13, 14, 15])
def test_batch_matches_oracle_per_proof(field, k, D, n):
    c = ctx_for(field)
    for B in (2, 5, 8):
        cases = _cases(field, B, k, D, n, 31000 + 100 * n)
        for consume in (False, True):
            polys = [ProductPoly.new([MLE.new(c, n, t) for t in tabs]) for tabs, _, _ in cases]
            got = SumcheckProver(D).prove_partial_batch(polys, np.stack([s for _, s, _ in cases]), consume=consume)
            merged, replayed = zk_amd.batch_last_stats()
            for b, ((proof, ch), (tabs, s, (want_rp, want_ch))) in enumerate(zip(got, cases)):
                assert np.array_equal(proof.round_polys, want_rp), (B, b, consume)
                assert np.array_equal(ch, want_ch), (B, b, consume)
                if not consume:
                    for q, t in zip(polys[b].polynomials, tabs):
                        assert np.array_equal(q.evaluation_slice(), t)   # inputs intact
            # these shapes have a batched twin for every kernel on their path: nothing was issued proof by proof.  (Up to nine
            # variables the classic one-workgroup finisher may take the whole proof in a single launch, which is replayed.)
            if n >= 10:
                assert merged > 0 and replayed == 0, (merged, replayed, B, n)


@pytest.mark.parametrize("k,D,n", [(1, 1, 9), (2, 3, 8), (1, 2, 12), (4, 4, 7), (2, 5, 6), (5, 2, 6), (1, 0, 4), (3, 2, 10), (2, 2, 0)])
def test_batch_of_shapes_without_a_batched_twin_still_exact(k, D, n):
    """shapes whose kernels have no batched form are replayed proof by proof (or proved one by one): same proofs"""
    field = zk_amd.BN254_FR
    c = ctx_for(field)
    B = 3
    if n == 0:
        polys = [ProductPoly.new([MLE.new(c, 0, orc.fill_random(field, 5 + f, 1)) for f in range(k)]) for _ in range(B)]
        got = SumcheckProver(D).prove_partial_batch(polys, np.zeros((B, 4), dtype=np.uint64))
        assert all(p.round_polys.shape[0] == 0 for p, _ in got)
        return
    cases = _cases(field, B, k, D, n, 32000 + 10 * k + D)
    polys = [ProductPoly.new([MLE.new(c, n, t) for t in tabs]) for tabs, _, _ in cases]
    got = SumcheckProver(D).prove_partial_batch(polys, np.stack([s for _, s, _ in cases]))
    for (proof, ch), (_, _, (want_rp, want_ch)) in zip(got, cases):
        assert np.array_equal(proof.round_polys, want_rp) and np.array_equal(ch, want_ch)


def test_batch_groups_shared_handles_and_errors():
    field = zk_amd.BLS12_381_FR
    c = ctx_for(field)
    k, D, n = 2, 2, 11
    cases = _cases(field, 11, k, D, n, 33000)     # 11 proofs = a group of 8 + a group of 3
    polys = [ProductPoly.new([MLE.new(c, n, t) for t in tabs]) for tabs, _, _ in cases]
    got = SumcheckProver(D).prove_partial_batch(polys, np.stack([s for _, s, _ in cases]))
    for (proof, ch), (_, _, (want_rp, want_ch)) in zip(got, cases):
        assert np.array_equal(proof.round_polys, want_rp) and np.array_equal(ch, want_ch)
    # two proofs over the SAME handles with consume requested: proved out of place (in-place folds would race), inputs intact
    twice = SumcheckProver(D).prove_partial_batch([polys[0], polys[0]], np.stack([cases[0][1], cases[3][1]]), consume=True)
    assert np.array_equal(twice[0][0].round_polys, cases[0][2][0])
    want_rp, want_ch = orc.sumcheck_prove(field, n, cases[0][0], D, cases[3][1], False)
    assert np.array_equal(twice[1][0].round_polys, want_rp) and np.array_equal(twice[1][1], want_ch)
    assert np.array_equal(polys[0].polynomials[0].evaluation_slice(), cases[0][0][0])
    # a batch is one size and one shape
    small = ProductPoly.new([MLE.new(c, n - 1, t[: 1 << (n - 1)]) for t in cases[0][0]])
    with pytest.raises(zk_amd.ZkError):
        SumcheckProver(D).prove_partial_batch([polys[1], small], np.stack([cases[1][1], cases[1][1]]))
    assert SumcheckProver(D).prove_partial_batch([], np.zeros((0, 4), dtype=np.uint64)) == []
    # the single-proof prover is unaffected by a batch before it (context-wide buffers are shared)
    proof, ch = SumcheckProver(D).prove_partial(polys[2], cases[2][1])
    assert np.array_equal(proof.round_polys, cases[2][2][0]) and np.array_equal(ch, cases[2][2][1])


@pytest.mark.parametrize("k,D", [(3, 3), (2, 2)])
def test_config4_eight_layers_n20_batched_bit_exact(k, D):
    """BASELINE config[3] as SURVEY 8d words it: 8 independent layers x (k factors, n = 20), D = k -- every one of the eight proofs,
    proved in ONE batch at the shipped thresholds (LEAD round 0, SKIP1 + LEAD fused rounds, quad rounds, pipelined rounds, finisher),
    equal to the faithful oracle's proof of that layer."""
    field = zk_amd.BN254_FR
    c = ctx_for(field)
    n, B = 20, 8
    polys, sums, want = [], [], []
    for layer in range(B):
        fs = [MLE.random(c, n, 0x6000 + 16 * layer + f, 0) for f in range(k)]
        tabs = [q.evaluation_slice() for q in fs]
        s = orc.sum_elems(field, orc.prod_reduce(field, n, tabs))
        polys.append(ProductPoly.new(fs))
        sums.append(s)
        want.append(orc.sumcheck_prove(field, n, tabs, D, s, False))
    got = SumcheckProver(D).prove_partial_batch(polys, np.stack(sums))
    merged, replayed = zk_amd.batch_last_stats()
    assert merged > 0 and replayed == 0, (merged, replayed)
    for layer, ((proof, ch), (want_rp, want_ch)) in enumerate(zip(got, want)):
        assert np.array_equal(proof.round_polys, want_rp), f"layer {layer}: round polynomials"
        assert np.array_equal(ch, want_ch), f"layer {layer}: challenges"
    sub = zk_amd.SumcheckVerifier.verify_partial(field, got[3][0])
    assert np.array_equal(polys[3].evaluate(sub.challenges), sub.sum)
    for pp in polys:
        for q in pp.polynomials:
            q.free()


@pytest.mark.parametrize("k,D,n,B", [(2, 2, 24, 2), (3, 3, 22, 3), (2, 2, 21, 8)])
def test_batch_at_the_big_round_sizes_equals_the_single_proofs(k, D, n, B):
    """the batched twins on grids the n <= 20 cases do not reach (round 0 over 2^24 elements per table, fused rounds of 2^22 pairs: 512-2048
    work blocks per proof): every batched proof equals the single-proof prover's, whose n = 24 / n = 20 proofs the parity tests compare
    with the oracle (tests/test_gpu_parity.py::test_config3_n24_prover_and_fold_bit_exact)."""
    field = zk_amd.BN254_FR
    c = ctx_for(field)
    polys = [ProductPoly.new([MLE.random(c, n, 0x7700 + 16 * b + f, 0) for f in range(k)]) for b in range(B)]
    sums = []
    for pp in polys:
        s = pp.round_sums(1)
        sums.append(orc.add(field, s[0], s[1]))
    single = [SumcheckProver(D).prove_partial(pp, s) for pp, s in zip(polys, sums)]
    got = SumcheckProver(D).prove_partial_batch(polys, np.stack(sums))
    merged, replayed = zk_amd.batch_last_stats()
    assert merged > 0 and replayed == 0, (merged, replayed)
    for b, ((p1, c1), (p2, c2)) in enumerate(zip(single, got)):
        assert np.array_equal(p1.round_polys, p2.round_polys) and np.array_equal(c1, c2), b
    sub = zk_amd.SumcheckVerifier.verify_partial(field, got[-1][0])     # true claimed sums: the proofs verify
    assert np.array_equal(polys[-1].evaluate(sub.challenges), sub.sum)
    for pp in polys:
        for q in pp.polynomials:
            q.free()
