"""Cross-check the C oracle against the independent Python big-int model (oracle/pyref.py) on seeded random
inputs, on all three fields.  Covers what no reference test pins: transcript bytes / challenges, forward FFT
values, BN254 Fr (SURVEY.md 8c "parity unpinned").  CPU only, small sizes.
"""
import random

import numpy as np
import pytest

from oracle import binding as orc
from oracle import pyref

FIELDS = [orc.BN254_FR, orc.BLS12_381_FR, orc.BLS12_377_FR]


def rand_ints(field, n, seed):
    rng = random.Random(seed)
    p = orc.modulus(field)
    return [rng.randrange(p) for _ in range(n)]


@pytest.mark.parametrize("field", FIELDS)
def test_constants(field):
    name, p, g, s = pyref.FIELDS[field]
    assert orc.modulus(field) == p
    assert orc.two_adicity(field) == s
    assert (p - 1) % (1 << s) == 0 and ((p - 1) >> s) % 2 == 1
    # F::one() in memory is R mod p; Montgomery layout matches pyref.to_mont_limbs
    assert orc.from_u64(field, 1).tolist() == pyref.to_mont_limbs(field, 1)
    for k in (0, 1, 2, 5, s):
        w = orc.to_int(field, orc.root_of_unity(field, 1 << k))
        assert w == pyref.root_of_unity(field, 1 << k)
        assert pow(w, 1 << k, p) == 1 and (k == 0 or pow(w, 1 << (k - 1), p) == p - 1)
    with pytest.raises(orc.OracleError):
        orc.root_of_unity(field, 1 << (s + 1))


@pytest.mark.parametrize("field", FIELDS)
def test_field_ops(field):
    p = orc.modulus(field)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 253) % p, (1 << 256) % p]
    vals = edge + rand_ints(field, 40, 11 + field)
    for i in range(0, len(vals) - 1):
        a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
        A, B = orc.from_int(field, a), orc.from_int(field, b)
        assert A.tolist() == pyref.to_mont_limbs(field, a)
        assert orc.to_int(field, orc.add(field, A, B)) == (a + b) % p
        assert orc.to_int(field, orc.sub(field, A, B)) == (a - b) % p
        assert orc.to_int(field, orc.mul(field, A, B)) == (a * b) % p
        assert orc.to_int(field, orc.pow_(field, A, 65537 + i)) == pow(a, 65537 + i, p)
        inv = orc.inverse(field, A)
        assert (inv is None) == (a == 0)
        if a:
            assert orc.to_int(field, inv) == pow(a, -1, p)
        assert orc.to_bytes_be(field, A) == a.to_bytes(32, "big")
    for n in (0, 1, 31, 32, 33, 64):
        b = bytes(random.Random(n).randrange(256) for _ in range(n))
        assert orc.to_int(field, orc.from_be_bytes_mod_order(field, b)) == int.from_bytes(b, "big") % p
    assert orc.to_int(field, orc.from_be_bytes_mod_order(field, b"\xff" * 32)) == ((1 << 256) - 1) % p


@pytest.mark.parametrize("field", FIELDS)
def test_fill_random_matches_model(field):
    got = orc.to_ints(field, orc.fill_random(field, 0x5EED0001, 64, first_index=1000))
    want = [pyref.random_element(field, 0x5EED0001, 1000 + i) for i in range(64)]
    assert got == want
    assert all(0 <= v < orc.modulus(field) for v in got)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n_vars", [1, 2, 3, 5, 7])
def test_partial_evaluate_all_positions(field, n_vars):
    p = orc.modulus(field)
    vals = rand_ints(field, 1 << n_vars, 100 * n_vars + field)
    tab = orc.from_ints(field, vals)
    model = pyref.MLE(field, n_vars, vals)
    rng = random.Random(n_vars)
    for initial_var in range(n_vars):
        for n_assign in range(0, n_vars - initial_var + 1):
            asg = [rng.choice([0, 1, p - 1, rng.randrange(p)]) for _ in range(n_assign)]
            got = orc.mle_partial_evaluate(field, n_vars, tab, initial_var, orc.from_ints(field, asg))
            assert orc.to_ints(field, got) == model.partial_evaluate(initial_var, asg).evals
    # misuse the reference panics on -> error code, never UB
    with pytest.raises(orc.OracleError):
        orc.mle_partial_evaluate(field, n_vars, tab, n_vars, orc.from_ints(field, [3]))
    with pytest.raises(orc.OracleError):
        orc.mle_partial_evaluate(field, n_vars, tab, 0, orc.from_ints(field, [3] * (n_vars + 1)))
    pt = rand_ints(field, n_vars, 5)
    assert orc.to_int(field, orc.mle_evaluate(field, n_vars, tab, orc.from_ints(field, pt))) == model.evaluate(pt)
    assert orc.mle_to_bytes(field, n_vars, tab) == model.to_bytes()


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,D,n_vars", [(1, 1, 4), (2, 2, 4), (3, 3, 3), (2, 1, 3), (1, 3, 2), (4, 4, 2), (2, 2, 1)])
@pytest.mark.parametrize("absorb", [False, True])
def test_sumcheck_transcript_matches_model(field, k, D, n_vars, absorb):
    p = orc.modulus(field)
    tabs_i = [rand_ints(field, 1 << n_vars, 7 * f + k + D + n_vars) for f in range(k)]
    tabs = [orc.from_ints(field, t) for t in tabs_i]
    model = pyref.Product([pyref.MLE(field, n_vars, t) for t in tabs_i])
    claimed = sum(model.prod_reduce()) % p
    rp, ch = orc.sumcheck_prove(field, n_vars, tabs, D, orc.from_int(field, claimed), absorb)
    mrp, mch = pyref.sumcheck_prove(model, claimed, D, absorb)
    assert [orc.to_ints(field, r) for r in rp] == mrp
    assert orc.to_ints(field, ch) == mch
    # verifier side (restated verifier.rs): accepts iff the degree bound covers the product
    tb = model.to_bytes() if absorb else None
    if D >= k:
        sub, vch = orc.sumcheck_verify_partial(field, D, orc.from_int(field, claimed), rp, tb)
        msub, mvch = pyref.sumcheck_verify_partial(field, claimed, mrp, tb)
        assert orc.to_int(field, sub) == msub and orc.to_ints(field, vch) == mvch == mch
        assert orc.to_int(field, orc.product_evaluate(field, n_vars, tabs, vch)) == msub
        if absorb:
            assert orc.sumcheck_verify(field, n_vars, tabs, D, orc.from_int(field, claimed), rp) is True
            assert pyref.sumcheck_verify(model, claimed, mrp) is True


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n_vars,n_terms", [(1, 2), (3, 5), (6, 20), (8, 256)])
def test_coeff_to_evaluation_matches_model(field, n_vars, n_terms):
    rng = random.Random(n_vars * 31 + field)
    p = orc.modulus(field)
    terms = {}
    while len(terms) < min(n_terms, 1 << n_vars):
        terms[rng.randrange(1 << n_vars)] = rng.randrange(p)
    keys = sorted(terms)
    got = orc.to_ints(field, orc.coeff_to_evaluation(field, n_vars, keys, orc.from_ints(field, [terms[k] for k in keys])))
    assert got == pyref.coeff_to_evaluation(field, n_vars, terms)
    # the table it produces is the MLE of the polynomial: evaluating it at a random point equals sum c_S prod_{v in S} r_v
    r = [rng.randrange(p) for _ in range(n_vars)]
    want = 0
    for key, cf in terms.items():
        t = cf
        for v in range(n_vars):
            if (key >> v) & 1:
                t = t * r[v] % p
        want = (want + t) % p
    assert pyref.MLE(field, n_vars, got).evaluate(r) == want


@pytest.mark.parametrize("field", FIELDS)
def test_fft_values_match_model(field):
    for lg in range(0, 7):
        n = 1 << lg
        vals = rand_ints(field, n, 31 * lg + field)
        got = orc.to_ints(field, orc.fft(field, orc.from_ints(field, vals)))
        assert got == pyref.fft(field, vals)
        if lg <= 4:
            assert got == pyref.dft_naive(field, vals)
        back = orc.to_ints(field, orc.ifft(field, orc.from_ints(field, got)))
        assert back == vals == pyref.ifft(field, got)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("lg", [0, 1, 2, 5, 10])
def test_ntt_fast_equals_faithful_fft(field, lg):
    v = orc.fill_random(field, 77 + lg, 1 << lg)
    f = orc.fft(field, v)
    assert np.array_equal(orc.ntt_fast(field, v), f)
    assert np.array_equal(orc.ntt_fast(field, f, inverse=True), v)
    assert np.array_equal(orc.ifft(field, f), v)


def test_parallel_fold_baseline_equals_faithful_fold():
    for field in FIELDS:
        tab = orc.fill_random(field, 4321, 1 << 12)
        r = orc.fill_random(field, 9, 1)
        got, used = orc.fold_msb_parallel(field, 12, tab, r[0], threads=4)
        assert used >= 1 and np.array_equal(got, orc.mle_partial_evaluate(field, 12, tab, 0, r))


def test_transcript_chain_matches_model():
    t, m = orc.Transcript(), pyref.Transcript()
    for chunk in (b"", b"abc", bytes(range(200)), b"\x00" * 136):
        t.append(chunk)
        m.append(chunk)
        assert t.sample_challenge() == m.sample_challenge()
    for field in FIELDS:
        assert orc.to_int(field, t.sample_field_element(field)) == m.sample_field_element(field)


@pytest.mark.parametrize("field", FIELDS)
def test_dft_point_equals_the_faithful_fft_outputs(field):
    """orc_dft_point (one output from the definition sum_j x[j] w^(jk), fft/src/lib.rs:39-45) against the faithful recursion
    and the Python model's naive DFT; used by the GPU suite to pin single outputs of the 2^24-point transform."""
    for lg in (0, 1, 4, 8, 13):
        n = 1 << lg
        v = orc.fill_random(field, 911 + lg, n)
        f = orc.fft(field, v) if lg <= 8 else orc.ntt_fast(field, v)
        g = orc.ifft(field, v) if lg <= 8 else orc.ntt_fast(field, v, inverse=True)
        for k in sorted({0, min(1, n - 1), n // 2, n - 1, (7 * n) // 11}):
            assert np.array_equal(orc.dft_point(field, v, k), f[k])
            assert np.array_equal(orc.dft_point(field, v, k, inverse=True), g[k])
    vals = rand_ints(field, 16, 5)
    naive = pyref.dft_naive(field, vals)
    for k in range(16):
        assert orc.to_int(field, orc.dft_point(field, orc.from_ints(field, vals), k)) == naive[k]


@pytest.mark.parametrize("field", FIELDS)
def test_ragged_verifier_matches_model(field):
    """verifier.rs:55-58 interpolates each round at its own length: the C restatement against the Python model on proofs
    whose rounds carry 0..6 evaluations (random, made to pass p(0) + p(1) round by round with the model's own claims)"""
    import random

    p = orc.modulus(field)
    rng = random.Random(field + 77)
    for lens in ([3, 2, 4, 1, 3], [1, 1], [5, 6, 2], [3, 0, 2], [2, 2, 2, 2]):
        claimed = rng.randrange(p)
        rounds, cur, alive = [], claimed, True
        for ln in lens:
            ys = [rng.randrange(p) for _ in range(ln)]
            if ln >= 2:
                ys[1] = (cur - ys[0]) % p
            elif ln == 1:
                ys[0] = cur * pow(2, -1, p) % p
            rounds.append(ys)
            if alive:
                try:
                    cur, _ = pyref.sumcheck_verify_partial(field, claimed, rounds)
                except ValueError:
                    alive = False
        arrs = [orc.from_ints(field, ys).reshape(-1, 4) for ys in rounds]
        if alive:
            want_sum, want_ch = pyref.sumcheck_verify_partial(field, claimed, rounds)
            got_sum, got_ch = orc.sumcheck_verify_partial_lengths(field, orc.from_int(field, claimed), arrs)
            assert orc.to_int(field, got_sum) == want_sum and orc.to_ints(field, got_ch) == want_ch
        else:
            with pytest.raises(orc.OracleError):
                orc.sumcheck_verify_partial_lengths(field, orc.from_int(field, claimed), arrs)


def test_vector_helpers_match_big_ints():
    """orc.sum_elems (iter().sum::<F>(), prover.rs:53-54) and the batched from_ints / to_ints against Python big ints"""
    import random

    for field in (0, 1, 2):
        p = orc.modulus(field)
        rng = random.Random(4400 + field)
        vs = [rng.randrange(-p, 2 * p) for _ in range(777)] + [0, 1, p - 1, p, -1]
        a = orc.from_ints(field, vs)
        assert np.array_equal(a, np.stack([orc.from_int(field, v) for v in vs]))
        assert orc.to_ints(field, a) == [v % p for v in vs]
        assert orc.to_int(field, orc.sum_elems(field, a)) == sum(vs) % p
        assert orc.to_int(field, orc.sum_elems(field, a[:0])) == 0
        assert orc.from_ints(field, []).shape == (0, 4) and orc.to_ints(field, np.zeros((0, 4), dtype=np.uint64)) == []
