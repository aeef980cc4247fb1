"""CPU-side checks of libzk_amd.so: it loads, exports every symbol include/zk_amd.h declares, fails loudly without
a GPU, and its HOST logic (field helpers, Keccak-256 transcript, verifier) agrees with the oracle.  No kernel runs.
"""
import ctypes
import random

import numpy as np
import pytest

import zk_amd
from oracle import binding as orc
from zk_amd import _lib

FIELDS = [zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR]


def test_library_exports_every_declared_symbol():
    names = _lib.declared_symbols()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(_lib.lib, n)]
    assert missing == []
    assert _lib.lib.zk_abi_version() == 6


def test_every_declared_symbol_has_a_python_signature():
    untyped = [n for n in _lib.declared_symbols()
               if n not in _lib._sig and n not in ("zk_abi_version", "zk_strerror", "zk_last_hip_error")]
    assert untyped == []


def test_strerror_reproduces_reference_messages():
    s = lambda code: _lib.lib.zk_strerror(code).decode()
    assert s(-1) == "evaluation vec len should equal 2^n_vars"                      # evaluation_form.rs:20
    assert s(-2) == "evaluate must assign to all variables"                         # evaluation_form.rs:85
    assert s(-3) == "cannot create product polynomial from empty polynomials"       # product_poly.rs:16
    assert s(-4).endswith("don't share the same number of variables")               # product_poly.rs:25
    assert s(-6) == "values must be a power of 2"                                   # fft/src/lib.rs:29
    assert s(-8) == "invalid proof: require 1 round poly for each variable in poly" # verifier.rs:18
    assert s(-9) == "verifier check failed: claimed_sum != p(0) + p(1)"             # verifier.rs:64


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zk_amd.ZkError) as e:
        zk_amd.Context(zk_amd.BN254_FR, 0)
    assert e.value.code == -22 and "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("field", FIELDS)
def test_host_field_helpers_match_oracle(field):
    p = orc.modulus(field)
    assert zk_amd.modulus(field) == p
    assert zk_amd.two_adicity(field) == orc.two_adicity(field)
    rng = random.Random(field)
    for v in [0, 1, 2, p - 1, (1 << 255) % p] + [rng.randrange(p) for _ in range(20)]:
        a = zk_amd.fe_from_int(field, v)
        assert np.array_equal(a, orc.from_int(field, v))
        assert zk_amd.fe_to_int(field, a) == v
    for n in (0, 1, 31, 32, 33, 64):
        b = bytes(rng.randrange(256) for _ in range(n))
        out = np.zeros(4, dtype=np.uint64)
        _lib.check(_lib.lib.zk_fe_from_be_bytes_mod_order(field, b, n, out.ctypes.data_as(_lib.u64p)))
        assert np.array_equal(out, orc.from_be_bytes_mod_order(field, b))
    # not-reduced canonical input is rejected, not silently wrapped
    limbs = np.array([(p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    assert _lib.lib.zk_fe_from_canonical(field, limbs.ctypes.data_as(_lib.u64p), out.ctypes.data_as(_lib.u64p)) == -20
    assert _lib.lib.zk_field_modulus(7, out.ctypes.data_as(_lib.u64p)) == -21


def test_keccak_and_transcript_match_oracle():
    assert zk_amd.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert zk_amd.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    rng = random.Random(5)
    for n in (1, 135, 136, 137, 272, 273, 5000):
        msg = bytes(rng.randrange(256) for _ in range(n))
        assert zk_amd.keccak256(msg) == orc.keccak256(msg)
    t, o = zk_amd.Transcript(), orc.Transcript()
    for chunk in (b"", b"abc", bytes(range(200)), b"\x00" * 136, bytes(300)):
        t.append(chunk)
        o.append(chunk)
        assert t.sample_challenge() == o.sample_challenge()
    for field in FIELDS:
        assert np.array_equal(t.sample_field_element(field), o.sample_field_element(field))
    assert np.array_equal(t.sample_n_field_elements(FIELDS[0], 3),
                          np.stack([o.sample_field_element(FIELDS[0]) for _ in range(3)]))


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,D,n_vars", [(1, 1, 3), (2, 2, 4), (3, 3, 3), (2, 4, 2)])
def test_verify_partial_matches_oracle(field, k, D, n_vars):
    """verifier.rs:38-78 host logic on proofs produced by the ORACLE prover (no GPU involved)."""
    tabs = [orc.fill_random(field, 100 + f, 1 << n_vars) for f in range(k)]
    claimed = orc.sum_elems(field, orc.prod_reduce(field, n_vars, tabs))   # iter().sum::<F>()
    rp, ch = orc.sumcheck_prove(field, n_vars, tabs, D, claimed, absorb_table=False)
    sub = zk_amd.SumcheckVerifier.verify_partial(field, zk_amd.SumcheckProof(claimed, rp))
    osub, och = orc.sumcheck_verify_partial(field, D, claimed, rp)
    assert np.array_equal(sub.sum, osub) and np.array_equal(sub.challenges, och) and np.array_equal(och, ch)
    # a wrong claimed sum is rejected with the reference's message
    bad = orc.add(field, claimed, orc.from_int(field, 1))
    rp2, _ = orc.sumcheck_prove(field, n_vars, tabs, D, bad, absorb_table=False)
    with pytest.raises(zk_amd.ZkError, match="claimed_sum != p\\(0\\) \\+ p\\(1\\)"):
        zk_amd.SumcheckVerifier.verify_partial(field, zk_amd.SumcheckProof(bad, rp2))


def _consistent_ragged_proof(field, lens, seed):
    """a proof whose round r carries lens[r] evaluations and passes every p(0) + p(1) check: random evaluations with ys[1]
    (or, for a constant, ys[0]) fixed by the running claim, the claim after each round taken from the oracle's verifier"""
    rng = random.Random(seed)
    p = orc.modulus(field)
    claimed = orc.from_int(field, rng.randrange(p))
    inv2 = orc.inverse(field, orc.from_int(field, 2))
    rounds, cur = [], claimed
    for ln in lens:
        ys = orc.from_ints(field, [rng.randrange(p) for _ in range(ln)]).reshape(ln, 4)
        if ln >= 2:
            ys[1] = orc.sub(field, cur, ys[0])
        elif ln == 1:
            ys[0] = orc.mul(field, cur, inv2)
        rounds.append(ys)
        try:
            cur, _ = orc.sumcheck_verify_partial_lengths(field, claimed, rounds)
        except orc.OracleError:
            break   # a zero-length round with a non-zero claim: the remaining rounds are never looked at
    while len(rounds) < len(lens):
        rounds.append(orc.from_ints(field, [rng.randrange(p) for _ in range(lens[len(rounds)])]).reshape(-1, 4))
    return claimed, rounds


@pytest.mark.parametrize("field", FIELDS)
def test_verify_partial_interpolates_each_round_at_its_own_length(field):
    """verifier.rs:55-58 + univariate_poly.rs:43-49: round_polys is a Vec<Vec<F>> and every round is interpolated at the
    length it carries.  Library (zk_sumcheck_verify_partial_lengths) against the oracle on proofs with mixed lengths,
    including constants (1 evaluation) and the zero polynomial (0 evaluations)."""
    cases = [[3, 2, 4, 1, 3], [1, 1, 1], [2, 5, 2, 7, 3, 3], [4], [3, 0, 3], [9, 2, 6], []]
    for i, lens in enumerate(cases):
        claimed, rounds = _consistent_ragged_proof(field, lens, 40 + i)
        proof = zk_amd.SumcheckProof(claimed, rounds)
        try:
            want = orc.sumcheck_verify_partial_lengths(field, claimed, rounds)
        except orc.OracleError as e:
            assert 0 in lens and e.code == -9
            with pytest.raises(zk_amd.ZkError, match="claimed_sum != p\\(0\\) \\+ p\\(1\\)"):
                zk_amd.SumcheckVerifier.verify_partial(field, proof)
            continue
        sub = zk_amd.SumcheckVerifier.verify_partial(field, proof)
        assert np.array_equal(sub.sum, want[0]) and np.array_equal(sub.challenges, want[1])
        if lens:   # one evaluation off by one in the last round: both reject
            rounds[-1][0] = orc.add(field, rounds[-1][0], orc.from_int(field, 1))
            with pytest.raises(orc.OracleError):
                orc.sumcheck_verify_partial_lengths(field, claimed, rounds)
            with pytest.raises(zk_amd.ZkError, match="claimed_sum"):
                zk_amd.SumcheckVerifier.verify_partial(field, zk_amd.SumcheckProof(claimed, rounds))
    # a zero claim with an empty first round passes that round in both (zero polynomial: p(0) + p(1) = 0)
    zero = np.zeros(4, dtype=np.uint64)
    rounds = [np.zeros((0, 4), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64)]
    want = orc.sumcheck_verify_partial_lengths(field, zero, rounds)
    sub = zk_amd.SumcheckVerifier.verify_partial(field, zk_amd.SumcheckProof(zero, rounds))
    assert np.array_equal(sub.sum, want[0]) and np.array_equal(sub.challenges, want[1]) and not sub.sum.any()


def test_argument_validation_without_gpu():
    # ProductPoly::new on an empty list (product_poly.rs:15-17) needs no device
    arr = (ctypes.c_void_p * 1)()
    assert _lib.lib.zk_product_check(ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p)), 0) == -3
    assert _lib.lib.zk_ctx_create(9, 0, ctypes.byref(ctypes.c_void_p())) == -21


def test_batch_argument_validation_without_gpu():
    """zk_sumcheck_prove_batch / zk_batch_last_stats reject bad arguments before anything touches a device"""
    import ctypes as c

    import numpy as np

    lib = _lib.lib
    s = np.zeros(4, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    sp, op = s.ctypes.data_as(_lib.u64p), out.ctypes.data_as(_lib.u64p)
    assert lib.zk_sumcheck_prove_batch(None, 1, None, 2, 2, sp, 0, op, op) == -20          # no context, no handles
    handles = (c.c_void_p * 2)()
    assert lib.zk_sumcheck_prove_batch(None, 0, c.cast(handles, _lib.vpp), 2, 2, sp, 0, op, op) == -20
    assert lib.zk_batch_last_stats(None, None) == -20
    a, b = c.c_uint64(7), c.c_uint64(7)
    assert lib.zk_batch_last_stats(c.byref(a), c.byref(b)) == 0 and (a.value, b.value) == (0, 0)   # nothing batched on this thread yet


def test_verifier_round_count_is_never_an_allocation_size():
    """A hostile round count must come back as a status, never as an exception across the C ABI: the uniform-degree entry
    points used to size a std::vector by n_rounds + 1 before looking at the proof (UINT64_MAX wrapped it to 0)."""
    field = zk_amd.BN254_FR
    u64p = ctypes.POINTER(ctypes.c_uint64)
    one = orc.from_int(field, 1)
    # claimed sum 5 against a first round that sums to 2: rejected at round 0, whatever the count says
    rps = np.ascontiguousarray(np.stack([one, one, one]))
    claimed = np.ascontiguousarray(orc.from_int(field, 5))
    out_sum = np.zeros(4, dtype=np.uint64)
    out_ch = np.zeros(4, dtype=np.uint64)
    for n_rounds in (2**64 - 1, 2**62, 2**40):
        rc = _lib.lib.zk_sumcheck_verify_partial(field, ctypes.c_uint64(n_rounds), 2, claimed.ctypes.data_as(u64p),
                                                 rps.ctypes.data_as(u64p), out_sum.ctypes.data_as(u64p),
                                                 out_ch.ctypes.data_as(u64p))
        assert rc == -9
    # and a missing proof pointer with a non-zero count is a plain argument error
    assert _lib.lib.zk_sumcheck_verify_partial(field, ctypes.c_uint64(2**63), 2, claimed.ctypes.data_as(u64p), None,
                                               out_sum.ctypes.data_as(u64p), out_ch.ctypes.data_as(u64p)) == -20
