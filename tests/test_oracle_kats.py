"""Pin the CPU oracle against every known-answer test the reference holds for the hot path.

Each test names the reference test it restates (file:line, paths relative to the reference checkout).
The reference runs these on ark_bls12_381::Fr; the small-integer KATs are field-agnostic, so they are
also run on BN254 Fr and BLS12-377 Fr.  CPU only.
"""
import numpy as np
import pytest

from oracle import binding as orc
from oracle import pyref

FIELDS = [orc.BLS12_381_FR, orc.BN254_FR, orc.BLS12_377_FR]


def F(field, vals):
    return orc.from_ints(field, vals)


def ints(field, arr):
    p = orc.modulus(field)
    return [v if v <= p // 2 else v - p for v in orc.to_ints(field, arr)]


# ---- pairing_index.rs:32-46 (insert_bit) ----
def test_insert_bit_kats():
    assert orc.insert_bit(0b10101, 0, 0) == 0b101010
    assert orc.insert_bit(0b10101, 0, 1) == 0b101011
    assert orc.insert_bit(0b10101, 5, 0) == 0b010101
    assert orc.insert_bit(0b10101, 5, 1) == 0b110101
    assert orc.insert_bit(0b10, 1, 0) == 0b100
    assert orc.insert_bit(0b10, 1, 1) == 0b110


# ---- pairing_index.rs:49-97 (index_pair) ----
@pytest.mark.parametrize("n,idx,expect", [
    (3, 0, [(0, 4), (1, 5), (2, 6), (3, 7)]),
    (3, 1, [(0, 2), (1, 3), (4, 6), (5, 7)]),
    (3, 2, [(0, 1), (2, 3), (4, 5), (6, 7)]),
    (2, 0, [(0, 2), (1, 3)]),
    (2, 1, [(0, 1), (2, 3)]),
    (1, 0, [(0, 1)]),
])
def test_index_pair_kats(n, idx, expect):
    assert orc.index_pair(n, idx) == expect
    assert pyref.index_pair(n, idx) == expect


def test_index_pair_underflow_is_error():
    with pytest.raises(orc.OracleError):
        orc.index_pair(2, 2)
    with pytest.raises(orc.OracleError):
        orc.index_pair(0, 0)


# ---- evaluation_form.rs:111-125 (constructor length check) ----
def test_mle_new_kats():
    with pytest.raises(orc.OracleError, match="evaluation vec len should equal 2\\^n_vars"):
        orc.mle_new_check(2, 3)
    with pytest.raises(orc.OracleError):
        orc.mle_new_check(2, 2)
    orc.mle_new_check(1, 2)
    orc.mle_new_check(2, 4)


# ---- evaluation_form.rs:128-146 (partial evaluate, single variable) ----
@pytest.mark.parametrize("field", FIELDS)
def test_partial_evaluate_single(field):
    t = F(field, [3, 1, 2, 5])
    assert ints(field, orc.mle_partial_evaluate(field, 2, t, 0, F(field, [5]))) == [-2, 21]
    assert ints(field, orc.mle_partial_evaluate(field, 2, t, 0, F(field, [0]))) == [3, 1]
    # r == 1 shortcut (evaluation_form.rs:62): returns `right`
    assert ints(field, orc.mle_partial_evaluate(field, 2, t, 0, F(field, [1]))) == [2, 5]


# ---- evaluation_form.rs:149-171 (two assignments, initial_var = 1) ----
@pytest.mark.parametrize("field", FIELDS)
def test_partial_evaluate_consecutive(field):
    t = F(field, [0, 0, 0, 3, 0, 0, 2, 5])  # 2ab + 3bc
    assert ints(field, orc.mle_partial_evaluate(field, 3, t, 1, F(field, [2, 3]))) == [18, 22]


# ---- evaluation_form.rs:174-202 (evaluate) ----
@pytest.mark.parametrize("field", FIELDS)
def test_evaluate(field):
    t = F(field, [0, 0, 0, 3, 0, 0, 2, 5])
    assert ints(field, orc.mle_evaluate(field, 3, t, F(field, [2, 3, 4]))) == [48]
    with pytest.raises(orc.OracleError, match="evaluate must assign to all variables"):
        orc.mle_evaluate(field, 3, t, F(field, [2, 3]))


# ---- product_poly.rs:98-121 (constructor) ----
def test_product_new_kats():
    with pytest.raises(orc.OracleError, match="share the same number of variables"):
        orc.product_new_check([2, 1])
    with pytest.raises(orc.OracleError, match="empty polynomials"):
        orc.product_new_check([])
    orc.product_new_check([2, 2, 2])


# ---- product_poly.rs:124-151 (evaluate = product of factor evaluations) ----
@pytest.mark.parametrize("field", FIELDS)
def test_product_evaluate(field):
    tabs = [F(field, [2, 8, 10, 14]), F(field, [2, 8, 10, 22]), F(field, [3, 1, 2, 5])]
    pt = F(field, [1, 10])
    want = 1
    for t in tabs:
        want = want * orc.to_int(field, orc.mle_evaluate(field, 2, t, pt)) % orc.modulus(field)
    assert orc.to_int(field, orc.product_evaluate(field, 2, tabs, pt)) == want
    with pytest.raises(orc.OracleError, match="evaluate must assign to all variables"):
        orc.product_evaluate(field, 2, tabs, F(field, [1]))


# ---- product_poly.rs:154-176 (partial_evaluate is factor-wise) ----
@pytest.mark.parametrize("field", FIELDS)
def test_product_partial_evaluate(field):
    a, b = F(field, [2, 8, 10, 14]), F(field, [2, 8, 10, 22])
    assert ints(field, orc.mle_partial_evaluate(field, 2, a, 1, F(field, [10]))) == [62, 50]
    assert ints(field, orc.mle_partial_evaluate(field, 2, b, 1, F(field, [10]))) == [62, 130]


# ---- product_poly.rs:179-196 (prod_reduce) ----
@pytest.mark.parametrize("field", FIELDS)
def test_prod_reduce(field):
    out = orc.prod_reduce(field, 2, [F(field, [2, 8, 10, 14]), F(field, [2, 8, 10, 22])])
    assert ints(field, out) == [4, 64, 100, 308]


# ---- coefficient_form.rs:1322-1347 pins the MSB-first table order of 2ab + 3bc ----
def test_table_order_2ab_3bc():
    table = [2 * a * b + 3 * b * c for a in (0, 1) for b in (0, 1) for c in (0, 1)]
    assert table == [0, 0, 0, 3, 0, 0, 2, 5]


# ---- coefficient_form.rs:1322-1347 (to_evaluation_form KAT; the reference runs it on a toy field mod 17) ----
@pytest.mark.parametrize("field", FIELDS)
def test_to_evaluation_form_kat(field):
    # p = 2ab + 3bc: selectors [t,t,f] -> key 0b011 = 3, [f,t,t] -> key 0b110 = 6 (selector_to_index :418-430)
    keys, coeffs = [3, 6], F(field, [2, 3])
    assert ints(field, orc.coeff_to_evaluation(field, 3, keys, coeffs)) == [0, 0, 0, 3, 0, 0, 2, 5]
    assert pyref.coeff_to_evaluation(field, 3, {3: 2, 6: 3}) == [0, 0, 0, 3, 0, 0, 2, 5]
    with pytest.raises(orc.OracleError, match="more than specificed number of variables"):
        orc.coeff_to_evaluation(field, 3, [8], F(field, [1]))                   # new_with_coefficient :183-186
    assert orc.coeff_to_evaluation(field, 0, [0], F(field, [7])).shape == (0, 4)  # bit_size 0: the hypercube yields nothing


# ---- sumcheck/src/lib.rs:53-122 (prover <-> verifier; accept / reject only) ----
@pytest.mark.parametrize("field", FIELDS)
def test_sumcheck_correct_sum_multilinear(field):
    t = [F(field, [0, 0, 0, 3, 0, 0, 2, 5])]
    rp, ch = orc.sumcheck_prove(field, 3, t, 1, orc.from_int(field, 10), absorb_table=True)
    assert ints(field, rp[0]) == [3, 7]  # derived round-0 sums
    assert orc.sumcheck_verify(field, 3, t, 1, orc.from_int(field, 10), rp) is True


@pytest.mark.parametrize("field", FIELDS)
def test_sumcheck_correct_sum_deg_2(field):
    t = [F(field, [3, 3, 5, 5]), F(field, [0, 0, 0, 1])]  # (2a+3) * (ab)
    rp, ch = orc.sumcheck_prove(field, 2, t, 2, orc.from_int(field, 5), absorb_table=True)
    assert ints(field, rp[0]) == [0, 5, 14]
    assert orc.sumcheck_verify(field, 2, t, 2, orc.from_int(field, 5), rp) is True


@pytest.mark.parametrize("field", FIELDS)
def test_sumcheck_prove_partial(field):
    t = [F(field, [0, 0, 0, 3, 0, 0, 2, 5])]
    rp, ch = orc.sumcheck_prove(field, 3, t, 1, orc.from_int(field, 10), absorb_table=False)
    sub, vch = orc.sumcheck_verify_partial(field, 1, orc.from_int(field, 10), rp)
    assert np.array_equal(vch, ch)
    assert np.array_equal(orc.product_evaluate(field, 3, t, vch), sub)


@pytest.mark.parametrize("field", FIELDS)
def test_sumcheck_invalid_sum(field):
    t = [F(field, [0, 0, 0, 3, 0, 0, 2, 5])]
    rp, _ = orc.sumcheck_prove(field, 3, t, 1, orc.from_int(field, 12), absorb_table=True)
    with pytest.raises(orc.OracleError, match="claimed_sum != p\\(0\\) \\+ p\\(1\\)"):
        orc.sumcheck_verify(field, 3, t, 1, orc.from_int(field, 12), rp)


def test_sumcheck_verify_round_count(field=orc.BLS12_381_FR):
    t = [F(field, [0, 0, 0, 3, 0, 0, 2, 5])]
    rp, _ = orc.sumcheck_prove(field, 3, t, 1, orc.from_int(field, 10), absorb_table=True)
    with pytest.raises(orc.OracleError, match="require 1 round poly"):
        orc.sumcheck_verify(field, 3, t, 1, orc.from_int(field, 10), rp[:2])


# ---- fft/src/lib.rs:78-82 (round trip on BLS12-377 Fr) ----
@pytest.mark.parametrize("field", [orc.BLS12_377_FR, orc.BN254_FR, orc.BLS12_381_FR])
def test_fft_roundtrip(field):
    a = F(field, [0, 2, 34, 3434])
    assert np.array_equal(orc.ifft(field, orc.fft(field, a)), a)


def test_fft_panics_are_errors():
    f = orc.BLS12_377_FR
    with pytest.raises(orc.OracleError):
        orc.fft(f, F(f, [1, 2, 3]))  # get_root_of_unity(3) is None -> unwrap panics (fft/src/lib.rs:6)
    with pytest.raises(orc.OracleError):
        orc.fft(orc.BN254_FR, np.zeros((0, 4), dtype=np.uint64))


# ---- sha3::Keccak256 public vectors ----
def test_keccak256_public_vectors():
    kat = {
        b"": "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470",
        b"abc": "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45",
    }
    for msg, want in kat.items():
        assert orc.keccak256(msg).hex() == want
        assert pyref.keccak256(msg).hex() == want
    # block-boundary lengths: C oracle vs independent Python model
    for n in (1, 31, 32, 135, 136, 137, 271, 272, 273, 1000):
        msg = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert orc.keccak256(msg) == pyref.keccak256(msg)


@pytest.mark.parametrize("field", [0, 1, 2])
@pytest.mark.parametrize("n,k,D", [(1, 1, 1), (5, 1, 1), (8, 3, 3), (10, 2, 2), (9, 2, 4)])
def test_bench_fused_parallel_prover_equals_the_restatement(field, n, k, D):
    """bench.py's "optimised CPU" prover row (fused rounds, OpenMP) must produce the restatement's proof bit for bit"""
    tabs = [orc.fill_random(field, 300 + i, 1 << n) for i in range(k)]
    s = orc.fill_random(field, 5, 1)[0]
    want_rp, want_ch = orc.sumcheck_prove(field, n, tabs, D, s, False)
    for threads in (1, 3):
        rp, ch, used = orc.sumcheck_prove_fused_parallel(field, n, tabs, D, s, threads)
        assert used >= 1 and np.array_equal(rp, want_rp) and np.array_equal(ch, want_ch)


# ---- published third-party constants (the only external pin this image allows: no Rust toolchain, ark-ff is not vendored) ----
# Literals as the field definitions publish them -- ark-bn254 / ark-bls12-381 / ark-bls12-377 `FrConfig` (MODULUS, GENERATOR,
# TWO_ADICITY, TWO_ADIC_ROOT_OF_UNITY as decimal strings), zkcrypto `bls12_381::Scalar` (INV, R, R2, ROOT_OF_UNITY) and
# halo2curves / `bn` `Fr` (INV, R, R2) -- typed here, NOT computed: the oracle derives every one of them at run time from
# (p, generator, two-adicity) alone (oracle/zk_oracle.c field_get), so agreement is a check of that derivation and of the
# Montgomery core against numbers the libraries the reference links carry.  Used by fft/src/lib.rs:6,14 (get_root_of_unity),
# transcript/src/lib.rs:27-30 (from_be_bytes_mod_order) and every field op of the path.
PUBLISHED = {
    orc.BN254_FR: dict(
        p=21888242871839275222246405745257275088548364400416034343698204186575808495617, generator=5, two_adicity=28,
        root=19103219067921713944291392827692070036145651957329286315305642004821462161904,
        inv=0xc2e1f593efffffff,
        R=0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb,
        R2=0x0216d0b17f4e44a58c49833d53bb808553fe3ab1e35c59e31bb8e645ae216da7),
    orc.BLS12_381_FR: dict(
        p=52435875175126190479447740508185965837690552500527637822603658699938581184513, generator=7, two_adicity=32,
        root=10238227357739495823651030575849232062558860180284477541189508159991286009131,
        inv=0xfffffffeffffffff,
        R=0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe,
        R2=0x0748d9d99f59ff1105d314967254398f2b6cedcb87925c23c999e990f3f29c6d),
    orc.BLS12_377_FR: dict(
        p=8444461749428370424248824938781546531375899335154063827935233455917409239041, generator=22, two_adicity=47,
        root=8065159656716812877374967518403273466521432693661810619979959746626482506078,
        inv=0x0a117fffffffffff,
        R=0x0d4bda322bbb9a9d16d81575512c0fee7257f50f6ffffff27d1c7ffffffffff3,
        R2=0x011fdae7eff1c939a7cc008fe5dc8593cc2c27b58860591f25d577bab861857b),
}


@pytest.mark.parametrize("field", FIELDS)
def test_published_field_constants(field):
    c = PUBLISHED[field]
    assert orc.modulus(field) == c["p"] and orc.two_adicity(field) == c["two_adicity"]
    inv, r1, r2, root = orc.field_constants(field)
    assert inv == c["inv"] and r1 == c["R"] and r2 == c["R2"] and root == c["root"]
    # F::one() in memory is R mod p (the layout that crosses the C ABI), and the independent big-int model agrees
    assert sum(int(v) << (64 * i) for i, v in enumerate(orc.from_int(field, 1))) == c["R"]
    assert pow(c["generator"], (c["p"] - 1) >> c["two_adicity"], c["p"]) == c["root"]


@pytest.mark.parametrize("field", FIELDS)
def test_root_of_unity_is_the_published_root_squared_down(field):
    """F::get_root_of_unity(2^k) = TWO_ADIC_ROOT_OF_UNITY^(2^(s-k)) for every k <= s, None above (fft/src/lib.rs:6,14)"""
    c = PUBLISHED[field]
    p, s = c["p"], c["two_adicity"]
    for k in range(s + 1):
        w = orc.to_int(field, orc.root_of_unity(field, 1 << k))
        assert w == pow(c["root"], 1 << (s - k), p), k
        assert pow(w, 1 << k, p) == 1 and (k == 0 or pow(w, 1 << (k - 1), p) == p - 1)
    with pytest.raises(orc.OracleError):
        orc.root_of_unity(field, 1 << (s + 1))


def test_bn254_root_of_unity_2p24_literal():
    """the omega the 2^24-point NTT of config 5 uses: published root squared four times"""
    p = PUBLISHED[orc.BN254_FR]["p"]
    w24 = pow(PUBLISHED[orc.BN254_FR]["root"], 16, p)
    assert orc.to_int(orc.BN254_FR, orc.root_of_unity(orc.BN254_FR, 1 << 24)) == w24
    assert pow(w24, 1 << 23, p) == p - 1


@pytest.mark.parametrize("field", FIELDS)
def test_from_be_bytes_mod_order_reduces_values_at_and_above_p(field):
    """transcript/src/lib.rs:27-30: a 32-byte digest is an integer below 2^256 that may exceed p (up to 5.2 p on BN254): the
    challenge is that integer mod p; to_bytes_be gives the canonical integer back"""
    p = PUBLISHED[field]["p"]
    cases = [p, p + 1, 2 * p - 1, 2 * p, (1 << 256) - 1, (1 << 255) + 12345, p - 1, 0, 1, ((1 << 256) - 1) // p * p, ((1 << 256) - 1) // p * p - 1]
    for v in cases:
        got = orc.from_be_bytes_mod_order(field, v.to_bytes(32, "big"))
        assert orc.to_int(field, got) == v % p, hex(v)
        assert orc.to_bytes_be(field, got) == (v % p).to_bytes(32, "big")
    # a Keccak-256 digest of a public vector as the challenge source (sha3 KAT above)
    d = orc.keccak256(b"abc")
    assert d.hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert orc.to_int(field, orc.from_be_bytes_mod_order(field, d)) == int.from_bytes(d, "big") % p
    # longer and shorter inputs (ark's from_be_bytes_mod_order takes any length)
    for b in (b"\x01", bytes(range(1, 41)), b"\xff" * 64):
        assert orc.to_int(field, orc.from_be_bytes_mod_order(field, b)) == int.from_bytes(b, "big") % p
