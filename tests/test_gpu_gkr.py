"""GPU: sum-of-products prove_partial, eq tables and the GKR-shaped driver against the big-int model (oracle/gkr_ref.py),
bit for bit, and through the verifier.  SURVEY 8 f3: the reference has no gkr crate -- parity here is definitional
(the model is the definition); the one-term case is pinned to the restated prove_partial (prover.rs:24-30)."""
import random

import numpy as np
import pytest

import zk_amd
from oracle import binding as orc
from oracle import gkr_ref, pyref
from zk_amd import MultiLinearPolynomial as MLE
from zk_amd import ZkError, gkr

pytestmark = pytest.mark.gpu

FIELDS = [zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR]
_ctx = {}


def ctx_for(field):
    if field not in _ctx:
        _ctx[field] = zk_amd.Context(field, 0)
    return _ctx[field]


def F(field, vals):
    return zk_amd.fe_from_ints(field, vals)


def I(field, arr):
    return zk_amd.fe_to_ints(field, np.asarray(arr).reshape(-1, 4))


def rand_circuit(rng, logs):
    layers = []
    for i in range(len(logs) - 1):
        n, n_in = 1 << logs[i], 1 << logs[i + 1]
        layers.append((logs[i], logs[i + 1], [rng.randrange(2) for _ in range(n)], [rng.randrange(n_in) for _ in range(n)],
                       [rng.randrange(n_in) for _ in range(n)]))
    return layers


def upload_circuit(c, layers):
    circ = gkr.Circuit(c)
    for lo, li, op, left, right in layers:
        circ.add_layer(lo, li, op, left, right)
    return circ


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("m", [0, 1, 2, 5, 9, 12])
def test_eq_table_vs_model(field, m):
    p = zk_amd.modulus(field)
    rng = random.Random(100 + m)
    pt = [rng.randrange(p) for _ in range(m)]
    got = gkr.eq_table(ctx_for(field), F(field, pt) if m else [])
    assert I(field, got.evaluation_slice()) == gkr_ref.eq_table(field, pt)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("shape,D", [([2, 1], 2), ([1], 1), ([2], 2), ([3], 3), ([1, 1], 1), ([2, 2], 2), ([3, 1], 3),
                                     ([2, 1, 1], 2), ([3, 2, 1, 1], 3), ([4, 2], 4)])
@pytest.mark.parametrize("n", [1, 2, 5, 10, 12])
def test_prove_terms_vs_model(field, shape, D, n):
    c = ctx_for(field)
    p = zk_amd.modulus(field)
    rng = random.Random(n * 1000 + sum(shape) * 10 + D)
    tabs = [[[rng.randrange(p) for _ in range(1 << n)] for _ in range(k)] for k in shape]
    s = 0
    for term in tabs:
        for j in range(1 << n):
            prod = 1
            for t in term:
                prod = prod * t[j] % p
            s += prod
    s %= p
    want_rp, want_ch, want_fin = gkr_ref.prove_partial_terms(field, tabs, D, s)
    poly = gkr.SumOfProductsPoly([[MLE.new(c, n, F(field, t)) for t in term] for term in tabs])
    rp, ch, fin = gkr.prove_partial_terms(poly, D, zk_amd.fe_from_int(field, s))
    assert [I(field, r) for r in rp] == want_rp
    assert I(field, ch) == want_ch
    assert I(field, fin) == want_fin
    # the tables were not consumed
    assert I(field, poly.terms[0][0].evaluation_slice()) == tabs[0][0]
    # and the proof verifies as a plain sumcheck (verifier.rs:38-41) down to the factors' values
    sub = zk_amd.SumcheckVerifier.verify_partial(field, zk_amd.SumcheckProof(zk_amd.fe_from_int(field, s), rp))
    acc, pos = 0, 0
    for k in shape:
        prod = 1
        for v in want_fin[pos:pos + k]:
            prod = prod * v % p
        acc, pos = acc + prod, pos + k
    assert zk_amd.fe_to_int(field, sub.sum) == acc % p


def test_prove_terms_argument_errors():
    field = zk_amd.BN254_FR
    c = ctx_for(field)
    a, b = MLE.random(c, 4, 1), MLE.random(c, 4, 2)
    other = MLE.random(c, 3, 3)
    s = zk_amd.fe_from_int(field, 0)
    with pytest.raises(ZkError):   # the same table twice
        gkr.prove_partial_terms(gkr.SumOfProductsPoly([[a, b], [a]]), 2, s)
    with pytest.raises(ZkError, match="same number of variables"):   # product_poly.rs:25
        gkr.prove_partial_terms(gkr.SumOfProductsPoly([[a, b], [other]]), 2, s)
    with pytest.raises(ZkError):   # a term of 3 factors has degree 3 > D
        gkr.prove_partial_terms(gkr.SumOfProductsPoly([[a, b, MLE.random(c, 4, 5)]]), 2, s)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("logs", [[0, 1], [2, 3, 2], [3, 2, 4, 3], [1, 1, 1, 1, 1], [6, 7, 5], [0, 10, 11], [4, 13, 12]])
def test_gkr_vs_model(field, logs):
    c = ctx_for(field)
    p = zk_amd.modulus(field)
    rng = random.Random(sum(logs) * 3 + field)
    layers = rand_circuit(rng, logs)
    inputs = [rng.randrange(p) for _ in range(1 << logs[-1])]
    seed = bytes(rng.randrange(256) for _ in range(32))
    want_out, want_proof = gkr_ref.gkr_prove(field, layers, inputs, seed)
    circ = upload_circuit(c, layers)
    assert circ.depth() == len(layers) and circ.layer_dims(0) == (logs[0], logs[1])
    x = MLE.new(c, logs[-1], F(field, inputs))
    assert I(field, circ.evaluate(x).evaluation_slice()) == want_out
    out, proof = gkr.gkr_prove(circ, x, seed)
    assert I(field, out.evaluation_slice()) == want_out
    assert I(field, proof) == want_proof                      # bit-exact against the definition
    assert gkr.gkr_verify(circ, x, out, seed, proof)
    assert gkr_ref.gkr_verify(field, layers, inputs, want_out, seed, I(field, proof))
    # tampering: proof elements, outputs, inputs, seed
    for pos in sorted(set([0, 1, len(want_proof) // 2, len(want_proof) - 2, len(want_proof) - 1])):
        bad = proof.copy()
        bad[pos] = zk_amd.fe_from_int(field, (want_proof[pos] + 1) % p)
        assert not gkr.gkr_verify(circ, x, out, seed, bad)
    bad_out = list(want_out)
    bad_out[-1] = (bad_out[-1] + 1) % p
    assert not gkr.gkr_verify(circ, x, MLE.new(c, logs[0], F(field, bad_out)), seed, proof)
    bad_in = list(inputs)
    bad_in[0] = (bad_in[0] + 1) % p
    assert not gkr.gkr_verify(circ, MLE.new(c, logs[-1], F(field, bad_in)), out, seed, proof)
    if logs[0]:
        assert not gkr.gkr_verify(circ, x, out, bytes(32), proof)


def test_gkr_known_small_circuit():
    """(a+b)*(c*d) on 1,2,3,4 = 36"""
    field = zk_amd.BLS12_381_FR
    c = ctx_for(field)
    circ = upload_circuit(c, [(0, 1, [1], [0], [1]), (1, 2, [0, 1], [0, 2], [1, 3])])
    x = MLE.new(c, 2, F(field, [1, 2, 3, 4]))
    out, proof = gkr.gkr_prove(circ, x, bytes(32))
    assert I(field, out.evaluation_slice()) == [36]
    assert gkr.gkr_verify(circ, x, out, bytes(32), proof)


def test_circuit_argument_errors():
    c = ctx_for(zk_amd.BN254_FR)
    circ = gkr.Circuit(c)
    with pytest.raises(ZkError):   # gate input out of range
        circ.add_layer(1, 1, [0, 1], [0, 2], [1, 1])
    with pytest.raises(ZkError):   # unknown op
        circ.add_layer(1, 1, [0, 2], [0, 1], [1, 1])
    circ.add_layer(1, 2, [0, 1], [0, 2], [1, 3])
    with pytest.raises(ZkError, match="same number of variables"):   # layer sizes must chain
        circ.add_layer(3, 2, [0] * 8, [0] * 8, [0] * 8)
    with pytest.raises(ZkError, match="same number of variables"):   # input arity
        gkr.gkr_prove(circ, MLE.random(c, 3, 1), bytes(32))


def test_gkr_width_2p16_depth4_verifies():
    """a circuit too large for the big-int model: accept an honest proof, reject a tampered one"""
    field = zk_amd.BN254_FR
    c = ctx_for(field)
    rng = np.random.default_rng(5)
    w, depth = 16, 4
    circ = gkr.Circuit(c)
    for _ in range(depth):
        circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32),
                       rng.integers(0, 1 << w, 1 << w, dtype=np.uint32))
    x = MLE.random(c, w, 77)
    seed = bytes(range(32))
    out, proof = gkr.gkr_prove(circ, x, seed)
    assert gkr.gkr_verify(circ, x, out, seed, proof)
    bad = proof.copy()
    bad[len(bad) // 3, 0] ^= np.uint64(1)
    assert not gkr.gkr_verify(circ, x, out, seed, bad)


@pytest.mark.parametrize("shape,D", [([2, 1], 3), ([1, 2], 2), ([1, 1, 1, 1], 1), ([2, 1], 4)])
def test_prove_terms_consume_and_fallback_shapes(shape, D):
    """terms {k, 1} with (k, D) outside the merged kernel's shapes take the term-by-term launches; consume = True may
    fold the caller's tables in place but must produce the same proof"""
    field = zk_amd.BN254_FR
    c = ctx_for(field)
    p = zk_amd.modulus(field)
    rng = random.Random(sum(shape) * 7 + D)
    n = 11
    tabs = [[[rng.randrange(p) for _ in range(1 << n)] for _ in range(k)] for k in shape]
    s = 0
    for term in tabs:
        for j in range(1 << n):
            prod = 1
            for t in term:
                prod = prod * t[j] % p
            s += prod
    s %= p
    want_rp, want_ch, want_fin = gkr_ref.prove_partial_terms(field, tabs, D, s)
    for consume in (False, True):
        poly = gkr.SumOfProductsPoly([[MLE.new(c, n, F(field, t)) for t in term] for term in tabs])
        rp, ch, fin = gkr.prove_partial_terms(poly, D, zk_amd.fe_from_int(field, s), consume=consume)
        assert [I(field, r) for r in rp] == want_rp and I(field, ch) == want_ch and I(field, fin) == want_fin


@pytest.mark.parametrize("s", [5, 9])
@pytest.mark.parametrize("field", FIELDS)
def test_gkr_degenerate_wiring(field, s):
    """extreme fan-out (every gate reads input 0 on the left: one CSR row holds all gates, the others are empty), all-add
    and all-mul layers, a zero input table -- against the model and through the verifier"""
    c = ctx_for(field)
    p = zk_amd.modulus(field)
    rng = random.Random(99 + field)
    n = 1 << s   # s = 9: rows of 512 entries go through the one-workgroup-per-row kernels (kGkrHeavyRow = 256)
    layers = [(s, s, [rng.randrange(2) for _ in range(n)], [0] * n, [rng.randrange(n) for _ in range(n)]),
              (s, s, [0] * n, [rng.randrange(n) for _ in range(n)], [n - 1] * n),
              (s, s, [1] * n, list(range(n)), list(range(n)))]
    for inputs in ([rng.randrange(p) for _ in range(n)], [0] * n):
        want_out, want_proof = gkr_ref.gkr_prove(field, layers, inputs, bytes(32))
        circ = upload_circuit(c, layers)
        x = MLE.new(c, s, F(field, inputs))
        out, proof = gkr.gkr_prove(circ, x, bytes(32))
        assert I(field, out.evaluation_slice()) == want_out and I(field, proof) == want_proof
        assert gkr.gkr_verify(circ, x, out, bytes(32), proof)


def test_gkr_heavy_fanout_width_2p18():
    """one input wire feeding every gate of a 2^18-wide layer must not serialise on one thread"""
    import time

    field = zk_amd.BN254_FR
    c = ctx_for(field)
    rng = np.random.default_rng(11)
    w = 18
    circ = gkr.Circuit(c)
    circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), np.zeros(1 << w, dtype=np.uint32),
                   rng.integers(0, 1 << w, 1 << w, dtype=np.uint32))
    circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32),
                   np.full(1 << w, 5, dtype=np.uint32))
    x = MLE.random(c, w, 3)
    out, proof = gkr.gkr_prove(circ, x, bytes(32))
    t0 = time.perf_counter()
    out, proof = gkr.gkr_prove(circ, x, bytes(32))
    dt = time.perf_counter() - t0
    assert gkr.gkr_verify(circ, x, out, bytes(32), proof)
    assert dt < 0.05, f"heavy fan-out proof took {dt * 1e3:.1f} ms"


@pytest.mark.parametrize("field", FIELDS)
def test_gkr_statement_is_bound_before_the_output_point(field):
    """the output point used to depend on the caller's seed alone, so outputs + delta with delta~(g) = 0 verified
    with the honest proof.  Circuit, inputs and outputs are now digested into the transcript first: the forged outputs, a
    flipped gate and a rewired gate must all be rejected by zk_gkr_verify (and the digests match the model's: the proofs of
    test_gkr_vs_model are bit-exact against gkr_ref, whose transcript starts with them)."""
    c = ctx_for(field)
    p = zk_amd.modulus(field)
    rng = random.Random(99 + field)
    logs = [2, 3, 2]
    layers = rand_circuit(rng, logs)
    inputs = [rng.randrange(p) for _ in range(1 << logs[-1])]
    seed = bytes(32)
    circ = upload_circuit(c, layers)
    x = MLE.new(c, logs[-1], F(field, inputs))
    out, proof = gkr.gkr_prove(circ, x, seed)
    assert gkr.gkr_verify(circ, x, out, seed, proof)
    # delta orthogonal to eq(g_old, .), g_old = the point the seed alone would have given
    from oracle import pyref
    tr = pyref.Transcript()
    tr.append(seed)
    g_old = [tr.sample_field_element(field) for _ in range(logs[0])]
    eq = gkr_ref.eq_table(field, g_old)
    delta = [eq[1], (-eq[0]) % p, 0, 0]
    forged = [(o + d) % p for o, d in zip(I(field, out.evaluation_slice()), delta)]
    assert not gkr.gkr_verify(circ, x, MLE.new(c, logs[0], F(field, forged)), seed, proof)
    lo, li, op, left, right = layers[1]
    flipped = list(layers)
    flipped[1] = (lo, li, [1 - op[0]] + list(op[1:]), left, right)
    assert not gkr.gkr_verify(upload_circuit(c, flipped), x, out, seed, proof)
    rewired = list(layers)
    rewired[1] = (lo, li, op, [(left[0] + 1) % (1 << li)] + list(left[1:]), right)
    assert not gkr.gkr_verify(upload_circuit(c, rewired), x, out, seed, proof)


def test_config4_gkr_depth8_width_2p20_prove_verify_tamper():
    """BASELINE config[3] at its own size: depth 8, width 2^20, random add/mul gates with random wiring.  The proof must
    verify; one tampered proof element (first, middle, last), a tampered output and a tampered input must each be rejected."""
    field = zk_amd.BN254_FR
    c = ctx_for(field)
    rng = np.random.default_rng(0x6B72)
    w = 20
    circ = gkr.Circuit(c)
    for _ in range(8):
        circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32),
                       rng.integers(0, 1 << w, 1 << w, dtype=np.uint32))
    x = MLE.random(c, w, 0x6B72, 0)
    seed = bytes(range(32))
    out, proof = gkr.gkr_prove(circ, x, seed)
    assert gkr.gkr_verify(circ, x, out, seed, proof)
    one = zk_amd.fe_from_int(field, 1)
    flat = proof.reshape(-1, 4)
    for idx in (0, flat.shape[0] // 2, flat.shape[0] - 1):
        bad = proof.copy()
        bad.reshape(-1, 4)[idx] = orc.add(field, flat[idx], one)
        assert not gkr.gkr_verify(circ, x, out, seed, bad), f"tampered proof element {idx} accepted"
    outs = out.evaluation_slice().copy()
    outs[12345] = orc.add(field, outs[12345], one)
    assert not gkr.gkr_verify(circ, x, MLE.new(c, w, outs), seed, proof)
    xs = x.evaluation_slice().copy()
    xs[54321] = orc.add(field, xs[54321], one)
    assert not gkr.gkr_verify(circ, MLE.new(c, w, xs), out, seed, proof)
    assert not gkr.gkr_verify(circ, x, out, bytes(32), proof)   # another seed: another transcript
    circ.free()


def test_config4_full_size_proof_checked_with_oracle_primitives_only():
    """BASELINE config[3] at its own size, checked by something that is NOT the library: the depth-8 x 2^20 GPU proof goes through
    tests/gkr_oracle_check.py -- every layer's values recomputed on the CPU (OpenMP), the output table compared, the statement
    digests and the transcript replayed on orc.Transcript, the round checks made by verify_internal on that transcript
    (sumcheck/src/verifier.rs:44-78), the wiring predicates summed over the gate lists on the CPU, and W(u), W(v) of EVERY layer
    compared with orc.mle_evaluate of the CPU's own layer values at the replayed points (evaluation_form.rs:83-89, the intent at
    :45-48).  The library's verifier shares k_eq_split / k_gkr_wiring_eval with its prover; this check shares nothing with either.
    One flipped proof element must be rejected by it as well."""
    import time

    from tests.gkr_oracle_check import check_proof

    field = zk_amd.BN254_FR
    c = ctx_for(field)
    rng = np.random.default_rng(0x6B73)
    w = 20
    circ = gkr.Circuit(c)
    layers = []
    for _ in range(8):
        op = rng.integers(0, 2, 1 << w, dtype=np.uint8)
        left = rng.integers(0, 1 << w, 1 << w, dtype=np.uint32)
        right = rng.integers(0, 1 << w, 1 << w, dtype=np.uint32)
        circ.add_layer(w, w, op, left, right)
        layers.append((w, w, op, left, right))
    x = MLE.random(c, w, 0x6B73, 0)
    seed = bytes(range(1, 33))
    out, proof = gkr.gkr_prove(circ, x, seed)
    t0 = time.time()
    ok, why = check_proof(field, layers, x.evaluation_slice(), out.evaluation_slice(), seed, proof)
    dt = time.time() - t0
    assert ok, why
    bad = proof.copy()
    bad.reshape(-1, 4)[bad.reshape(-1, 4).shape[0] // 3] = orc.add(field, bad.reshape(-1, 4)[bad.reshape(-1, 4).shape[0] // 3],
                                                                  zk_amd.fe_from_int(field, 1))
    assert not check_proof(field, layers, x.evaluation_slice(), out.evaluation_slice(), seed, bad)[0]
    print(f"oracle-only check of the depth-8 x 2^20 proof: {dt:.1f} s")
    circ.free()
