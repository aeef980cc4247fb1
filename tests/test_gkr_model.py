"""CPU checks of the GKR-shaped driver's definition (oracle/gkr_ref.py; SURVEY 8 f3: no reference crate, so the model is
the definition -- "parity definitional").  What CAN be pinned to the reference is pinned here: with ONE term the
sum-of-products prover must reproduce the restated prove_partial (sumcheck/src/prover.rs:24-30) exactly."""
import random

import pytest

from oracle import gkr_ref, pyref

FIELDS = [0, 1, 2]


def rand_circuit(rng, logs):
    """logs = [log_out(0), log_out(1), ..., log_in(last)] -> layers"""
    layers = []
    for i in range(len(logs) - 1):
        n, n_in = 1 << logs[i], 1 << logs[i + 1]
        layers.append((logs[i], logs[i + 1], [rng.randrange(2) for _ in range(n)], [rng.randrange(n_in) for _ in range(n)],
                       [rng.randrange(n_in) for _ in range(n)]))
    return layers


@pytest.mark.parametrize("field", FIELDS)
def test_eq_table_definition(field):
    p = pyref.modulus(field)
    rng = random.Random(field)
    pt = [rng.randrange(p) for _ in range(4)]
    t = gkr_ref.eq_table(field, pt)
    assert sum(t) % p == 1
    for idx in range(16):
        want = 1
        for v in range(4):
            bit = (idx >> (3 - v)) & 1          # variable 0 = index MSB (pairing_index.rs:61-65)
            want = want * (pt[v] if bit else 1 - pt[v]) % p
        assert t[idx] == want
    # eq(point, .) is the table whose MLE evaluates to 1 at `point`... of the indicator: sum_x eq(pt,x) f(x) = f~(pt)
    f = [rng.randrange(p) for _ in range(16)]
    assert sum(a * b for a, b in zip(t, f)) % p == pyref.MLE(field, 4, f).evaluate(pt)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,D", [(1, 1), (2, 2), (3, 3), (2, 3)])
def test_one_term_is_prove_partial(field, k, D):
    p = pyref.modulus(field)
    rng = random.Random(10 * k + D)
    n = 4
    tabs = [[rng.randrange(p) for _ in range(1 << n)] for _ in range(k)]
    prod = pyref.Product([pyref.MLE(field, n, t) for t in tabs])
    s = sum(prod.prod_reduce()) % p
    want_rp, want_ch = pyref.sumcheck_prove(prod, s, D, False)
    rp, ch, fin = gkr_ref.prove_partial_terms(field, [tabs], D, s)
    assert rp == want_rp and ch == want_ch
    assert fin == [pyref.MLE(field, n, t).evaluate(ch) for t in tabs]


@pytest.mark.parametrize("field", FIELDS)
def test_sum_of_products_verifies(field):
    p = pyref.modulus(field)
    rng = random.Random(7)
    n = 3
    a, b, c = ([rng.randrange(p) for _ in range(1 << n)] for _ in range(3))
    s = sum(x * y + z for x, y, z in zip(a, b, c)) % p
    rp, ch, fin = gkr_ref.prove_partial_terms(field, [[a, b], [c]], 2, s)
    sub, ch2 = pyref.sumcheck_verify_partial(field, s, rp)
    assert ch2 == ch
    assert sub == (fin[0] * fin[1] + fin[2]) % p


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("logs", [[0, 1], [2, 3, 2], [3, 2, 4, 3], [1, 1, 1, 1, 1]])
def test_gkr_model_accepts_and_rejects(field, logs):
    p = pyref.modulus(field)
    rng = random.Random(sum(logs) + field)
    layers = rand_circuit(rng, logs)
    inputs = [rng.randrange(p) for _ in range(1 << logs[-1])]
    seed = bytes(rng.randrange(256) for _ in range(32))
    outputs, proof = gkr_ref.gkr_prove(field, layers, inputs, seed)
    assert outputs == gkr_ref.evaluate_circuit(field, layers, inputs)[0]
    assert len(proof) == sum(6 * l[1] + 2 for l in layers)
    assert gkr_ref.gkr_verify(field, layers, inputs, outputs, seed, proof)
    # any single corrupted proof element, a wrong output, a wrong input or another seed must be rejected
    for pos in range(len(proof)):
        bad = list(proof)
        bad[pos] = (bad[pos] + 1) % p
        assert not gkr_ref.gkr_verify(field, layers, inputs, outputs, seed, bad)
    bad_out = list(outputs)
    bad_out[0] = (bad_out[0] + 1) % p
    assert not gkr_ref.gkr_verify(field, layers, inputs, bad_out, seed, proof)
    bad_in = list(inputs)
    bad_in[-1] = (bad_in[-1] + 1) % p
    assert not gkr_ref.gkr_verify(field, layers, bad_in, outputs, seed, proof)
    if logs[0] > 0:
        assert not gkr_ref.gkr_verify(field, layers, inputs, outputs, bytes(32), proof)


def test_known_small_circuit():
    """(a+b)*(c*d) with inputs 1,2,3,4 -> 36; layer 1: [a+b, c*d], layer 0: [mul]  (log_out 0 needs no point)."""
    field = 1
    layers = [(0, 1, [1], [0], [1]), (1, 2, [0, 1], [0, 2], [1, 3])]
    vals = gkr_ref.evaluate_circuit(field, layers, [1, 2, 3, 4])
    assert vals[1] == [3, 12] and vals[0] == [36]
    outputs, proof = gkr_ref.gkr_prove(field, layers, [1, 2, 3, 4], bytes(32))
    assert outputs == [36] and gkr_ref.gkr_verify(field, layers, [1, 2, 3, 4], outputs, bytes(32), proof)


@pytest.mark.parametrize("field", FIELDS)
def test_statement_is_bound_before_the_output_point(field):
    """with the output point g drawn from the caller's seed alone, outputs + delta with delta~(g) = 0 verified
    with the honest proof.  The transcript now absorbs digests of circuit, inputs and OUTPUTS before g: the forged outputs
    move g and are rejected; so are a changed gate and a changed wire."""
    p = pyref.modulus(field)
    rng = random.Random(99 + field)
    logs = [2, 3, 2]
    layers = rand_circuit(rng, logs)
    inputs = [rng.randrange(p) for _ in range(1 << logs[-1])]
    seed = bytes(32)
    outputs, proof = gkr_ref.gkr_prove(field, layers, inputs, seed)
    assert gkr_ref.gkr_verify(field, layers, inputs, outputs, seed, proof)
    # delta with MLE zero at the point the OLD protocol would have used (seed only): eq(g, .)-orthogonal vector
    tr = pyref.Transcript()
    tr.append(seed)
    g_old = [tr.sample_field_element(field) for _ in range(logs[0])]
    eq = gkr_ref.eq_table(field, g_old)
    delta = [eq[1], (-eq[0]) % p, 0, 0]                     # sum_x eq(g,x) delta[x] = 0
    assert sum(a * b for a, b in zip(eq, delta)) % p == 0 and any(delta)
    forged = [(o + d) % p for o, d in zip(outputs, delta)]
    assert not gkr_ref.gkr_verify(field, layers, inputs, forged, seed, proof)
    lo, li, op, left, right = layers[1]
    flipped = list(layers)
    flipped[1] = (lo, li, [1 - op[0]] + list(op[1:]), left, right)
    assert not gkr_ref.gkr_verify(field, flipped, inputs, outputs, seed, proof)
    rewired = list(layers)
    rewired[1] = (lo, li, op, [(left[0] + 1) % (1 << li)] + list(left[1:]), right)
    assert not gkr_ref.gkr_verify(field, rewired, inputs, outputs, seed, proof)


def test_tree_digest_shape():
    """leaves of 128 bytes, 4-ary nodes; pinned against direct Keccak compositions"""
    k = pyref.keccak256
    assert gkr_ref.tree_digest(b"") == k(b"")
    assert gkr_ref.tree_digest(b"a" * 100) == k(b"a" * 100)
    d = bytes(range(256)) * 3                                # 768 bytes = 6 leaves -> 2 nodes -> root
    leaves = [k(d[i:i + 128]) for i in range(0, 768, 128)]
    assert gkr_ref.tree_digest(d) == k(k(b"".join(leaves[:4])) + k(b"".join(leaves[4:])))
    assert gkr_ref.tree_digest(d[:129]) == k(k(d[:128]) + k(d[128:129]))


# ---- the oracle-primitives-only checker (tests/gkr_oracle_check.py) against the big-int model -------------------------------
def _np_layers(layers):
    import numpy as np

    return [(lo, li, np.array(op, dtype=np.uint8), np.array(left, dtype=np.uint32), np.array(right, dtype=np.uint32))
            for lo, li, op, left, right in layers]


@pytest.mark.parametrize("field", FIELDS)
def test_oracle_gkr_pieces_match_the_model(field):
    """orc.tree_digest / eq_table / circuit_layer / gkr_wiring_sums / sumcheck_verify_partial_lengths_on are the C forms of
    gkr_ref's tree_digest / eq_table / evaluate_circuit / wiring sums / _verify_partial_from: same values on small inputs."""
    import numpy as np

    from oracle import binding as orc
    from tests.gkr_oracle_check import circuit_digest

    p = pyref.modulus(field)
    rng = random.Random(100 + field)
    for ln in (0, 1, 31, 128, 129, 512, 513, 4096, 5000):
        data = bytes(rng.randrange(256) for _ in range(ln))
        assert orc.tree_digest(data) == gkr_ref.tree_digest(data), ln
    pt = [rng.randrange(p) for _ in range(5)]
    assert orc.to_ints(field, orc.eq_table(field, orc.from_ints(field, pt))) == gkr_ref.eq_table(field, pt)
    layers = rand_circuit(rng, [2, 4, 3])
    ins = [rng.randrange(p) for _ in range(8)]
    vals = gkr_ref.evaluate_circuit(field, layers, ins)
    w = orc.from_ints(field, ins)
    for i in (1, 0):
        _, _, op, left, right = layers[i]
        w = orc.circuit_layer(field, op, left, right, w)
        assert orc.to_ints(field, w) == vals[i]
    assert circuit_digest(_np_layers(layers)) == gkr_ref.circuit_digest(layers)
    # wiring sums of layer 0 at random points, with a two-point E
    lo, li, op, left, right = layers[0]
    g1, g2 = [rng.randrange(p) for _ in range(lo)], [rng.randrange(p) for _ in range(lo)]
    u, v = [rng.randrange(p) for _ in range(li)], [rng.randrange(p) for _ in range(li)]
    al, be = rng.randrange(p), rng.randrange(p)
    E = gkr_ref._E(field, {"g1": g1, "g2": g2, "alpha": al, "beta": be}, lo)
    equ, eqv = gkr_ref.eq_table(field, u), gkr_ref.eq_table(field, v)
    want_add = sum(E[z] * equ[x] * eqv[y] for z, (o, x, y) in enumerate(zip(op, left, right)) if o == 0) % p
    want_mul = sum(E[z] * equ[x] * eqv[y] for z, (o, x, y) in enumerate(zip(op, left, right)) if o == 1) % p
    F = lambda xs: orc.from_ints(field, xs)
    a, m = orc.gkr_wiring_sums(field, op, left, right, orc.eq_table(field, F(g1)), orc.eq_table(field, F(g2)), orc.from_int(field, al),
                               orc.from_int(field, be), orc.eq_table(field, F(u)), orc.eq_table(field, F(v)))
    assert orc.to_int(field, a) == want_add and orc.to_int(field, m) == want_mul
    # verify_internal on a transcript that already holds bytes == the model's continuation
    tabs = [[rng.randrange(p) for _ in range(8)] for _ in range(2)]
    s = sum(x * y for x, y in zip(*tabs)) % p
    tr_m = pyref.Transcript()
    tr_m.append(b"prefix")
    rp, ch, _ = gkr_ref.prove_partial_terms(field, [tabs], 2, s, tr_m)
    tr_v = pyref.Transcript()
    tr_v.append(b"prefix")
    want_sub, want_ch = gkr_ref._verify_partial_from(tr_v, field, s, rp)
    tr_o = orc.Transcript()
    tr_o.append(b"prefix")
    sub, chs = orc.sumcheck_verify_partial_lengths_on(tr_o, field, orc.from_int(field, s), [F(r) for r in rp])
    assert orc.to_int(field, sub) == want_sub and orc.to_ints(field, chs) == want_ch == ch
    assert orc.to_int(field, tr_o.sample_field_element(field)) == tr_v.sample_field_element(field)   # same state afterwards


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("logs", [[1, 3, 2], [3, 3, 3, 3], [0, 2, 4], [4, 1, 3]])
def test_oracle_checker_accepts_model_proofs_and_rejects_corruptions(field, logs):
    """The checker that validates the depth-8 x 2^20 GPU proof (tests/test_gpu_gkr.py) is itself checked here: it accepts the
    big-int model's proofs and rejects every single-element corruption of proof, outputs or inputs."""
    import numpy as np

    from oracle import binding as orc
    from tests.gkr_oracle_check import check_proof

    p = pyref.modulus(field)
    rng = random.Random(sum(logs) + 31 * field)
    layers = rand_circuit(rng, logs)
    ins = [rng.randrange(p) for _ in range(1 << logs[-1])]
    seed = bytes(rng.randrange(256) for _ in range(32))
    outs, proof = gkr_ref.gkr_prove(field, layers, ins, seed)
    assert gkr_ref.gkr_verify(field, layers, ins, outs, seed, proof)
    L = _np_layers(layers)
    F = lambda xs: orc.from_ints(field, xs)
    ok, why = check_proof(field, L, F(ins), F(outs), seed, F(proof))
    assert ok, why
    for idx in range(len(proof)):
        bad = list(proof)
        bad[idx] = (bad[idx] + 1) % p
        ok, _ = check_proof(field, L, F(ins), F(outs), seed, F(bad))
        assert not ok, f"corrupted proof element {idx} accepted"
    bad_out = list(outs)
    bad_out[0] = (bad_out[0] + 1) % p
    assert not check_proof(field, L, F(ins), F(bad_out), seed, F(proof))[0]
    bad_in = list(ins)
    bad_in[-1] = (bad_in[-1] + 1) % p
    assert not check_proof(field, L, F(bad_in), F(outs), seed, F(proof))[0]
    assert not check_proof(field, L, F(ins), F(outs), bytes(32), F(proof))[0] or seed == bytes(32)
