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
