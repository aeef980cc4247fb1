"""A verifier for the GKR-shaped driver's proofs built from ORACLE primitives only -- nothing from
libzk_amd.so: CPU circuit evaluation, the statement digests, the transcript replay on orc.Transcript, the round checks by
verify_internal on that transcript (orc.sumcheck_verify_partial_lengths_on: sumcheck/src/verifier.rs:44-78), the wiring
predicates, and W(u), W(v) of EVERY layer recomputed with orc.mle_evaluate on the CPU's own layer values
(evaluation_form.rs:83-89).  The protocol is this repository's (the reference has no gkr crate); tests/test_gkr_model.py checks
this checker against the big-int model on small circuits, tests/test_gpu_gkr.py runs it on the depth-8 x 2^20 GPU proof."""
import numpy as np

from oracle import binding as orc


def _be(field, elems):
    e = np.ascontiguousarray(elems, dtype=np.uint64).reshape(-1, 4)
    return orc.mle_to_bytes(field, max(e.shape[0].bit_length() - 1, 0), e) if e.shape[0] & (e.shape[0] - 1) == 0 else \
        b"".join(orc.to_bytes_be(field, x) for x in e)


def circuit_digest(layers):
    h = b""
    for log_out, log_in, op, left, right in layers:
        h += int(log_out).to_bytes(8, "little") + int(log_in).to_bytes(8, "little")
        h += orc.tree_digest(np.ascontiguousarray(op, dtype=np.uint8))
        h += orc.tree_digest(np.ascontiguousarray(left, dtype="<u4"))
        h += orc.tree_digest(np.ascontiguousarray(right, dtype="<u4"))
    return orc.keccak256(h)


def evaluate_circuit(field, layers, inputs):
    vals = [None] * (len(layers) + 1)
    vals[len(layers)] = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 4)
    for i in range(len(layers) - 1, -1, -1):
        _, _, op, left, right = layers[i]
        vals[i] = orc.circuit_layer(field, op, left, right, vals[i + 1])
    return vals


def check_proof(field, layers, inputs, outputs, seed, proof):
    """-> (True, "") or (False, reason).  layers: [(log_out, log_in, op, left, right)], output layer first; inputs / outputs /
    proof: (n, 4) uint64 arrays (ark-ff layout)."""
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 4)
    outputs = np.ascontiguousarray(outputs, dtype=np.uint64).reshape(-1, 4)
    proof = np.ascontiguousarray(proof, dtype=np.uint64).reshape(-1, 4)
    vals = evaluate_circuit(field, layers, inputs)
    if not np.array_equal(vals[0], outputs):
        return False, "outputs differ from the CPU evaluation of the circuit"
    tr = orc.Transcript()
    tr.append(bytes(seed))
    tr.append(circuit_digest(layers))
    tr.append(orc.tree_digest(_be(field, inputs)))
    tr.append(orc.tree_digest(_be(field, outputs)))
    g = np.stack([tr.sample_field_element(field) for _ in range(layers[0][0])]) if layers[0][0] else np.zeros((0, 4), dtype=np.uint64)
    one = orc.from_int(field, 1)
    claim = {"g1": g, "g2": None, "alpha": one, "beta": orc.from_int(field, 0), "c": orc.mle_evaluate(field, layers[0][0], outputs, g)}
    pos = 0
    for i, (log_out, log_in, op, left, right) in enumerate(layers):
        s = log_in
        lp = proof[pos:pos + 6 * s + 2]
        pos += 6 * s + 2
        rp1 = [lp[3 * r:3 * r + 3] for r in range(s)]
        rp2 = [lp[3 * s + 3 * r:3 * s + 3 * r + 3] for r in range(s)]
        wu, wv = lp[-2], lp[-1]
        try:
            sub1, u = orc.sumcheck_verify_partial_lengths_on(tr, field, claim["c"], rp1)
            tr.append(orc.to_bytes_be(field, wu))
            sub2, v = orc.sumcheck_verify_partial_lengths_on(tr, field, sub1, rp2)
            tr.append(orc.to_bytes_be(field, wv))
        except orc.OracleError as e:
            return False, f"layer {i}: round check failed ({e})"
        e1 = orc.eq_table(field, claim["g1"])
        e2 = orc.eq_table(field, claim["g2"]) if claim["g2"] is not None else None
        add_e, mul_e = orc.gkr_wiring_sums(field, op, left, right, e1, e2, claim["alpha"], claim["beta"], orc.eq_table(field, u),
                                           orc.eq_table(field, v))
        want = orc.add(field, orc.mul(field, add_e, orc.add(field, wu, wv)), orc.mul(field, mul_e, orc.mul(field, wu, wv)))
        if not np.array_equal(sub2, want):
            return False, f"layer {i}: wiring check failed"
        # the claimed evaluations against the CPU's own values of layer i + 1
        if not np.array_equal(wu, orc.mle_evaluate(field, log_in, vals[i + 1], u)):
            return False, f"layer {i}: W(u) is not the MLE of the CPU layer at u"
        if not np.array_equal(wv, orc.mle_evaluate(field, log_in, vals[i + 1], v)):
            return False, f"layer {i}: W(v) is not the MLE of the CPU layer at v"
        alpha = tr.sample_field_element(field)
        beta = tr.sample_field_element(field)
        claim = {"g1": u, "g2": v, "alpha": alpha, "beta": beta,
                 "c": orc.add(field, orc.mul(field, alpha, wu), orc.mul(field, beta, wv))}
    if pos != proof.shape[0]:
        return False, "proof length"
    return True, ""
