"""GKR-shaped driver over the C ABI (include/zk_amd.h, "sum of products + GKR-shaped driver").

The reference has NO gkr crate; what it offers GKR is `prove_partial` / `verify_partial`
(sumcheck/src/prover.rs:24-30, sumcheck/src/verifier.rs:38-41; the intent is noted at
polynomial/src/multilinear/evaluation_form.rs:45-48).  This module is the layered loop on top of those calls: the
circuit format, transcript schedule and proof layout are this library's own.
"""
import ctypes as c

import numpy as np

from ._lib import check, lib, u8p
from .api import MultiLinearPolynomial, _elems, _handles, _p

u32p = c.POINTER(c.c_uint32)
ADD, MUL = 0, 1


class SumOfProductsPoly:
    """sum_i prod_{f in term i} MLE_f: `terms` is a list of lists of MultiLinearPolynomial (one inner list == the
    reference's ProductPoly, product_poly.rs:7-10).  A table may appear once only."""

    def __init__(self, terms):
        if not terms or any(not t for t in terms):
            raise ValueError("cannot create product polynomial from empty polynomials")
        self.terms = [list(t) for t in terms]
        self.ctx = self.terms[0][0].ctx

    def n_vars(self):
        return self.terms[0][0].n_vars()

    def flat(self):
        return [f for t in self.terms for f in t]


def prove_partial_terms(poly, max_var_degree, sum_, consume=False):
    """SumcheckProver::<D,F>::prove_partial (prover.rs:24-30) on a sum of products.  Returns
    (round_polys [n][D+1][4], challenges [n][4], factor evaluations at the challenge point [k][4])."""
    D, n = max_var_degree, poly.n_vars()
    flat = poly.flat()
    tk = np.array([len(t) for t in poly.terms], dtype=np.uint64)
    s = _elems(sum_, 1)
    rp = np.zeros((n, D + 1, 4), dtype=np.uint64)
    ch = np.zeros((n, 4), dtype=np.uint64)
    fin = np.zeros((len(flat), 4), dtype=np.uint64)
    hp, keep = _handles(flat)
    check(lib.zk_sumcheck_prove_terms(poly.ctx._h, hp, _p(tk), len(tk), D, _p(s), int(consume), _p(rp), _p(ch), _p(fin)))
    return rp, ch, fin


def eq_table(ctx, point):
    """eq(point, .) as a device table (variable 0 = index MSB)."""
    pt = _elems(point) if len(point) else np.zeros((0, 4), dtype=np.uint64)
    h = c.c_void_p()
    check(lib.zk_eq_table(ctx._h, _p(pt), pt.shape[0], c.byref(h)))
    return MultiLinearPolynomial(ctx, h)


class Circuit:
    """Layered fan-in-2 arithmetic circuit; layers are appended from the output layer towards the inputs."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._h = c.c_void_p()
        check(lib.zk_circuit_create(ctx._h, c.byref(self._h)))

    def add_layer(self, log_out, log_in, op, left, right):
        op = np.ascontiguousarray(op, dtype=np.uint8)
        left = np.ascontiguousarray(left, dtype=np.uint32)
        right = np.ascontiguousarray(right, dtype=np.uint32)
        if not (op.shape == left.shape == right.shape == (1 << log_out,)):
            raise ValueError("a layer needs 2^log_out gates")
        check(lib.zk_circuit_add_layer(self._h, log_out, log_in, op.ctypes.data_as(u8p), left.ctypes.data_as(u32p),
                                       right.ctypes.data_as(u32p)))
        return self

    def depth(self):
        n = c.c_uint64()
        check(lib.zk_circuit_depth(self._h, c.byref(n)))
        return n.value

    def layer_dims(self, i):
        a, b = c.c_uint64(), c.c_uint64()
        check(lib.zk_circuit_layer_dims(self._h, i, c.byref(a), c.byref(b)))
        return a.value, b.value

    def proof_elems(self):
        n = c.c_uint64()
        check(lib.zk_circuit_proof_elems(self._h, c.byref(n)))
        return n.value

    def evaluate(self, inputs):
        h = c.c_void_p()
        check(lib.zk_gkr_evaluate(self._h, inputs._h, c.byref(h)))
        return MultiLinearPolynomial(self.ctx, h)

    def free(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None) and lib is not None:   # (lib is None at interpreter exit)
            lib.zk_circuit_free(self._h)
        self._h = None

    def __del__(self):
        self.free()


def _seed(seed):
    b = bytes(seed)
    if len(b) != 32:
        raise ValueError("seed must be 32 bytes")
    return (c.c_uint8 * 32).from_buffer_copy(b)


def gkr_prove(circuit, inputs, seed):
    """-> (outputs table, proof [proof_elems][4]).  seed: 32 bytes binding the statement."""
    proof = np.zeros((circuit.proof_elems(), 4), dtype=np.uint64)
    h = c.c_void_p()
    check(lib.zk_gkr_prove(circuit._h, inputs._h, _seed(seed), c.byref(h), _p(proof)))
    return MultiLinearPolynomial(circuit.ctx, h), proof


def gkr_verify(circuit, inputs, outputs, seed, proof):
    """True = accept; False = a sumcheck or GKR check failed (status -9 / -27); other errors raise."""
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    if proof.shape != (circuit.proof_elems(), 4):
        raise ValueError("proof has the wrong length for this circuit")
    rc = lib.zk_gkr_verify(circuit._h, inputs._h, outputs._h, _seed(seed), _p(proof))
    if rc in (-9, -27):
        return False
    check(rc)
    return True
