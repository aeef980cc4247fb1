"""ctypes binding for oracle/libzk_oracle.so (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Arrays cross as numpy uint64 arrays of shape (n, 4): ark-ff layout (4 LE limbs, Montgomery, R = 2^256).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZK_ORACLE_LIB: another build of the same source (the sanitizer build of `make -C oracle asan`, tools/run_sanitized.sh)
_LIB_PATH = os.environ.get("ZK_ORACLE_LIB") or os.path.join(_HERE, "libzk_oracle.so")
if os.path.realpath(_LIB_PATH) != os.path.realpath(os.path.join(_HERE, "libzk_oracle.so")):
    import sys as _sys

    _root = os.path.realpath(os.path.dirname(_HERE))
    if not os.path.realpath(_LIB_PATH).startswith(_root + os.sep):
        raise ImportError(f"ZK_ORACLE_LIB={_LIB_PATH} is outside {_root}: refused (the override is for builds of this tree only)")
    print(f"oracle: ZK_ORACLE_LIB override active: loading {os.path.realpath(_LIB_PATH)}", file=_sys.stderr, flush=True)

BN254_FR, BLS12_381_FR, BLS12_377_FR = 0, 1, 2

ERRORS = {
    -1: "evaluation vec len should equal 2^n_vars",
    -2: "evaluate must assign to all variables",
    -3: "cannot create product polynomial from empty polynomials",
    -4: "cannot create product polynomial from polynomial that don't share the same number of variables",
    -5: "reference panics: index underflow",
    -6: "values must be a power of 2",
    -7: "reference panics: no root of unity (unwrap on None)",
    -8: "invalid proof: require 1 round poly for each variable in poly",
    -9: "verifier check failed: claimed_sum != p(0) + p(1)",
    -10: "coefficient map represents more than specificed number of variables",
}


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(ERRORS.get(code, f"oracle error {code}"))
        self.code = code


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    return ctypes.CDLL(_LIB_PATH)


_lib = _load()
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_c = ctypes


def _p(a):
    return a.ctypes.data_as(_u64p)


def _arr(x, n=None):
    a = np.ascontiguousarray(x, dtype=np.uint64)
    if n is not None:
        assert a.size == 4 * n, (a.shape, n)
    return a


def _check(rc):
    if rc < 0:
        raise OracleError(rc)
    return rc


for _name, _res in [
    ("orc_insert_bit", _c.c_uint64),
    ("orc_transcript_new", _c.c_void_p),
]:
    getattr(_lib, _name).restype = _res
_lib.orc_insert_bit.argtypes = [_c.c_uint64, _c.c_uint, _c.c_uint64]
_lib.orc_transcript_free.argtypes = [_c.c_void_p]
_lib.orc_transcript_append.argtypes = [_c.c_void_p, _c.c_char_p, _c.c_size_t]
_lib.orc_transcript_sample_challenge.argtypes = [_c.c_void_p, _c.c_char_p]
_lib.orc_transcript_sample_field_element.argtypes = [_c.c_void_p, _c.c_int, _u64p]
_lib.orc_keccak256.argtypes = [_c.c_char_p, _c.c_size_t, _c.c_char_p]
_lib.orc_fill_random.argtypes = [_c.c_int, _c.c_uint64, _c.c_uint64, _c.c_uint64, _u64p]
_lib.orc_from_be_bytes_mod_order.argtypes = [_c.c_int, _c.c_char_p, _c.c_size_t, _u64p]
_lib.orc_pow.argtypes = [_c.c_int, _u64p, _c.c_uint64, _u64p]
_lib.orc_from_u64.argtypes = [_c.c_int, _c.c_uint64, _u64p]
_lib.orc_root_of_unity.argtypes = [_c.c_int, _c.c_uint64, _u64p]


# ---------------- field ----------------
def modulus(field):
    out = np.zeros(4, dtype=np.uint64)
    _check(_lib.orc_field_modulus(field, _p(out)))
    return sum(int(v) << (64 * i) for i, v in enumerate(out))


def two_adicity(field):
    return _check(_lib.orc_field_two_adicity(field))


def field_constants(field):
    """(-p^-1 mod 2^64, R mod p, R^2 mod p, TWO_ADIC_ROOT_OF_UNITY as a canonical integer) as the C oracle derived them"""
    inv = _c.c_uint64(0)
    r1, r2, root = (np.zeros(4, dtype=np.uint64) for _ in range(3))
    _lib.orc_field_constants.argtypes = [_c.c_int, _c.POINTER(_c.c_uint64), _u64p, _u64p, _u64p]
    _check(_lib.orc_field_constants(field, _c.byref(inv), _p(r1), _p(r2), _p(root)))
    as_int = lambda a: sum(int(v) << (64 * i) for i, v in enumerate(a))
    return int(inv.value), as_int(r1), as_int(r2), to_int(field, root)


def _binop(fn):
    def f(field, a, b):
        a, b = _arr(a, 1), _arr(b, 1)
        out = np.zeros(4, dtype=np.uint64)
        fn(field, _p(a), _p(b), _p(out))
        return out

    return f


add = _binop(_lib.orc_add)
sub = _binop(_lib.orc_sub)
mul = _binop(_lib.orc_mul)


def pow_(field, a, e):
    a = _arr(a, 1)
    out = np.zeros(4, dtype=np.uint64)
    _lib.orc_pow(field, _p(a), e, _p(out))
    return out


def inverse(field, a):
    a = _arr(a, 1)
    out = np.zeros(4, dtype=np.uint64)
    if _lib.orc_inverse(field, _p(a), _p(out)):
        return None
    return out


def from_u64(field, v):
    out = np.zeros(4, dtype=np.uint64)
    _lib.orc_from_u64(field, v, _p(out))
    return out


def from_int(field, v):
    """canonical Python int (any sign/size, reduced mod p) -> Montgomery limbs."""
    v %= modulus(field)
    limbs = np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    _lib.orc_from_canonical(field, _p(limbs), _p(out))
    return out


def from_ints(field, vs):
    """canonical Python ints (any sign/size, reduced mod p) -> (n, 4) Montgomery elements, one library call"""
    n = len(vs)
    if n == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    p = modulus(field)
    raw = b"".join((int(v) % p).to_bytes(32, "little") for v in vs)
    limbs = np.frombuffer(raw, dtype="<u8").astype(np.uint64).reshape(n, 4)
    out = np.zeros((n, 4), dtype=np.uint64)
    _lib.orc_from_canonical_n(field, _p(limbs), _c.c_uint64(n), _p(out))
    return out


def to_int(field, a):
    a = _arr(a, 1)
    out = np.zeros(4, dtype=np.uint64)
    _lib.orc_to_canonical(field, _p(a), _p(out))
    return sum(int(v) << (64 * i) for i, v in enumerate(out))


def to_ints(field, arr):
    arr = _arr(arr).reshape(-1, 4)
    n = arr.shape[0]
    if n == 0:
        return []
    limbs = np.zeros((n, 4), dtype=np.uint64)
    _lib.orc_to_canonical_n(field, _p(arr), _c.c_uint64(n), _p(limbs))
    raw = limbs.astype("<u8").tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(n)]


def to_bytes_be(field, a):
    a = _arr(a, 1)
    buf = _c.create_string_buffer(32)
    _lib.orc_to_bytes_be(field, _p(a), _c.cast(buf, _u8p))
    return buf.raw


def from_be_bytes_mod_order(field, b):
    out = np.zeros(4, dtype=np.uint64)
    _lib.orc_from_be_bytes_mod_order(field, bytes(b), len(b), _p(out))
    return out


def root_of_unity(field, n):
    out = np.zeros(4, dtype=np.uint64)
    _check(_lib.orc_root_of_unity(field, n, _p(out)))
    return out


def fill_random(field, seed, count, first_index=0):
    out = np.zeros((count, 4), dtype=np.uint64)
    _lib.orc_fill_random(field, seed, first_index, count, _p(out))
    return out


# ---------------- pairing_index ----------------
def insert_bit(val, index, bit):
    return int(_lib.orc_insert_bit(val, index, bit))


def index_pair(n_vars, index):
    n = 1 << max(n_vars - 1, 0)
    l = np.zeros(n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    _check(_lib.orc_index_pair(n_vars, index, _p(l), _p(r)))
    return list(zip(l.tolist(), r.tolist()))


# ---------------- MLE ----------------
def mle_new_check(n_vars, length):
    _check(_lib.orc_mle_new_check(_c.c_uint64(n_vars), _c.c_uint64(length)))


def mle_partial_evaluate(field, n_vars, evals, initial_var, assignments):
    evals = _arr(evals, 1 << n_vars)
    assignments = _arr(assignments).reshape(-1, 4)
    na = assignments.shape[0]
    out = np.zeros((1 << max(n_vars - na, 0), 4), dtype=np.uint64)
    _check(_lib.orc_mle_partial_evaluate(field, _c.c_uint64(n_vars), _p(evals), _c.c_uint64(initial_var),
                                         _p(assignments), _c.c_uint64(na), _p(out)))
    return out


def mle_evaluate(field, n_vars, evals, point):
    evals = _arr(evals, 1 << n_vars)
    point = _arr(point).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    _check(_lib.orc_mle_evaluate(field, _c.c_uint64(n_vars), _p(evals), _p(point),
                                 _c.c_uint64(point.shape[0]), _p(out)))
    return out


def mle_to_bytes(field, n_vars, evals):
    evals = _arr(evals, 1 << n_vars)
    out = np.zeros(32 << n_vars, dtype=np.uint8)
    _lib.orc_mle_to_bytes(field, _c.c_uint64(n_vars), _p(evals), out.ctypes.data_as(_u8p))
    return out.tobytes()


def fold_msb_parallel(field, n_vars, evals, r, threads=0):
    evals = _arr(evals, 1 << n_vars)
    r = _arr(r, 1)
    out = np.zeros((1 << (n_vars - 1), 4), dtype=np.uint64)
    used = _check(_lib.orc_fold_msb_parallel(field, _c.c_uint64(n_vars), _p(evals), _p(r), _p(out), int(threads)))
    return out, used


def coeff_to_evaluation(field, n_vars, keys, coeffs):
    keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
    coeffs = _arr(coeffs).reshape(-1, 4)
    out = np.zeros(((1 << n_vars) if n_vars else 0, 4), dtype=np.uint64)
    _check(_lib.orc_coeff_to_evaluation(field, _c.c_uint64(n_vars), _p(keys), _p(coeffs), _c.c_uint64(keys.size), _p(out)))
    return out


# ---------------- product ----------------
def _table_ptrs(tables, n_vars):
    tabs = [_arr(t, 1 << n_vars) for t in tables]
    ptrs = (_u64p * len(tabs))(*[_p(t) for t in tabs])
    return tabs, ptrs


def product_new_check(n_vars_each):
    a = np.array(list(n_vars_each), dtype=np.uint64)
    _check(_lib.orc_product_new_check(_c.c_uint64(len(a)), _p(a)))


def prod_reduce(field, n_vars, tables):
    tabs, ptrs = _table_ptrs(tables, n_vars)
    out = np.zeros((1 << n_vars, 4), dtype=np.uint64)
    _lib.orc_prod_reduce(field, _c.c_uint64(len(tabs)), _c.c_uint64(n_vars), ptrs, _p(out))
    return out


def sum_elems(field, elems):
    """iter().sum::<F>() (prover.rs:53-54) of an (n, 4) array"""
    a = _arr(elems)
    out = np.zeros(4, dtype=np.uint64)
    _lib.orc_sum(field, _p(a), _c.c_uint64(a.shape[0]), _p(out))
    return out


def product_evaluate(field, n_vars, tables, point):
    tabs, ptrs = _table_ptrs(tables, n_vars)
    point = _arr(point).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    _check(_lib.orc_product_evaluate(field, _c.c_uint64(len(tabs)), _c.c_uint64(n_vars), ptrs, _p(point),
                                     _c.c_uint64(point.shape[0]), _p(out)))
    return out


# ---------------- keccak / transcript ----------------
def keccak256(data: bytes) -> bytes:
    buf = _c.create_string_buffer(32)
    _lib.orc_keccak256(bytes(data), len(data), buf)
    return buf.raw


class Transcript:
    def __init__(self):
        self._t = _lib.orc_transcript_new()

    def __del__(self):
        if getattr(self, "_t", None):
            _lib.orc_transcript_free(self._t)
            self._t = None

    def append(self, data: bytes):
        _lib.orc_transcript_append(self._t, bytes(data), len(data))

    def sample_challenge(self) -> bytes:
        buf = _c.create_string_buffer(32)
        _lib.orc_transcript_sample_challenge(self._t, buf)
        return buf.raw

    def sample_field_element(self, field):
        out = np.zeros(4, dtype=np.uint64)
        _lib.orc_transcript_sample_field_element(self._t, field, _p(out))
        return out


# ---------------- sumcheck ----------------
def sumcheck_prove(field, n_vars, tables, D, claimed_sum, absorb_table):
    tabs, ptrs = _table_ptrs(tables, n_vars)
    s = _arr(claimed_sum, 1)
    rp = np.zeros((n_vars, D + 1, 4), dtype=np.uint64)
    ch = np.zeros((n_vars, 4), dtype=np.uint64)
    _check(_lib.orc_sumcheck_prove(field, _c.c_uint64(len(tabs)), _c.c_uint64(n_vars), ptrs, _c.c_uint(D),
                                   _p(s), int(bool(absorb_table)), _p(rp), _p(ch)))
    return rp, ch


def sumcheck_prove_fused_parallel(field, n_vars, tables, D, claimed_sum, threads=0):
    """bench 'optimised CPU' row: prove_partial outputs via fused rounds + OpenMP -> (round_polys, challenges, threads)"""
    tabs, ptrs = _table_ptrs(tables, n_vars)
    s = _arr(claimed_sum, 1)
    rp = np.zeros((n_vars, D + 1, 4), dtype=np.uint64)
    ch = np.zeros((n_vars, 4), dtype=np.uint64)
    used = _check(_lib.orc_sumcheck_prove_fused_parallel(field, _c.c_uint64(len(tabs)), _c.c_uint64(n_vars), ptrs, _c.c_uint(D),
                                                         _p(s), _p(rp), _p(ch), int(threads)))
    return rp, ch, used


def sumcheck_verify_partial(field, D, claimed_sum, round_polys, table_bytes=None):
    rp = _arr(round_polys).reshape(-1, D + 1, 4)
    n = rp.shape[0]
    s = _arr(claimed_sum, 1)
    sub_ = np.zeros(4, dtype=np.uint64)
    ch = np.zeros((max(n, 1), 4), dtype=np.uint64)
    tb = bytes(table_bytes) if table_bytes is not None else None
    _lib.orc_sumcheck_verify_partial.argtypes = [_c.c_int, _c.c_uint64, _c.c_uint, _u64p, _u64p, _c.c_char_p,
                                                 _c.c_size_t, _u64p, _u64p]
    _check(_lib.orc_sumcheck_verify_partial(field, n, D, _p(s), _p(rp), tb, len(tb) if tb else 0, _p(sub_), _p(ch)))
    return sub_, ch[:n]


def sumcheck_verify(field, n_vars, tables, D, claimed_sum, round_polys):
    tabs, ptrs = _table_ptrs(tables, n_vars)
    rp = _arr(round_polys).reshape(-1, D + 1, 4)
    s = _arr(claimed_sum, 1)
    rc = _lib.orc_sumcheck_verify(field, _c.c_uint64(len(tabs)), _c.c_uint64(n_vars), ptrs,
                                  _c.c_uint64(rp.shape[0]), _c.c_uint(D), _p(s), _p(rp))
    return bool(_check(rc))


def _ragged(round_polys):
    """list of per-round (len_r, 4) arrays -> (lens uint32, concatenated (sum len, 4) array)"""
    rounds = [_arr(r).reshape(-1, 4) for r in round_polys]
    lens = np.array([r.shape[0] for r in rounds], dtype=np.uint32)
    flat = np.concatenate(rounds, axis=0) if rounds and int(lens.sum()) else np.zeros((1, 4), dtype=np.uint64)
    return lens, np.ascontiguousarray(flat)


def sumcheck_verify_partial_lengths(field, claimed_sum, round_polys, table_bytes=None):
    """verify_partial on a proof whose rounds carry different numbers of evaluations (verifier.rs:55-58)"""
    lens, flat = _ragged(round_polys)
    n = len(lens)
    s = _arr(claimed_sum, 1)
    sub_ = np.zeros(4, dtype=np.uint64)
    ch = np.zeros((max(n, 1), 4), dtype=np.uint64)
    tb = bytes(table_bytes) if table_bytes is not None else None
    fn = _lib.orc_sumcheck_verify_partial_lengths
    fn.argtypes = [_c.c_int, _c.c_uint64, _c.c_void_p, _u64p, _u64p, _c.c_char_p, _c.c_size_t, _u64p, _u64p]
    lens_buf = np.ascontiguousarray(np.append(lens, np.uint32(0)))
    _check(fn(field, n, lens_buf.ctypes.data, _p(s), _p(flat), tb, len(tb) if tb else 0, _p(sub_), _p(ch)))
    return sub_, ch[:n]


def sumcheck_verify_lengths(field, n_vars, tables, claimed_sum, round_polys):
    tabs, ptrs = _table_ptrs(tables, n_vars)
    lens, flat = _ragged(round_polys)
    s = _arr(claimed_sum, 1)
    fn = _lib.orc_sumcheck_verify_lengths
    fn.argtypes = [_c.c_int, _c.c_uint64, _c.c_uint64, type(ptrs), _c.c_uint64, _c.c_void_p, _u64p, _u64p]
    lens_buf = np.ascontiguousarray(np.append(lens, np.uint32(0)))
    return bool(_check(fn(field, len(tabs), n_vars, ptrs, len(lens), lens_buf.ctypes.data, _p(s), _p(flat))))


# ---------------- fft ----------------
def _fft_call(fn, field, values, *extra):
    v = _arr(values).reshape(-1, 4)
    n = v.shape[0]
    out = np.zeros((n, 4), dtype=np.uint64)
    _check(fn(field, _p(v), _c.c_uint64(n), *extra, _p(out)))
    return out


def fft(field, values):
    return _fft_call(_lib.orc_fft, field, values)


def ifft(field, values):
    return _fft_call(_lib.orc_ifft, field, values)


def ntt_fast(field, values, inverse=False):
    return _fft_call(_lib.orc_ntt_fast, field, values, _c.c_int(int(inverse)))


def dft_point(field, values, k, inverse=False):
    """one output of fft / ifft from the definition sum_j values[j] * omega^(j*k) (fft/src/lib.rs:39-45)"""
    v = _arr(values).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    fn = _lib.orc_dft_point
    fn.argtypes = [_c.c_int, _u64p, _c.c_uint64, _c.c_uint64, _c.c_int, _u64p]
    _check(fn(field, _p(v), v.shape[0], int(k), int(bool(inverse)), _p(out)))
    return out


# ---------------- checker pieces for the GKR-shaped driver (no reference crate: DESIGN.md section 10, oracle/gkr_ref.py) ----------------
def sumcheck_verify_partial_lengths_on(tr, field, claimed_sum, round_polys):
    """verify_internal's own signature (verifier.rs:44-48): the rounds on a Transcript the caller holds and keeps"""
    lens, flat = _ragged(round_polys)
    n = len(lens)
    s = _arr(claimed_sum, 1)
    sub_ = np.zeros(4, dtype=np.uint64)
    ch = np.zeros((max(n, 1), 4), dtype=np.uint64)
    fn = _lib.orc_sumcheck_verify_partial_lengths_on
    fn.argtypes = [_c.c_void_p, _c.c_int, _c.c_uint64, _c.c_void_p, _u64p, _u64p, _u64p, _u64p]
    lens_buf = np.ascontiguousarray(np.append(lens, np.uint32(0)))
    _check(fn(tr._t, field, n, lens_buf.ctypes.data, _p(s), _p(flat), _p(sub_), _p(ch)))
    return sub_, ch[:n]


def circuit_layer(field, op, left, right, w):
    """one layer of a fan-in-2 circuit over the values w: add (op 0) / mul (op 1) gates"""
    op = np.ascontiguousarray(op, dtype=np.uint8)
    left = np.ascontiguousarray(left, dtype=np.uint32)
    right = np.ascontiguousarray(right, dtype=np.uint32)
    w = _arr(w).reshape(-1, 4)
    assert left.max(initial=0) < w.shape[0] and right.max(initial=0) < w.shape[0]
    out = np.zeros((op.shape[0], 4), dtype=np.uint64)
    fn = _lib.orc_circuit_layer
    fn.argtypes = [_c.c_int, _c.c_uint64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _u64p, _u64p]
    _check(fn(field, op.shape[0], op.ctypes.data, left.ctypes.data, right.ctypes.data, _p(w), _p(out)))
    return out


def tree_digest(data) -> bytes:
    """the driver's statement digest: Keccak-256 over 128-byte leaves, then 4-ary nodes (gkr_ref.tree_digest)"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    out = _c.create_string_buffer(32)
    fn = _lib.orc_tree_digest
    fn.argtypes = [_c.c_void_p, _c.c_size_t, _c.c_char_p]
    keep = np.ascontiguousarray(buf) if buf.size else np.zeros(1, dtype=np.uint8)
    _check(fn(keep.ctypes.data, buf.size, out))
    return out.raw


def eq_table(field, point):
    pt = _arr(point).reshape(-1, 4)
    out = np.zeros((1 << pt.shape[0], 4), dtype=np.uint64)
    fn = _lib.orc_eq_table
    fn.argtypes = [_c.c_int, _u64p, _c.c_uint64, _u64p]
    _check(fn(field, _p(pt if pt.size else np.zeros((1, 4), dtype=np.uint64)), pt.shape[0], _p(out)))
    return out


def gkr_wiring_sums(field, op, left, right, e1, e2, alpha, beta, eq_u, eq_v):
    """(add_e, mul_e) = sums over the add / mul gates of (alpha e1 + beta e2)[z] eq_u[left[z]] eq_v[right[z]]; e2 may be None"""
    op = np.ascontiguousarray(op, dtype=np.uint8)
    left = np.ascontiguousarray(left, dtype=np.uint32)
    right = np.ascontiguousarray(right, dtype=np.uint32)
    e1, eq_u, eq_v = _arr(e1), _arr(eq_u), _arr(eq_v)
    e2 = _arr(e2) if e2 is not None else None
    a, m = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    fn = _lib.orc_gkr_wiring_sums
    fn.argtypes = [_c.c_int, _c.c_uint64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _u64p, _c.c_void_p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p]
    _check(fn(field, op.shape[0], op.ctypes.data, left.ctypes.data, right.ctypes.data, _p(e1), e2.ctypes.data if e2 is not None else None,
              _p(_arr(alpha, 1)), _p(_arr(beta, 1)), _p(eq_u), _p(eq_v), _p(a), _p(m)))
    return a, m
