"""Python mirror of the reference's public API for the hot path, over the C ABI (include/zk_amd.h).

Same names, argument meaning and error behaviour as the reference crates (paths relative to the reference):
  polynomial::multilinear::evaluation_form::MultiLinearPolynomial   evaluation_form.rs:7-103
  polynomial::product_poly::ProductPoly                             product_poly.rs:7-88
  sumcheck::prover::SumcheckProver<MAX_VAR_DEGREE, F>               prover.rs:9-73
  sumcheck::verifier::SumcheckVerifier<F>                           verifier.rs:9-78
  transcript::Transcript                                            transcript/src/lib.rs:5-35
  fft::{fft, ifft, fft_internal}                                    fft/src/lib.rs:4-46
`Err(&'static str)` becomes ZkError with the same text; misuse the reference panics on becomes ZkError too.
Field elements are numpy uint64 arrays of shape (..., 4): ark-ff's in-memory layout (Montgomery, 4 LE limbs).
Everything computes on the GPU; this module is the test/bench harness and the host side of the sharded prover.
"""
import numpy as np

from ._lib import ZkError, c, check, lib, u8p, u64p

BN254_FR, BLS12_381_FR, BLS12_377_FR = 0, 1, 2
FIELD_NAMES = {BN254_FR: "bn254_fr", BLS12_381_FR: "bls12_381_fr", BLS12_377_FR: "bls12_377_fr"}


def _p(a):
    return a.ctypes.data_as(u64p)


def _elems(x, n=None):
    a = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    if n is not None and a.shape[0] != n:
        raise ValueError(f"expected {n} field elements, got {a.shape[0]}")
    return a


# ---- host-side element helpers (no GPU needed) ---------------------------------------------------------------
def modulus(field):
    out = np.zeros(4, dtype=np.uint64)
    check(lib.zk_field_modulus(field, _p(out)))
    return sum(int(v) << (64 * i) for i, v in enumerate(out))


def two_adicity(field):
    s = c.c_int32()
    check(lib.zk_field_two_adicity(field, c.byref(s)))
    return s.value


def root_of_unity(field, log_n):
    """F::get_root_of_unity(2^log_n) (fft/src/lib.rs:6) as a Montgomery element"""
    out = np.zeros(4, dtype=np.uint64)
    check(lib.zk_field_root_of_unity(field, log_n, _p(out)))
    return out


def fe_from_int(field, v):
    """F::from(v) for any Python int (reduced mod p first)."""
    v %= modulus(field)
    limbs = np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    check(lib.zk_fe_from_canonical(field, _p(limbs), _p(out)))
    return out


def fe_from_ints(field, vs):
    return np.stack([fe_from_int(field, v) for v in vs]) if len(vs) else np.zeros((0, 4), dtype=np.uint64)


def fe_to_int(field, a):
    a = _elems(a, 1)
    out = np.zeros(4, dtype=np.uint64)
    check(lib.zk_fe_to_canonical(field, _p(a), _p(out)))
    return sum(int(v) << (64 * i) for i, v in enumerate(out))


def fe_to_ints(field, arr):
    arr = _elems(arr)
    return [fe_to_int(field, arr[i]) for i in range(arr.shape[0])]


def mask(n):  # polynomial/src/multilinear/pairing_index.rs:24-26
    """a bit sequence of n ones: mask(1) -> 1, mask(3) -> 0b111"""
    if not 0 <= n < 64:
        raise ZkError(-5, lib.zk_strerror(-5).decode())   # the reference's usize shift overflows here
    return (1 << n) - 1


def index_pair(n_vars, index):  # polynomial/src/multilinear/pairing_index.rs:2-9
    """the (left, right) table indices that partial_evaluate pairs up when it assigns variable `index` (variable 0 = index MSB)"""
    if n_vars < 1 or not 0 <= index < n_vars:
        raise ZkError(-5, lib.zk_strerror(-5).decode())   # u8 underflow in the reference
    pos = n_vars - 1 - index
    out = []
    for j in range(1 << (n_vars - 1)):
        left = ((j >> pos) << (pos + 1)) | (j & mask(pos))
        out.append((left, left | (1 << pos)))
    return out


def keccak256(data: bytes) -> bytes:
    buf = c.create_string_buffer(32)
    check(lib.zk_keccak256(bytes(data), len(data), buf))
    return buf.raw


class Transcript:
    """transcript::Transcript (transcript/src/lib.rs:5-35)."""

    def __init__(self):
        h = c.c_void_p()
        check(lib.zk_transcript_new(c.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.zk_transcript_free(self._h)
            self._h = None

    def append(self, new_data: bytes):
        check(lib.zk_transcript_append(self._h, bytes(new_data), len(new_data)))

    def sample_challenge(self) -> bytes:
        buf = c.create_string_buffer(32)
        check(lib.zk_transcript_sample_challenge(self._h, buf))
        return buf.raw

    def sample_field_element(self, field):
        out = np.zeros(4, dtype=np.uint64)
        check(lib.zk_transcript_sample_field_element(self._h, field, _p(out)))
        return out

    def sample_n_field_elements(self, field, n):   # transcript/src/lib.rs:32-34
        out = np.zeros((n, 4), dtype=np.uint64)
        check(lib.zk_transcript_sample_n_field_elements(self._h, field, n, _p(out)))
        return out


# ---- device context ------------------------------------------------------------------------------------------
class Context:
    """One device + stream (zk_ctx).  Fails loudly when there is no gfx950 device."""

    def __init__(self, field=BN254_FR, device=0):
        h = c.c_void_p()
        check(lib.zk_ctx_create(field, device, c.byref(h)))
        self._h, self.field, self.device = h, field, device

    def close(self):
        if getattr(self, "_h", None):
            lib.zk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def synchronize(self):
        check(lib.zk_ctx_synchronize(self._h))

    def set_stream(self, stream_ptr):
        """run on a caller-owned hipStream_t (0 / None = the legacy default stream)"""
        check(lib.zk_ctx_set_stream(self._h, c.c_void_p(stream_ptr or 0)))

    def use_own_stream(self):
        check(lib.zk_ctx_use_own_stream(self._h))

    def trim(self):
        """drop the context's cache of freed device blocks (zk_ctx_trim)"""
        check(lib.zk_ctx_trim(self._h))

    def use_torch_stream(self):
        """share torch's current stream on this device, so kernels and torch.distributed collectives are ordered"""
        import torch

        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    # measurement hooks
    def bench_copy(self, nbytes, reps=10):
        out = c.c_double()
        check(lib.zk_bench_copy(self._h, nbytes, reps, c.byref(out)))
        return out.value

    def bench_modmul(self, iters=2000, variant=0):
        out = c.c_double()
        check(lib.zk_bench_modmul(self._h, variant, iters, c.byref(out)))
        return out.value


def _handles(polys):
    arr = (c.c_void_p * max(len(polys), 1))(*[q._h for q in polys])
    return c.cast(arr, c.POINTER(c.c_void_p)), arr


class MultiLinearPolynomial:
    """Dense MLE table resident in HBM (evaluation_form.rs:7-10)."""

    def __init__(self, ctx, handle):
        self.ctx, self._h = ctx, handle

    # MultiLinearPolynomial::new (evaluation_form.rs:15-27)
    @classmethod
    def new(cls, ctx, n_vars, evaluations):
        ev = _elems(evaluations)
        h = c.c_void_p()
        check(lib.zk_mle_upload(ctx._h, n_vars, _p(ev), ev.shape[0], c.byref(h)))
        return cls(ctx, h)

    @classmethod
    def alloc(cls, ctx, n_vars):
        h = c.c_void_p()
        check(lib.zk_mle_alloc(ctx._h, n_vars, c.byref(h)))
        return cls(ctx, h)

    @classmethod
    def random(cls, ctx, n_vars, seed, first_index=0):
        t = cls.alloc(ctx, n_vars)
        check(lib.zk_mle_fill_random(ctx._h, t._h, seed, first_index))
        return t

    def free(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib.zk_mle_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        self.free()

    def clone(self):
        h = c.c_void_p()
        check(lib.zk_mle_clone(self.ctx._h, self._h, c.byref(h)))
        return MultiLinearPolynomial(self.ctx, h)

    def n_vars(self):
        n = c.c_uint64()
        check(lib.zk_mle_n_vars(self._h, c.byref(n)))
        return n.value

    def device_ptr(self):
        p = c.c_void_p()
        check(lib.zk_mle_device_ptr(self._h, c.byref(p)))
        return p.value

    # partial_evaluate (evaluation_form.rs:40-80)
    def partial_evaluate(self, initial_var, assignments):
        a = _elems(assignments)
        h = c.c_void_p()
        check(lib.zk_mle_partial_evaluate(self.ctx._h, self._h, initial_var, _p(a), a.shape[0], c.byref(h)))
        return MultiLinearPolynomial(self.ctx, h)

    def fold_into(self, r, out):
        r = _elems(r, 1)
        check(lib.zk_mle_fold_into(self.ctx._h, self._h, _p(r), out._h))
        return out

    def bench_fold(self, r, out, reps):
        r = _elems(r, 1)
        ms = c.c_double()
        check(lib.zk_bench_fold(self.ctx._h, self._h, _p(r), out._h, reps, c.byref(ms)))
        return ms.value

    def bench_fold_samples(self, r, out, reps, group=1):
        """launch durations (ms) of `reps` folds: one HIP event every `group` launches, each sample the average inside its group"""
        r = _elems(r, 1)
        ms = np.zeros((reps + group - 1) // group, dtype=np.float64)
        check(lib.zk_bench_fold_samples(self.ctx._h, self._h, _p(r), out._h, reps, group, ms.ctypes.data_as(c.POINTER(c.c_double))))
        return ms

    # evaluate (evaluation_form.rs:83-89)
    def evaluate(self, assignments):
        a = _elems(assignments)
        out = np.zeros(4, dtype=np.uint64)
        check(lib.zk_mle_evaluate(self.ctx._h, self._h, _p(a), a.shape[0], _p(out)))
        return out

    # evaluation_slice (evaluation_form.rs:92-94)
    def evaluation_slice(self):
        out = np.zeros((1 << self.n_vars(), 4), dtype=np.uint64)
        check(lib.zk_mle_download(self.ctx._h, self._h, _p(out)))
        return out

    # to_bytes (evaluation_form.rs:97-103)
    def to_bytes_array(self, out=None):
        """to_bytes (evaluation_form.rs:97-103) into a numpy uint8 array (a fresh one, or `out`): no second host copy"""
        if out is None:
            out = np.empty(32 << self.n_vars(), dtype=np.uint8)
        want = 32 << self.n_vars()   # zk_mle_to_bytes writes exactly this many bytes, on several host threads
        if not isinstance(out, np.ndarray) or out.dtype != np.uint8 or out.size != want:
            raise ValueError(f"to_bytes_array: out must be a numpy uint8 array of {want} bytes")
        if not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise ValueError("to_bytes_array: out must be C-contiguous and writeable")
        check(lib.zk_mle_to_bytes(self.ctx._h, self._h, out.ctypes.data_as(u8p)))
        return out

    def to_bytes(self):
        return self.to_bytes_array().tobytes()

    def __eq__(self, other):  # #[derive(PartialEq)] (evaluation_form.rs:4)
        if not isinstance(other, MultiLinearPolynomial):
            return NotImplemented
        if other.ctx is not self.ctx:   # different contexts: compare on the host
            return self.n_vars() == other.n_vars() and np.array_equal(self.evaluation_slice(), other.evaluation_slice())
        eq = c.c_int32()
        check(lib.zk_mle_equal(self.ctx._h, self._h, other._h, c.byref(eq)))
        return bool(eq.value)

    __hash__ = None


class CoeffMultilinearPolynomial:
    """polynomial::multilinear::coefficient_form::CoeffMultilinearPolynomial (coefficient_form.rs:27-30), only as the
    producer of evaluation tables: ::new (:158-176), ::new_with_coefficient (:178-193), ::to_evaluation_form (:340-347,
    computed on the GPU and left resident as a MultiLinearPolynomial)."""

    def __init__(self, field, n_vars, coefficients):
        # {key: element}, key bit v <-> variable v.  The term map is IMMUTABLE (the reference's BTreeMap is a private field,
        # coefficient_form.rs:27-30): a private copy, exposed read-only, flattened once into the two arrays the library takes
        import types

        self.field, self._n_vars = field, n_vars
        self._terms = {int(k): np.array(v, dtype=np.uint64).reshape(4) for k, v in dict(coefficients).items()}
        self.coefficients = types.MappingProxyType(self._terms)
        keys = np.array(sorted(self._terms), dtype=np.uint64)
        coeffs = np.stack([self._terms[int(k)] for k in keys]) if len(keys) else np.zeros((0, 4), dtype=np.uint64)
        for v in self._terms.values():
            v.setflags(write=False)
        self._flat = (keys, np.ascontiguousarray(coeffs, dtype=np.uint64))

    @classmethod
    def new(cls, field, number_of_variables, terms):
        """terms: [(coefficient element, [bool] * n_vars)]; duplicate selectors are summed (:164-171)."""
        p = modulus(field)
        acc = {}
        for coeff, selector in terms:
            if len(selector) != number_of_variables:
                raise ZkError(-20, "the selector array len should be the same as the number of variables")
            key = sum(1 << v for v, present in enumerate(selector) if present)          # selector_to_index :418-430
            acc[key] = (acc.get(key, 0) + fe_to_int(field, coeff)) % p
        return cls(field, number_of_variables, {k: fe_from_int(field, v) for k, v in acc.items()})

    @classmethod
    def new_with_coefficient(cls, field, number_of_variables, coefficients):
        if coefficients and max(coefficients) >= (1 << number_of_variables):
            raise ZkError(-10, "coefficient map represents more than specificed number of variables")
        return cls(field, number_of_variables, dict(coefficients))

    def n_vars(self):
        return self._n_vars

    def to_evaluation_form(self, ctx):
        keys, coeffs = self._flat
        h = c.c_void_p()
        check(lib.zk_coeff_to_evaluation(ctx._h, self._n_vars, _p(keys), _p(coeffs), len(keys), c.byref(h)))
        return MultiLinearPolynomial(ctx, h)


class ProductPoly:
    """P(x) = A(x).B(x).C(x) (product_poly.rs:7-10)."""

    def __init__(self, polynomials):
        self.polynomials = list(polynomials)
        self.ctx = self.polynomials[0].ctx

    # ProductPoly::new (product_poly.rs:14-32)
    @classmethod
    def new(cls, polynomials):
        polynomials = list(polynomials)
        hp, keep = _handles(polynomials)
        check(lib.zk_product_check(hp, len(polynomials)))
        return cls(polynomials)

    def n_vars(self):  # product_poly.rs:86
        return self.polynomials[0].n_vars()

    def evaluate(self, assignments):  # product_poly.rs:36-44
        a = _elems(assignments)
        hp, keep = _handles(self.polynomials)
        out = np.zeros(4, dtype=np.uint64)
        check(lib.zk_product_evaluate(self.ctx._h, hp, len(self.polynomials), _p(a), a.shape[0], _p(out)))
        return out

    def partial_evaluate(self, initial_var, assignments):  # product_poly.rs:48-63
        return ProductPoly([q.partial_evaluate(initial_var, assignments) for q in self.polynomials])

    def prod_reduce_device(self):
        """prod_reduce with the result left resident: a new MultiLinearPolynomial (the reference returns a Vec<F>)"""
        hp, keep = _handles(self.polynomials)
        h = c.c_void_p()
        check(lib.zk_prod_reduce(self.ctx._h, hp, len(self.polynomials), c.byref(h)))
        return MultiLinearPolynomial(self.ctx, h)

    def prod_reduce(self):  # product_poly.rs:66-74
        return self.prod_reduce_device().evaluation_slice()

    def round_sums(self, max_var_degree):  # prover.rs:49-56 for one round
        hp, keep = _handles(self.polynomials)
        out = np.zeros((max_var_degree + 1, 4), dtype=np.uint64)
        check(lib.zk_round_sums(self.ctx._h, hp, len(self.polynomials), max_var_degree, _p(out)))
        return out

    def to_bytes(self):  # product_poly.rs:77-83
        return b"".join(q.to_bytes() for q in self.polynomials)

    def clone(self):
        return ProductPoly([q.clone() for q in self.polynomials])


class SumcheckProof:
    """sumcheck::SumcheckProof (sumcheck/src/lib.rs:8-11)."""

    def __init__(self, sum_, round_polys):
        self.sum, self.round_polys = sum_, round_polys


class SubClaim:
    """sumcheck::SubClaim (sumcheck/src/lib.rs:17-20)."""

    def __init__(self, sum_, challenges):
        self.sum, self.challenges = sum_, challenges


class SumcheckProver:
    """SumcheckProver::<MAX_VAR_DEGREE, F> (prover.rs:9-73); MAX_VAR_DEGREE is the constructor argument."""

    def __init__(self, max_var_degree):
        self.max_var_degree = max_var_degree

    def _run(self, poly, sum_, absorb, consume):
        D, n, k = self.max_var_degree, poly.n_vars(), len(poly.polynomials)
        s = _elems(sum_, 1)
        rp = np.zeros((n, D + 1, 4), dtype=np.uint64)
        ch = np.zeros((n, 4), dtype=np.uint64)
        hp, keep = _handles(poly.polynomials)
        check(lib.zk_sumcheck_prove(poly.ctx._h, hp, k, D, _p(s), int(absorb), int(consume), _p(rp), _p(ch)))
        return SumcheckProof(s.reshape(4).copy(), rp), ch

    def prove_partial_batch(self, polys, sums, consume=False):
        """len(polys) independent prove_partial calls (prover.rs:24-30), one ProductPoly each, same number of factors and variables,
        proved side by side (zk_sumcheck_prove_batch: one launch per round for all proofs).  Returns [(SumcheckProof, challenges)], each
        pair equal to what prove_partial(poly, sum) returns."""
        polys = list(polys)
        if not polys:
            return []
        D, n, k = self.max_var_degree, polys[0].n_vars(), len(polys[0].polynomials)
        if any(len(q.polynomials) != k for q in polys):
            raise ValueError("prove_partial_batch: every ProductPoly must have the same number of factors")
        B = len(polys)
        s = _elems(sums, B)
        rp = np.zeros((B, n, D + 1, 4), dtype=np.uint64)
        ch = np.zeros((B, n, 4), dtype=np.uint64)
        hp, keep = _handles([f for q in polys for f in q.polynomials])
        check(lib.zk_sumcheck_prove_batch(polys[0].ctx._h, B, hp, k, D, _p(s), int(consume), _p(rp), _p(ch)))
        return [(SumcheckProof(s[b].copy(), rp[b]), ch[b]) for b in range(B)]

    def prove(self, poly, sum_, consume=False):  # prover.rs:15-20
        return self._run(poly, sum_, True, consume)[0]

    def prove_partial(self, poly, sum_, consume=False):  # prover.rs:24-30
        return self._run(poly, sum_, False, consume)


def _ragged_rounds(round_polys):
    """SumcheckProof.round_polys is a Vec<Vec<F>> (sumcheck/src/lib.rs:8-11): a (rounds, D+1, 4) array, or a list of
    per-round (len_r, 4) arrays of different lengths -> (lens uint32, the evaluations back to back)."""
    if isinstance(round_polys, np.ndarray) and round_polys.ndim == 3:
        rp = np.ascontiguousarray(round_polys, dtype=np.uint64)
        lens = np.full(rp.shape[0] + 1, rp.shape[1], dtype=np.uint32)
        return rp.shape[0], lens, rp.reshape(-1, 4) if rp.size else np.zeros((1, 4), dtype=np.uint64)
    rounds = [np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4) for r in round_polys]
    lens = np.array([r.shape[0] for r in rounds] + [0], dtype=np.uint32)
    flat = np.concatenate(rounds + [np.zeros((1, 4), dtype=np.uint64)], axis=0)
    return len(rounds), lens, np.ascontiguousarray(flat)


class SumcheckVerifier:
    """SumcheckVerifier::<F> (verifier.rs:9-78).  Each round polynomial is interpolated at its own length, as
    verifier.rs:55-58 does (round_polys may be a list of arrays of different lengths)."""

    @staticmethod
    def verify(poly, proof):  # verifier.rs:15-33
        n_rp, lens, rp = _ragged_rounds(proof.round_polys)
        hp, keep = _handles(poly.polynomials)
        ok = c.c_int32()
        s = _elems(proof.sum, 1)
        check(lib.zk_sumcheck_verify_lengths(poly.ctx._h, hp, len(poly.polynomials), n_rp,
                                             lens.ctypes.data_as(c.POINTER(c.c_uint32)), _p(s), _p(rp), c.byref(ok)))
        return bool(ok.value)

    @staticmethod
    def verify_partial(field, proof):  # verifier.rs:38-41
        n_rp, lens, rp = _ragged_rounds(proof.round_polys)
        s = _elems(proof.sum, 1)
        sub = np.zeros(4, dtype=np.uint64)
        ch = np.zeros((max(n_rp, 1), 4), dtype=np.uint64)
        check(lib.zk_sumcheck_verify_partial_lengths(field, n_rp, lens.ctypes.data_as(c.POINTER(c.c_uint32)), _p(s), _p(rp),
                                                     _p(sub), _p(ch)))
        return SubClaim(sub, ch[:n_rp])


# ---- fft crate (fft/src/lib.rs) ---------------------------------------------------------------------------------
def _fft_call(fn, ctx, values, *extra):
    v = _elems(values)
    out = np.zeros_like(v)
    check(fn(ctx._h, _p(v), v.shape[0], *extra, _p(out)))
    return out


def fft(ctx, coefficients):  # fft/src/lib.rs:4-8
    return _fft_call(lib.zk_fft_host, ctx, coefficients)


def ifft(ctx, evaluations):  # fft/src/lib.rs:11-19
    return _fft_call(lib.zk_ifft_host, ctx, evaluations)


def fft_internal(ctx, values, omega):  # fft/src/lib.rs:21-46
    w = _elems(omega, 1)
    return _fft_call(lib.zk_fft_internal_host, ctx, values, _p(w))


def ntt(ctx, vec_in, vec_out, inverse=False):
    """device-resident form: MultiLinearPolynomial handles double as vectors of 2^n elements."""
    check(lib.zk_ntt(ctx._h, vec_in._h, int(inverse), vec_out._h))
    return vec_out


def batch_last_stats():
    """(launches merged for all proofs, launches replayed proof by proof) of this thread's last prove_partial_batch"""
    a, b = c.c_uint64(), c.c_uint64()
    check(lib.zk_batch_last_stats(c.byref(a), c.byref(b)))
    return a.value, b.value


def bench_prove_partial(poly, max_var_degree, sum_, reps=10):
    """per-call wall clock (ms) of `reps` prove_partial calls measured inside the library with std::chrono (what a compiled host
    sees: no binding overhead)"""
    hp, keep = _handles(poly.polynomials)
    out = (c.c_double * reps)()
    check(lib.zk_bench_prove_partial(poly.ctx._h, hp, len(poly.polynomials), int(max_var_degree), _p(np.ascontiguousarray(sum_, dtype=np.uint64)),
                                     reps, out))
    return [float(v) for v in out]


def bench_evaluate(table, point, reps=10):
    """per-call wall clock (ms) of `reps` evaluate calls measured inside the library with std::chrono"""
    pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    out = (c.c_double * reps)()
    check(lib.zk_bench_evaluate(table.ctx._h, table._h, _p(pt if pt.size else np.zeros((1, 4), dtype=np.uint64)), pt.shape[0], reps, out))
    return [float(v) for v in out]


def bench_evaluate_device(table, point, reps=20):
    """average device time (ms) of one evaluate: `reps` back-to-back enqueues between two HIP events, no host wait in between"""
    pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    out = c.c_double()
    check(lib.zk_bench_evaluate_device(table.ctx._h, table._h, _p(pt if pt.size else np.zeros((1, 4), dtype=np.uint64)), pt.shape[0], reps, c.byref(out)))
    return out.value


def bench_ntt(ctx, vec_in, vec_out, inverse=False, reps=5):
    ms = c.c_double()
    check(lib.zk_bench_ntt(ctx._h, vec_in._h, int(inverse), vec_out._h, reps, c.byref(ms)))
    return ms.value


__all__ = [
    "BN254_FR", "BLS12_381_FR", "BLS12_377_FR", "Context", "MultiLinearPolynomial", "CoeffMultilinearPolynomial", "ProductPoly", "SumcheckProof",
    "SubClaim", "SumcheckProver", "SumcheckVerifier", "Transcript", "ZkError", "fft", "ifft", "fft_internal", "ntt", "bench_ntt", "bench_prove_partial", "batch_last_stats", "bench_evaluate", "bench_evaluate_device",
    "fe_from_int", "fe_from_ints", "fe_to_int", "fe_to_ints", "keccak256", "modulus", "two_adicity", "root_of_unity", "mask", "index_pair",
]
