//! zk-amd-shim — the reference's hot-path API (same names, same signatures, same `&'static str` errors) over the C ABI
//! of libzk_amd.so (`include/zk_amd.h`).  SOURCE ONLY: no Rust toolchain exists in the build image, so this file is the
//! documented binding a maintainer adds.  It is checked two ways without `rustc`: `tests/test_rust_shim_audit.py` parses
//! the `extern "C"` block below and compares every function's arity and parameter widths with `include/zk_amd.h`, and
//! `tests/cpp/test_reference_kats.cpp` runs the same call sequences through the C++ mirror (`zk_amd/host/zk.hpp`).
//!
//! Reference items mirrored (paths relative to the reference checkout):
//!   polynomial/src/multilinear/evaluation_form.rs:4-103  MultiLinearPolynomial<F>  (Clone, Debug, PartialEq)
//!   polynomial/src/multilinear/pairing_index.rs:2-9      index_pair
//!   polynomial/src/multilinear/pairing_index.rs:24-26    mask
//!   polynomial/src/product_poly.rs:6-88                  ProductPoly<F>            (Clone, Debug, PartialEq)
//!   sumcheck/src/prover.rs:9-73                          SumcheckProver<MAX_VAR_DEGREE, F>
//!   sumcheck/src/verifier.rs:9-41                        SumcheckVerifier<F>
//!   sumcheck/src/lib.rs:8-20                             SumcheckProof<F>, SubClaim<F>
//!   transcript/src/lib.rs:5-35                           Transcript
//!   fft/src/lib.rs:4-46                                  fft, ifft, fft_internal
//!
//! With this crate the four tests at sumcheck/src/lib.rs:53-122 read unchanged apart from their `use` lines
//! (`polynomial::…::MultiLinearPolynomial` / `ProductPoly`, `crate::prover::SumcheckProver`,
//! `crate::verifier::SumcheckVerifier` -> `zk_amd_shim::…`; `CoeffMultilinearPolynomial` stays the reference's).  The
//! reference's tests sit inside the `sumcheck` crate and read the private fields `subclaim.challenges` / `subclaim.sum`;
//! here those fields are `pub` for the same reason.
//!
//! `F` must be one of the 4-limb Montgomery fields libzk_amd knows; ark-ff's in-memory layout of
//! `Fp<MontBackend<_, 4>>` (4 LE u64 limbs, Montgomery form) IS the wire format, so `Vec<F>` is passed by pointer.
//!
//! Contexts: ONE `zk_ctx` per (thread, field), created on first use and shared by reference count — every table holds an
//! `Rc` of its context, so the context (stream, scratch, block pool) is destroyed exactly when the thread's cache entry
//! and the last table are gone.  `Rc` and raw pointers make every type here `!Send`/`!Sync`: a table cannot leave the
//! thread whose context owns it, which is the C ABI's threading rule.  Device: `ZK_AMD_DEVICE` (default 0).
use ark_ff::PrimeField;
use std::cell::{OnceCell, RefCell};
use std::fmt;
use std::marker::PhantomData;
use std::os::raw::c_char;
use std::rc::Rc;

#[allow(non_camel_case_types)]
#[repr(C)]
pub struct zk_ctx { _opaque: [u8; 0] }
#[allow(non_camel_case_types)]
#[repr(C)]
pub struct zk_mle { _opaque: [u8; 0] }
#[allow(non_camel_case_types)]
#[repr(C)]
pub struct zk_transcript { _opaque: [u8; 0] }
#[allow(non_camel_case_types)]
#[repr(C)]
pub struct zk_circuit { _opaque: [u8; 0] }

const ZK_ERR_EMPTY_PRODUCT: i32 = -3;
const ZK_ERR_ARITY_MISMATCH: i32 = -4;
const ZK_ERR_BAD_ARG: i32 = -20;
const ZK_ERR_PANIC_INDEX: i32 = -5;
const ZK_ERR_FFT_NOT_POW2: i32 = -6;
const ZK_ERR_FFT_NO_ROOT: i32 = -7;
const ZK_ERR_VERIFY_SUM: i32 = -9;
const ZK_ERR_GKR_REJECT: i32 = -27;

/// The ABI revision this file was written against (`ZK_AMD_ABI_VERSION` in include/zk_amd.h); checked once per context.
const ZK_AMD_ABI_VERSION: i32 = 6;

extern "C" {
    fn zk_abi_version() -> i32;
    fn zk_strerror(status: i32) -> *const c_char;
    fn zk_ctx_create(field: i32, device: i32, out_ctx: *mut *mut zk_ctx) -> i32;
    fn zk_ctx_destroy(ctx: *mut zk_ctx) -> i32;
    fn zk_mle_upload(ctx: *mut zk_ctx, n_vars: u64, evals: *const u64, len: u64, out: *mut *mut zk_mle) -> i32;
    fn zk_mle_clone(ctx: *mut zk_ctx, t: *const zk_mle, out: *mut *mut zk_mle) -> i32;
    fn zk_mle_free(ctx: *mut zk_ctx, t: *mut zk_mle) -> i32;
    fn zk_mle_n_vars(t: *const zk_mle, out_n_vars: *mut u64) -> i32;
    fn zk_mle_download(ctx: *mut zk_ctx, t: *const zk_mle, out_evals: *mut u64) -> i32;
    fn zk_mle_equal(ctx: *mut zk_ctx, a: *const zk_mle, b: *const zk_mle, out_equal: *mut i32) -> i32;
    fn zk_mle_partial_evaluate(ctx: *mut zk_ctx, t: *const zk_mle, initial_var: u64, assignments: *const u64,
                               n_assign: u64, out: *mut *mut zk_mle) -> i32;
    fn zk_mle_evaluate(ctx: *mut zk_ctx, t: *const zk_mle, point: *const u64, n_point: u64, out: *mut u64) -> i32;
    fn zk_mle_to_bytes(ctx: *mut zk_ctx, t: *const zk_mle, out_bytes: *mut u8) -> i32;
    fn zk_product_check(factors: *const *const zk_mle, k: u64) -> i32;
    fn zk_prod_reduce(ctx: *mut zk_ctx, factors: *const *const zk_mle, k: u64, out: *mut *mut zk_mle) -> i32;
    fn zk_product_evaluate(ctx: *mut zk_ctx, factors: *const *const zk_mle, k: u64, point: *const u64, n_point: u64,
                           out: *mut u64) -> i32;
    fn zk_transcript_new(out: *mut *mut zk_transcript) -> i32;
    fn zk_transcript_free(t: *mut zk_transcript) -> i32;
    fn zk_transcript_append(t: *mut zk_transcript, data: *const u8, len: usize) -> i32;
    fn zk_transcript_sample_field_element(t: *mut zk_transcript, field: i32, out: *mut u64) -> i32;
    fn zk_transcript_sample_n_field_elements(t: *mut zk_transcript, field: i32, n: u64, out: *mut u64) -> i32;
    fn zk_sumcheck_prove(ctx: *mut zk_ctx, factors: *const *mut zk_mle, k: u64, max_var_degree: u32, sum: *const u64,
                         absorb_table: i32, consume: i32, out_round_polys: *mut u64, out_challenges: *mut u64) -> i32;
    fn zk_sumcheck_prove_batch(ctx: *mut zk_ctx, n_proofs: u64, factors: *const *mut zk_mle, k: u64, max_var_degree: u32, sums: *const u64,
                               consume: i32, out_round_polys: *mut u64, out_challenges: *mut u64) -> i32;
    fn zk_sumcheck_verify_partial(field: i32, n_rounds: u64, max_var_degree: u32, sum: *const u64, round_polys: *const u64,
                                  out_subclaim_sum: *mut u64, out_challenges: *mut u64) -> i32;
    fn zk_sumcheck_verify(ctx: *mut zk_ctx, factors: *const *const zk_mle, k: u64, n_round_polys: u64, max_var_degree: u32,
                          sum: *const u64, round_polys: *const u64, out_ok: *mut i32) -> i32;
    fn zk_sumcheck_verify_partial_lengths(field: i32, n_rounds: u64, evals_per_round: *const u32, sum: *const u64,
                                          round_polys: *const u64, out_subclaim_sum: *mut u64, out_challenges: *mut u64) -> i32;
    fn zk_sumcheck_verify_lengths(ctx: *mut zk_ctx, factors: *const *const zk_mle, k: u64, n_round_polys: u64,
                                  evals_per_round: *const u32, sum: *const u64, round_polys: *const u64, out_ok: *mut i32) -> i32;
    fn zk_fft_host(ctx: *mut zk_ctx, input: *const u64, n: u64, out: *mut u64) -> i32;
    fn zk_ifft_host(ctx: *mut zk_ctx, input: *const u64, n: u64, out: *mut u64) -> i32;
    fn zk_fft_internal_host(ctx: *mut zk_ctx, input: *const u64, n: u64, omega: *const u64, out: *mut u64) -> i32;
    // GKR-shaped driver (no reference crate; include/zk_amd.h "sum of products + GKR-shaped driver")
    fn zk_sumcheck_prove_terms(ctx: *mut zk_ctx, factors: *const *mut zk_mle, term_k: *const u64, n_terms: u64,
                               max_var_degree: u32, sum: *const u64, consume: i32, out_round_polys: *mut u64,
                               out_challenges: *mut u64, out_final: *mut u64) -> i32;
    fn zk_circuit_create(ctx: *mut zk_ctx, out: *mut *mut zk_circuit) -> i32;
    fn zk_circuit_add_layer(c: *mut zk_circuit, log_out: u64, log_in: u64, op: *const u8, left: *const u32,
                            right: *const u32) -> i32;
    fn zk_circuit_free(c: *mut zk_circuit) -> i32;
    fn zk_circuit_proof_elems(c: *const zk_circuit, out: *mut u64) -> i32;
    fn zk_gkr_prove(c: *const zk_circuit, input: *const zk_mle, seed: *const u8, out_outputs: *mut *mut zk_mle,
                    out_proof: *mut u64) -> i32;
    fn zk_gkr_verify(c: *const zk_circuit, input: *const zk_mle, outputs: *const zk_mle, seed: *const u8,
                     proof: *const u64) -> i32;
}

/// Maps an arkworks field onto libzk_amd's `zk_field` enum (only 4-limb Montgomery fields qualify).
pub trait GpuField: PrimeField {
    const ZK_FIELD: i32;
}
impl GpuField for ark_bn254::Fr { const ZK_FIELD: i32 = 0; }
impl GpuField for ark_bls12_381::Fr { const ZK_FIELD: i32 = 1; }
impl GpuField for ark_bls12_377::Fr { const ZK_FIELD: i32 = 2; }
// `limbs()` below hands `&[F]` to C as `*const u64`: that is only sound while an element IS four u64 limbs.  Checked at
// compile time for every field the library is bound to (polynomial/src/multilinear/evaluation_form.rs:7-10 stores Vec<F>).
const _: () = assert!(std::mem::size_of::<ark_bn254::Fr>() == 32 && std::mem::align_of::<ark_bn254::Fr>() == 8);
const _: () = assert!(std::mem::size_of::<ark_bls12_381::Fr>() == 32 && std::mem::align_of::<ark_bls12_381::Fr>() == 8);
const _: () = assert!(std::mem::size_of::<ark_bls12_377::Fr>() == 32 && std::mem::align_of::<ark_bls12_377::Fr>() == 8);

fn err(status: i32) -> &'static str {
    // zk_strerror returns pointers to static strings that reproduce the reference's own messages
    unsafe { std::ffi::CStr::from_ptr(zk_strerror(status)).to_str().unwrap_or("zk_amd error") }
}

/// One `zk_ctx` (device + stream + scratch + block pool); destroyed with its last `Rc`.
struct Ctx { raw: *mut zk_ctx }
impl Drop for Ctx {
    fn drop(&mut self) { unsafe { zk_ctx_destroy(self.raw); } }
}
thread_local! {
    static CTXS: RefCell<[Option<Rc<Ctx>>; 3]> = const { RefCell::new([None, None, None]) };
}
/// The calling thread's context for field `F`: created on first use, then shared (tables built independently by `new`
/// therefore always belong to one context and can be combined in a `ProductPoly`).
fn ctx<F: GpuField>() -> Result<Rc<Ctx>, &'static str> {
    CTXS.with(|cell| {
        let mut slots = cell.borrow_mut();
        let slot = &mut slots[F::ZK_FIELD as usize];
        if let Some(c) = slot { return Ok(Rc::clone(c)); }
        // a libzk_amd.so of another ABI revision must not be driven through these declarations
        if unsafe { zk_abi_version() } != ZK_AMD_ABI_VERSION { return Err("zk_amd: libzk_amd.so ABI version mismatch"); }
        let device = std::env::var("ZK_AMD_DEVICE").ok().and_then(|s| s.parse::<i32>().ok()).unwrap_or(0);
        let mut raw: *mut zk_ctx = std::ptr::null_mut();
        let rc = unsafe { zk_ctx_create(F::ZK_FIELD, device, &mut raw) };
        if rc != 0 { return Err(err(rc)); }
        let c = Rc::new(Ctx { raw });
        *slot = Some(Rc::clone(&c));
        Ok(c)
    })
}
/// polynomial/src/multilinear/pairing_index.rs:24-26 — a bit sequence of n ones (`mask(1) -> 1`, `mask(3) -> 0b111`); public in the
/// reference, so part of the surface a caller may use.  Overflows (a panic in debug builds) at n >= usize::BITS like the original.
pub const fn mask(n: u8) -> usize {
    (1 << n) - 1
}

/// `&[F]` as the limb array the C ABI expects (ark-ff stores exactly this).
fn limbs<F: GpuField>(v: &[F]) -> *const u64 { v.as_ptr() as *const u64 }
fn limbs_mut<F: GpuField>(v: &mut [F]) -> *mut u64 { v.as_mut_ptr() as *mut u64 }

/// polynomial/src/multilinear/pairing_index.rs:2-9 — the (left, right) index pairs of one hypercube direction.
/// Host index arithmetic only (the kernels compute the same indices inline); panics on underflow like the reference.
pub fn index_pair(n_vars: u8, index: u8) -> impl Iterator<Item = (usize, usize)> {
    let pos = n_vars - 1 - index;
    (0..1usize << (n_vars - 1)).map(move |j| {
        let left = ((j >> pos) << (pos + 1)) | (j & mask(pos));
        (left, left | (1usize << pos))
    })
}

/// polynomial::multilinear::evaluation_form::MultiLinearPolynomial (evaluation_form.rs:7-10), table resident in HBM.
pub struct MultiLinearPolynomial<F: GpuField> {
    ctx: Rc<Ctx>,
    h: *mut zk_mle,
    n_vars: usize,
    host: OnceCell<Vec<F>>, // lazily downloaded mirror backing `evaluation_slice`; tables are immutable once built
}

impl<F: GpuField> Drop for MultiLinearPolynomial<F> {
    fn drop(&mut self) { unsafe { zk_mle_free(self.ctx.raw, self.h); } }
}
/// #[derive(Clone)] evaluation_form.rs:4 — a device-to-device copy
impl<F: GpuField> Clone for MultiLinearPolynomial<F> {
    fn clone(&self) -> Self {
        let mut h: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_mle_clone(self.ctx.raw, self.h, &mut h) };
        assert!(rc == 0, "{}", err(rc));
        Self::from_handle(Rc::clone(&self.ctx), h)
    }
}
/// #[derive(PartialEq)] evaluation_form.rs:4 — n_vars and all evaluations, compared on the device
impl<F: GpuField> PartialEq for MultiLinearPolynomial<F> {
    fn eq(&self, other: &Self) -> bool {
        if self.n_vars != other.n_vars { return false; }
        if !Rc::ptr_eq(&self.ctx, &other.ctx) { return self.evaluation_slice() == other.evaluation_slice(); }
        let mut eq = 0i32;
        let rc = unsafe { zk_mle_equal(self.ctx.raw, self.h, other.h, &mut eq) };
        assert!(rc == 0, "{}", err(rc));
        eq != 0
    }
}
/// #[derive(Debug)] evaluation_form.rs:4
impl<F: GpuField> fmt::Debug for MultiLinearPolynomial<F> {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.debug_struct("MultiLinearPolynomial").field("n_vars", &self.n_vars).field("evaluations", &self.evaluation_slice()).finish()
    }
}
impl<F: GpuField> MultiLinearPolynomial<F> {
    fn from_handle(ctx: Rc<Ctx>, h: *mut zk_mle) -> Self {
        let mut n = 0u64;
        unsafe { zk_mle_n_vars(h, &mut n); }
        Self { ctx, h, n_vars: n as usize, host: OnceCell::new() }
    }
    /// evaluation_form.rs:15-27
    pub fn new(n_vars: usize, evaluations: Vec<F>) -> Result<Self, &'static str> {
        let c = ctx::<F>()?;
        let mut h: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_mle_upload(c.raw, n_vars as u64, limbs(&evaluations), evaluations.len() as u64, &mut h) };
        if rc != 0 { return Err(err(rc)); } // "evaluation vec len should equal 2^n_vars"
        Ok(Self { ctx: c, h, n_vars, host: OnceCell::from(evaluations) }) // the caller's Vec IS the host mirror
    }
    /// evaluation_form.rs:30
    pub fn n_vars(&self) -> usize { self.n_vars }
    /// evaluation_form.rs:40-80
    pub fn partial_evaluate(&self, initial_var: usize, assignments: &[F]) -> Result<Self, &'static str> {
        let mut h: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_mle_partial_evaluate(self.ctx.raw, self.h, initial_var as u64, limbs(assignments),
                                                  assignments.len() as u64, &mut h) };
        if rc == ZK_ERR_PANIC_INDEX { panic!("{}", err(rc)); } // the reference panics here (u8 / usize underflow, :55,:75)
        if rc != 0 { return Err(err(rc)); }
        Ok(Self::from_handle(Rc::clone(&self.ctx), h))
    }
    /// evaluation_form.rs:83-89
    pub fn evaluate(&self, assignments: &[F]) -> Result<F, &'static str> {
        let mut out = [F::zero()];
        let rc = unsafe { zk_mle_evaluate(self.ctx.raw, self.h, limbs(assignments), assignments.len() as u64, limbs_mut(&mut out)) };
        if rc != 0 { return Err(err(rc)); } // "evaluate must assign to all variables"
        Ok(out[0])
    }
    /// evaluation_form.rs:92-94 — a borrowed slice, like the reference; the first call on a table that was produced on the
    /// device downloads it once (2^n_vars * 32 bytes over PCIe): keep tables on the device where speed matters.
    pub fn evaluation_slice(&self) -> &[F] {
        self.host.get_or_init(|| {
            let mut v = vec![F::zero(); 1usize << self.n_vars];
            let rc = unsafe { zk_mle_download(self.ctx.raw, self.h, limbs_mut(&mut v)) };
            assert!(rc == 0, "{}", err(rc));
            v
        })
    }
    /// evaluation_form.rs:97-103
    pub fn to_bytes(&self) -> Vec<u8> {
        let mut b = vec![0u8; 32usize << self.n_vars];
        let rc = unsafe { zk_mle_to_bytes(self.ctx.raw, self.h, b.as_mut_ptr()) };
        assert!(rc == 0, "{}", err(rc));
        b
    }
}

/// polynomial::product_poly::ProductPoly (product_poly.rs:6-10)
#[derive(Clone, Debug, PartialEq)]
pub struct ProductPoly<F: GpuField> { n_vars: usize, polynomials: Vec<MultiLinearPolynomial<F>> }

impl<F: GpuField> ProductPoly<F> {
    fn handles(&self) -> Vec<*const zk_mle> { self.polynomials.iter().map(|p| p.h as *const zk_mle).collect() }
    fn ctx_raw(&self) -> *mut zk_ctx { self.polynomials[0].ctx.raw }
    /// product_poly.rs:14-32
    pub fn new(polynomials: Vec<MultiLinearPolynomial<F>>) -> Result<Self, &'static str> {
        let h: Vec<*const zk_mle> = polynomials.iter().map(|p| p.h as *const zk_mle).collect();
        let rc = unsafe { zk_product_check(h.as_ptr(), h.len() as u64) };
        if rc != 0 { return Err(err(rc)); } // the two Err texts of product_poly.rs:16,25
        Ok(Self { n_vars: polynomials[0].n_vars(), polynomials })
    }
    /// product_poly.rs:86
    pub fn n_vars(&self) -> usize { self.n_vars }
    /// product_poly.rs:36-44
    pub fn evaluate(&self, assignments: &[F]) -> Result<F, &'static str> {
        let h = self.handles();
        let mut out = [F::zero()];
        let rc = unsafe { zk_product_evaluate(self.ctx_raw(), h.as_ptr(), h.len() as u64, limbs(assignments),
                                              assignments.len() as u64, limbs_mut(&mut out)) };
        if rc != 0 { return Err(err(rc)); }
        Ok(out[0])
    }
    /// product_poly.rs:48-63
    pub fn partial_evaluate(&self, initial_var: usize, assignments: &[F]) -> Result<Self, &'static str> {
        let polynomials = self.polynomials.iter().map(|p| p.partial_evaluate(initial_var, assignments))
            .collect::<Result<Vec<_>, _>>()?;
        Ok(Self { n_vars: polynomials[0].n_vars(), polynomials })
    }
    /// product_poly.rs:66-74
    pub fn prod_reduce(&self) -> Vec<F> {
        let h = self.handles();
        let mut o: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_prod_reduce(self.ctx_raw(), h.as_ptr(), h.len() as u64, &mut o) };
        assert!(rc == 0, "{}", err(rc));
        MultiLinearPolynomial::<F>::from_handle(Rc::clone(&self.polynomials[0].ctx), o).evaluation_slice().to_vec()
    }
    /// product_poly.rs:77-83
    pub fn to_bytes(&self) -> Vec<u8> { self.polynomials.iter().flat_map(|p| p.to_bytes()).collect() }
}

/// transcript::Transcript (transcript/src/lib.rs:5-35): Keccak-256 sponge, host side.
pub struct Transcript { h: *mut zk_transcript }
impl Drop for Transcript {
    fn drop(&mut self) { unsafe { zk_transcript_free(self.h); } }
}
impl Transcript {
    /// transcript/src/lib.rs:10-14
    pub fn new() -> Self {
        let mut h: *mut zk_transcript = std::ptr::null_mut();
        let rc = unsafe { zk_transcript_new(&mut h) };
        assert!(rc == 0, "{}", err(rc));
        Self { h }
    }
    /// transcript/src/lib.rs:16-18
    pub fn append(&mut self, new_data: &[u8]) {
        let rc = unsafe { zk_transcript_append(self.h, new_data.as_ptr(), new_data.len()) };
        assert!(rc == 0, "{}", err(rc));
    }
    /// transcript/src/lib.rs:27-30
    pub fn sample_field_element<F: GpuField>(&mut self) -> F {
        let mut out = [F::zero()];
        let rc = unsafe { zk_transcript_sample_field_element(self.h, F::ZK_FIELD, limbs_mut(&mut out)) };
        assert!(rc == 0, "{}", err(rc));
        out[0]
    }
    /// transcript/src/lib.rs:32-34
    pub fn sample_n_field_elements<F: GpuField>(&mut self, n: usize) -> Vec<F> {
        let mut out = vec![F::zero(); n];
        let rc = unsafe { zk_transcript_sample_n_field_elements(self.h, F::ZK_FIELD, n as u64, limbs_mut(&mut out)) };
        assert!(rc == 0, "{}", err(rc));
        out
    }
}

/// sumcheck::SumcheckProof (sumcheck/src/lib.rs:6-11)
#[derive(Debug)]
pub struct SumcheckProof<F: PrimeField> { pub sum: F, pub round_polys: Vec<Vec<F>> }

/// sumcheck::SubClaim (sumcheck/src/lib.rs:17-20): sum = initial_poly(challenges) is the check left to the caller
pub struct SubClaim<F: PrimeField> { pub sum: F, pub challenges: Vec<F> }

/// sumcheck::prover::SumcheckProver (prover.rs:9-12)
pub struct SumcheckProver<const MAX_VAR_DEGREE: u8, F: GpuField> { _marker: PhantomData<F> }

impl<const MAX_VAR_DEGREE: u8, F: GpuField> SumcheckProver<MAX_VAR_DEGREE, F> {
    fn run(poly: ProductPoly<F>, sum: F, absorb: i32) -> Result<(SumcheckProof<F>, Vec<F>), &'static str> {
        let n = poly.n_vars();
        let ns = MAX_VAR_DEGREE as usize + 1;
        let h: Vec<*mut zk_mle> = poly.polynomials.iter().map(|p| p.h).collect();
        let mut rp = vec![F::zero(); n * ns];
        let mut ch = vec![F::zero(); n];
        let s = [sum];
        // the reference takes `poly` by value: the library may use its tables as scratch (consume = 1); they are freed when
        // `poly` drops at the end of this function
        let rc = unsafe { zk_sumcheck_prove(poly.ctx_raw(), h.as_ptr(), h.len() as u64, MAX_VAR_DEGREE as u32, limbs(&s), absorb, 1,
                                            limbs_mut(&mut rp), limbs_mut(&mut ch)) };
        if rc != 0 { return Err(err(rc)); }
        let round_polys = rp.chunks(ns).map(|c| c.to_vec()).collect();
        Ok((SumcheckProof { sum, round_polys }, ch))
    }
    /// prover.rs:15-20
    pub fn prove(poly: ProductPoly<F>, sum: F) -> Result<SumcheckProof<F>, &'static str> { Ok(Self::run(poly, sum, 1)?.0) }
    /// prover.rs:24-30
    pub fn prove_partial(poly: ProductPoly<F>, sum: F) -> Result<(SumcheckProof<F>, Vec<F>), &'static str> {
        Self::run(poly, sum, 0)
    }
    /// `polys.len()` independent `prove_partial` calls (same factor count and arity) proved side by side: one kernel launch per
    /// round for all of them (zk_sumcheck_prove_batch).  Element i equals `prove_partial(polys[i], sums[i])`, bit for bit.
    pub fn prove_partial_batch(polys: Vec<ProductPoly<F>>, sums: Vec<F>) -> Result<Vec<(SumcheckProof<F>, Vec<F>)>, &'static str> {
        if polys.len() != sums.len() { return Err(err(ZK_ERR_BAD_ARG)); }
        if polys.is_empty() { return Ok(Vec::new()); }
        let (b, k, n, ns) = (polys.len(), polys[0].handles().len(), polys[0].n_vars(), MAX_VAR_DEGREE as usize + 1);
        let mut h: Vec<*mut zk_mle> = Vec::with_capacity(b * k);
        for p in &polys {
            let hp = p.handles();
            if hp.len() != k { return Err(err(ZK_ERR_ARITY_MISMATCH)); }
            h.extend(hp.iter().map(|x| *x as *mut zk_mle));
        }
        let (mut rp, mut ch) = (vec![F::zero(); b * n * ns + 1], vec![F::zero(); b * n + 1]);
        let rc = unsafe { zk_sumcheck_prove_batch(polys[0].ctx_raw(), b as u64, h.as_ptr(), k as u64, MAX_VAR_DEGREE as u32, limbs(&sums), 0,
                                                  limbs_mut(&mut rp), limbs_mut(&mut ch)) };
        if rc != 0 { return Err(err(rc)); }
        Ok((0..b).map(|i| (SumcheckProof { sum: sums[i], round_polys: rp[i * n * ns..(i + 1) * n * ns].chunks(ns).map(|c| c.to_vec()).collect() },
                           ch[i * n..(i + 1) * n].to_vec())).collect())
    }
}

/// sumcheck::verifier::SumcheckVerifier (verifier.rs:9-11)
pub struct SumcheckVerifier<F: GpuField> { _marker: PhantomData<F> }

impl<F: GpuField> SumcheckVerifier<F> {
    /// `round_polys` is a `Vec<Vec<F>>` and the reference interpolates every round at its own length (verifier.rs:55-58):
    /// the evaluations go to the library back to back with one length per round.
    fn flatten(proof: &SumcheckProof<F>) -> (Vec<F>, Vec<u32>) {
        let mut lens: Vec<u32> = proof.round_polys.iter().map(|r| r.len() as u32).collect();
        lens.push(0);
        let mut rps: Vec<F> = proof.round_polys.iter().flatten().copied().collect();
        rps.push(F::zero());
        (rps, lens)
    }
    /// verifier.rs:15-33
    pub fn verify(poly: ProductPoly<F>, proof: SumcheckProof<F>) -> Result<bool, &'static str> {
        let (rps, lens) = Self::flatten(&proof);
        let h = poly.handles();
        let s = [proof.sum];
        let mut ok = 0i32;
        let rc = unsafe { zk_sumcheck_verify_lengths(poly.ctx_raw(), h.as_ptr(), h.len() as u64, proof.round_polys.len() as u64,
                                                     lens.as_ptr(), limbs(&s), limbs(&rps), &mut ok) };
        if rc != 0 { return Err(err(rc)); } // "invalid proof: require 1 round poly ..." / "verifier check failed: ..."
        Ok(ok != 0)
    }
    /// verifier.rs:38-41
    pub fn verify_partial(proof: SumcheckProof<F>) -> Result<SubClaim<F>, &'static str> {
        let (rps, lens) = Self::flatten(&proof);
        let n = proof.round_polys.len();
        let s = [proof.sum];
        let mut sum = [F::zero()];
        let mut challenges = vec![F::zero(); n.max(1)];
        let rc = unsafe { zk_sumcheck_verify_partial_lengths(F::ZK_FIELD, n as u64, lens.as_ptr(), limbs(&s), limbs(&rps),
                                                             limbs_mut(&mut sum), limbs_mut(&mut challenges)) };
        if rc != 0 { return Err(err(rc)); }
        challenges.truncate(n);
        Ok(SubClaim { sum: sum[0], challenges })
    }
}

/// fft/src/lib.rs:4-8 (panics like the reference when the length has no root of unity)
pub fn fft<F: GpuField>(coefficients: Vec<F>) -> Vec<F> {
    let c = ctx::<F>().unwrap_or_else(|e| panic!("{}", e));
    let mut out = vec![F::zero(); coefficients.len()];
    let rc = unsafe { zk_fft_host(c.raw, limbs(&coefficients), coefficients.len() as u64, limbs_mut(&mut out)) };
    assert!(rc == 0, "{}", err(rc));
    out
}
/// fft/src/lib.rs:11-19
pub fn ifft<F: GpuField>(evaluations: Vec<F>) -> Vec<F> {
    let c = ctx::<F>().unwrap_or_else(|e| panic!("{}", e));
    let mut out = vec![F::zero(); evaluations.len()];
    let rc = unsafe { zk_ifft_host(c.raw, limbs(&evaluations), evaluations.len() as u64, limbs_mut(&mut out)) };
    assert!(rc == 0, "{}", err(rc));
    out
}
/// fft/src/lib.rs:21-46 — any omega (for a non-primitive one the library uses the literal omega^(i + n/2) form);
/// panics with "values must be a power of 2" like the reference
pub fn fft_internal<F: GpuField>(values: Vec<F>, omega: F) -> Vec<F> {
    let c = ctx::<F>().unwrap_or_else(|e| panic!("{}", e));
    let mut out = vec![F::zero(); values.len()];
    let w = [omega];
    let rc = unsafe { zk_fft_internal_host(c.raw, limbs(&values), values.len() as u64, limbs(&w), limbs_mut(&mut out)) };
    assert!(rc != ZK_ERR_FFT_NOT_POW2 && rc != ZK_ERR_FFT_NO_ROOT, "{}", err(rc));
    assert!(rc == 0, "{}", err(rc));
    out
}

// ---- GKR-shaped driver ---------------------------------------------------------------------------------------------
// The reference has no gkr crate (its building block is prove_partial, prover.rs:24-30; intent at
// evaluation_form.rs:45-48).  These are the bindings of this library's own layered driver.

/// prove_partial on sum_i prod_{f in terms[i]}: (round polys, challenges, every factor at the challenge point)
pub fn prove_partial_terms<const MAX_VAR_DEGREE: u8, F: GpuField>(terms: &[Vec<MultiLinearPolynomial<F>>], sum: F)
    -> Result<(SumcheckProof<F>, Vec<F>, Vec<F>), &'static str> {
    let h: Vec<*mut zk_mle> = terms.iter().flat_map(|t| t.iter().map(|p| p.h)).collect();
    let tk: Vec<u64> = terms.iter().map(|t| t.len() as u64).collect();
    if h.is_empty() { return Err(err(ZK_ERR_EMPTY_PRODUCT)); }
    let first = terms.iter().find(|t| !t.is_empty()).unwrap();
    let n = first[0].n_vars();
    let ns = MAX_VAR_DEGREE as usize + 1;
    let (mut rp, mut ch, mut fin) = (vec![F::zero(); n * ns], vec![F::zero(); n], vec![F::zero(); h.len()]);
    let s = [sum];
    let rc = unsafe { zk_sumcheck_prove_terms(first[0].ctx.raw, h.as_ptr(), tk.as_ptr(), tk.len() as u64, MAX_VAR_DEGREE as u32,
                                              limbs(&s), 0, limbs_mut(&mut rp), limbs_mut(&mut ch), limbs_mut(&mut fin)) };
    if rc != 0 { return Err(err(rc)); }
    Ok((SumcheckProof { sum, round_polys: rp.chunks(ns).map(|c| c.to_vec()).collect() }, ch, fin))
}

/// One layer: 2^log_out gates (op 0 = add, 1 = mul) over the 2^log_in values of the layer below.
pub struct Layer { pub log_out: usize, pub log_in: usize, pub op: Vec<u8>, pub left: Vec<u32>, pub right: Vec<u32> }

pub struct Circuit<F: GpuField> { ctx: Rc<Ctx>, h: *mut zk_circuit, _f: PhantomData<F> }
impl<F: GpuField> Drop for Circuit<F> { fn drop(&mut self) { unsafe { zk_circuit_free(self.h); } } }
impl<F: GpuField> Circuit<F> {
    /// layers[0] = output layer.  Uses the thread's context for `F`, the one every `MultiLinearPolynomial<F>` lives in.
    pub fn new(layers: &[Layer]) -> Result<Self, &'static str> {
        let c = ctx::<F>()?;
        let mut h: *mut zk_circuit = std::ptr::null_mut();
        let rc = unsafe { zk_circuit_create(c.raw, &mut h) };
        if rc != 0 { return Err(err(rc)); }
        let circuit = Circuit { ctx: c, h, _f: PhantomData };
        for l in layers {
            let gates = 1usize << l.log_out;
            if l.op.len() != gates || l.left.len() != gates || l.right.len() != gates { return Err(err(-20)); }
            let rc = unsafe { zk_circuit_add_layer(h, l.log_out as u64, l.log_in as u64, l.op.as_ptr(), l.left.as_ptr(), l.right.as_ptr()) };
            if rc != 0 { return Err(err(rc)); }
        }
        Ok(circuit)
    }
    fn proof_elems(&self) -> usize {
        let mut n = 0u64;
        unsafe { zk_circuit_proof_elems(self.h, &mut n); }
        n as usize
    }
    /// -> (outputs, proof elements: per layer [round polys #1 | round polys #2 | W(u) | W(v)])
    pub fn prove(&self, input: &MultiLinearPolynomial<F>, seed: &[u8; 32]) -> Result<(MultiLinearPolynomial<F>, Vec<F>), &'static str> {
        let mut proof = vec![F::zero(); self.proof_elems()];
        let mut out: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_gkr_prove(self.h, input.h, seed.as_ptr(), &mut out, limbs_mut(&mut proof)) };
        if rc != 0 { return Err(err(rc)); }
        Ok((MultiLinearPolynomial::from_handle(Rc::clone(&self.ctx), out), proof))
    }
    /// Ok(true) accept / Ok(false) reject (a sumcheck round check, a layer's wiring check or the input check failed)
    pub fn verify(&self, input: &MultiLinearPolynomial<F>, outputs: &MultiLinearPolynomial<F>, seed: &[u8; 32], proof: &[F])
        -> Result<bool, &'static str> {
        if proof.len() != self.proof_elems() { return Ok(false); } // the library reads exactly proof_elems elements
        match unsafe { zk_gkr_verify(self.h, input.h, outputs.h, seed.as_ptr(), limbs(proof)) } {
            0 => Ok(true),
            ZK_ERR_VERIFY_SUM | ZK_ERR_GKR_REJECT => Ok(false),
            rc => Err(err(rc)),
        }
    }
}
