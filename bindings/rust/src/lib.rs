//! zk-amd-shim — the reference's hot-path API (same names, same signatures, same `&'static str` errors) over the C ABI
//! of libzk_amd.so (`include/zk_amd.h`).  SOURCE ONLY: no Rust toolchain exists in the build image, so this file is the
//! documented binding a maintainer adds, not something the test-suite compiles.
//!
//! Reference items mirrored (paths relative to the reference checkout):
//!   polynomial/src/multilinear/evaluation_form.rs:7-103  MultiLinearPolynomial<F>
//!   polynomial/src/product_poly.rs:7-88                  ProductPoly<F>
//!   sumcheck/src/prover.rs:9-73                          SumcheckProver<MAX_VAR_DEGREE, F>
//!   sumcheck/src/lib.rs:8-20                             SumcheckProof<F>, SubClaim<F>
//!   fft/src/lib.rs:4-19                                  fft, ifft
//!
//! `F` must be one of the 4-limb Montgomery fields libzk_amd knows; ark-ff's in-memory layout of
//! `Fp<MontBackend<_, 4>>` (4 LE u64 limbs, Montgomery form) IS the wire format, so `Vec<F>` is passed by pointer.
use ark_ff::PrimeField;
use std::marker::PhantomData;
use std::os::raw::{c_char, c_void};

#[allow(non_camel_case_types)]
type zk_ctx = c_void;
#[allow(non_camel_case_types)]
type zk_mle = c_void;
#[allow(non_camel_case_types)]
type zk_circuit = c_void;

extern "C" {
    fn zk_strerror(status: i32) -> *const c_char;
    fn zk_ctx_create(field: i32, device: i32, out: *mut *mut zk_ctx) -> i32;
    fn zk_mle_upload(ctx: *mut zk_ctx, n_vars: u64, evals: *const u64, len: u64, out: *mut *mut zk_mle) -> i32;
    fn zk_mle_free(ctx: *mut zk_ctx, t: *mut zk_mle) -> i32;
    fn zk_mle_n_vars(t: *const zk_mle, out: *mut u64) -> i32;
    fn zk_mle_download(ctx: *mut zk_ctx, t: *const zk_mle, out: *mut u64) -> i32;
    fn zk_mle_partial_evaluate(ctx: *mut zk_ctx, t: *const zk_mle, initial_var: u64, assignments: *const u64,
                               n_assign: u64, out: *mut *mut zk_mle) -> i32;
    fn zk_mle_evaluate(ctx: *mut zk_ctx, t: *const zk_mle, point: *const u64, n_point: u64, out: *mut u64) -> i32;
    fn zk_mle_to_bytes(ctx: *mut zk_ctx, t: *const zk_mle, out: *mut u8) -> i32;
    fn zk_product_check(factors: *const *const zk_mle, k: u64) -> i32;
    fn zk_prod_reduce(ctx: *mut zk_ctx, factors: *const *const zk_mle, k: u64, out: *mut *mut zk_mle) -> i32;
    fn zk_product_evaluate(ctx: *mut zk_ctx, factors: *const *const zk_mle, k: u64, point: *const u64, n_point: u64,
                           out: *mut u64) -> i32;
    fn zk_sumcheck_prove(ctx: *mut zk_ctx, factors: *const *mut zk_mle, k: u64, max_var_degree: u32, sum: *const u64,
                         absorb_table: i32, consume: i32, out_round_polys: *mut u64, out_challenges: *mut u64) -> i32;
    fn zk_fft_host(ctx: *mut zk_ctx, input: *const u64, n: u64, out: *mut u64) -> i32;
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
    fn zk_ifft_host(ctx: *mut zk_ctx, input: *const u64, n: u64, out: *mut u64) -> i32;
}

/// Maps an arkworks field onto libzk_amd's `zk_field` enum (sealed: only 4-limb Montgomery fields qualify).
pub trait GpuField: PrimeField {
    const ZK_FIELD: i32;
}
impl GpuField for ark_bn254::Fr { const ZK_FIELD: i32 = 0; }
impl GpuField for ark_bls12_381::Fr { const ZK_FIELD: i32 = 1; }
impl GpuField for ark_bls12_377::Fr { const ZK_FIELD: i32 = 2; }

fn err(status: i32) -> &'static str {
    // zk_strerror returns pointers to static strings that reproduce the reference's own messages
    unsafe { std::ffi::CStr::from_ptr(zk_strerror(status)).to_str().unwrap_or("zk_amd error") }
}
fn ctx<F: GpuField>() -> *mut zk_ctx {
    // one context per field per thread; a real integration would cache this in a thread_local
    let mut c: *mut zk_ctx = std::ptr::null_mut();
    let rc = unsafe { zk_ctx_create(F::ZK_FIELD, 0, &mut c) };
    assert!(rc == 0, "{}", err(rc));
    c
}
/// `&[F]` as the limb array the C ABI expects (ark-ff stores exactly this).
fn limbs<F: GpuField>(v: &[F]) -> *const u64 { v.as_ptr() as *const u64 }

/// polynomial::multilinear::evaluation_form::MultiLinearPolynomial (evaluation_form.rs:7-10), table resident in HBM.
pub struct MultiLinearPolynomial<F: GpuField> { ctx: *mut zk_ctx, h: *mut zk_mle, _f: PhantomData<F> }

impl<F: GpuField> Drop for MultiLinearPolynomial<F> {
    fn drop(&mut self) { unsafe { zk_mle_free(self.ctx, self.h); } }
}
impl<F: GpuField> MultiLinearPolynomial<F> {
    /// evaluation_form.rs:15-27
    pub fn new(n_vars: usize, evaluations: Vec<F>) -> Result<Self, &'static str> {
        let c = ctx::<F>();
        let mut h: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_mle_upload(c, n_vars as u64, limbs(&evaluations), evaluations.len() as u64, &mut h) };
        if rc != 0 { return Err(err(rc)); }
        Ok(Self { ctx: c, h, _f: PhantomData })
    }
    /// evaluation_form.rs:30
    pub fn n_vars(&self) -> usize { let mut n = 0u64; unsafe { zk_mle_n_vars(self.h, &mut n); } n as usize }
    /// evaluation_form.rs:40-80
    pub fn partial_evaluate(&self, initial_var: usize, assignments: &[F]) -> Result<Self, &'static str> {
        let mut h: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_mle_partial_evaluate(self.ctx, self.h, initial_var as u64, limbs(assignments),
                                                  assignments.len() as u64, &mut h) };
        if rc != 0 { return Err(err(rc)); }
        Ok(Self { ctx: self.ctx, h, _f: PhantomData })
    }
    /// evaluation_form.rs:83-89
    pub fn evaluate(&self, assignments: &[F]) -> Result<F, &'static str> {
        let mut out = F::zero();
        let rc = unsafe { zk_mle_evaluate(self.ctx, self.h, limbs(assignments), assignments.len() as u64,
                                          &mut out as *mut F as *mut u64) };
        if rc != 0 { return Err(err(rc)); }
        Ok(out)
    }
    /// evaluation_form.rs:92-94 — downloads (the reference returns a borrowed slice of host memory)
    pub fn evaluations(&self) -> Vec<F> {
        let mut v = vec![F::zero(); 1 << self.n_vars()];
        unsafe { zk_mle_download(self.ctx, self.h, v.as_mut_ptr() as *mut u64); }
        v
    }
    /// evaluation_form.rs:97-103
    pub fn to_bytes(&self) -> Vec<u8> {
        let mut b = vec![0u8; 32 << self.n_vars()];
        unsafe { zk_mle_to_bytes(self.ctx, self.h, b.as_mut_ptr()); }
        b
    }
}

/// polynomial::product_poly::ProductPoly (product_poly.rs:7-10)
pub struct ProductPoly<F: GpuField> { polynomials: Vec<MultiLinearPolynomial<F>> }

impl<F: GpuField> ProductPoly<F> {
    fn handles(&self) -> Vec<*const zk_mle> { self.polynomials.iter().map(|p| p.h as *const zk_mle).collect() }
    /// product_poly.rs:14-32
    pub fn new(polynomials: Vec<MultiLinearPolynomial<F>>) -> Result<Self, &'static str> {
        let h: Vec<*const zk_mle> = polynomials.iter().map(|p| p.h as *const zk_mle).collect();
        let rc = unsafe { zk_product_check(h.as_ptr(), h.len() as u64) };
        if rc != 0 { return Err(err(rc)); }
        Ok(Self { polynomials })
    }
    /// product_poly.rs:86
    pub fn n_vars(&self) -> usize { self.polynomials[0].n_vars() }
    /// product_poly.rs:36-44
    pub fn evaluate(&self, assignments: &[F]) -> Result<F, &'static str> {
        let h = self.handles();
        let mut out = F::zero();
        let rc = unsafe { zk_product_evaluate(self.polynomials[0].ctx, h.as_ptr(), h.len() as u64, limbs(assignments),
                                              assignments.len() as u64, &mut out as *mut F as *mut u64) };
        if rc != 0 { return Err(err(rc)); }
        Ok(out)
    }
    /// product_poly.rs:48-63
    pub fn partial_evaluate(&self, initial_var: usize, assignments: &[F]) -> Result<Self, &'static str> {
        let polynomials = self.polynomials.iter().map(|p| p.partial_evaluate(initial_var, assignments))
            .collect::<Result<Vec<_>, _>>()?;
        Ok(Self { polynomials })
    }
    /// product_poly.rs:66-74
    pub fn prod_reduce(&self) -> Vec<F> {
        let h = self.handles();
        let mut o: *mut zk_mle = std::ptr::null_mut();
        let c = self.polynomials[0].ctx;
        let rc = unsafe { zk_prod_reduce(c, h.as_ptr(), h.len() as u64, &mut o) };
        assert!(rc == 0, "{}", err(rc));
        let t = MultiLinearPolynomial::<F> { ctx: c, h: o, _f: PhantomData };
        t.evaluations()
    }
    /// product_poly.rs:77-83
    pub fn to_bytes(&self) -> Vec<u8> { self.polynomials.iter().flat_map(|p| p.to_bytes()).collect() }
}

/// sumcheck::SumcheckProof (sumcheck/src/lib.rs:8-11)
#[derive(Debug)]
pub struct SumcheckProof<F: PrimeField> { pub sum: F, pub round_polys: Vec<Vec<F>> }

/// sumcheck::prover::SumcheckProver (prover.rs:9-12)
pub struct SumcheckProver<const MAX_VAR_DEGREE: u8, F: GpuField> { _marker: PhantomData<F> }

impl<const MAX_VAR_DEGREE: u8, F: GpuField> SumcheckProver<MAX_VAR_DEGREE, F> {
    fn run(poly: ProductPoly<F>, sum: F, absorb: i32) -> Result<(SumcheckProof<F>, Vec<F>), &'static str> {
        let n = poly.n_vars();
        let ns = MAX_VAR_DEGREE as usize + 1;
        let h: Vec<*mut zk_mle> = poly.polynomials.iter().map(|p| p.h).collect();
        let mut rp = vec![F::zero(); n * ns];
        let mut ch = vec![F::zero(); n];
        // the reference takes `poly` by value: let the library reuse the tables as scratch (consume = 1)
        let rc = unsafe { zk_sumcheck_prove(poly.polynomials[0].ctx, h.as_ptr(), h.len() as u64, MAX_VAR_DEGREE as u32,
                                            &sum as *const F as *const u64, absorb, 1,
                                            rp.as_mut_ptr() as *mut u64, ch.as_mut_ptr() as *mut u64) };
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
}

/// fft/src/lib.rs:4-8 (panics like the reference when the length has no root of unity)
pub fn fft<F: GpuField>(coefficients: Vec<F>) -> Vec<F> {
    let mut out = vec![F::zero(); coefficients.len()];
    let rc = unsafe { zk_fft_host(ctx::<F>(), limbs(&coefficients), coefficients.len() as u64, out.as_mut_ptr() as *mut u64) };
    assert!(rc == 0, "{}", err(rc));
    out
}
/// fft/src/lib.rs:11-19
pub fn ifft<F: GpuField>(evaluations: Vec<F>) -> Vec<F> {
    let mut out = vec![F::zero(); evaluations.len()];
    let rc = unsafe { zk_ifft_host(ctx::<F>(), limbs(&evaluations), evaluations.len() as u64, out.as_mut_ptr() as *mut u64) };
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
    if h.is_empty() { return Err(err(-3)); }
    let n = terms[0][0].n_vars();
    let ns = MAX_VAR_DEGREE as usize + 1;
    let (mut rp, mut ch, mut fin) = (vec![F::zero(); n * ns], vec![F::zero(); n], vec![F::zero(); h.len()]);
    let rc = unsafe { zk_sumcheck_prove_terms(terms[0][0].ctx, h.as_ptr(), tk.as_ptr(), tk.len() as u64, MAX_VAR_DEGREE as u32,
                                              &sum as *const F as *const u64, 0, rp.as_mut_ptr() as *mut u64,
                                              ch.as_mut_ptr() as *mut u64, fin.as_mut_ptr() as *mut u64) };
    if rc != 0 { return Err(err(rc)); }
    Ok((SumcheckProof { sum, round_polys: rp.chunks(ns).map(|c| c.to_vec()).collect() }, ch, fin))
}

/// One layer: 2^log_out gates (op 0 = add, 1 = mul) over the 2^log_in values of the layer below.
pub struct Layer { pub log_out: usize, pub log_in: usize, pub op: Vec<u8>, pub left: Vec<u32>, pub right: Vec<u32> }

pub struct Circuit<F: GpuField> { h: *mut zk_circuit, _f: PhantomData<F> }
impl<F: GpuField> Drop for Circuit<F> { fn drop(&mut self) { unsafe { zk_circuit_free(self.h); } } }
impl<F: GpuField> Circuit<F> {
    /// layers[0] = output layer
    pub fn new(layers: &[Layer]) -> Result<Self, &'static str> {
        let mut h: *mut zk_circuit = std::ptr::null_mut();
        let rc = unsafe { zk_circuit_create(ctx::<F>(), &mut h) };
        if rc != 0 { return Err(err(rc)); }
        let c = Circuit { h, _f: PhantomData };
        for l in layers {
            let rc = unsafe { zk_circuit_add_layer(h, l.log_out as u64, l.log_in as u64, l.op.as_ptr(), l.left.as_ptr(), l.right.as_ptr()) };
            if rc != 0 { return Err(err(rc)); }
        }
        Ok(c)
    }
    /// -> (outputs, proof elements: per layer [round polys #1 | round polys #2 | W(u) | W(v)])
    pub fn prove(&self, input: &MultiLinearPolynomial<F>, seed: &[u8; 32]) -> Result<(MultiLinearPolynomial<F>, Vec<F>), &'static str> {
        let mut n = 0u64;
        unsafe { zk_circuit_proof_elems(self.h, &mut n); }
        let mut proof = vec![F::zero(); n as usize];
        let mut out: *mut zk_mle = std::ptr::null_mut();
        let rc = unsafe { zk_gkr_prove(self.h, input.h, seed.as_ptr(), &mut out, proof.as_mut_ptr() as *mut u64) };
        if rc != 0 { return Err(err(rc)); }
        Ok((MultiLinearPolynomial { ctx: input.ctx, h: out, _f: PhantomData }, proof))
    }
    /// Ok(true) accept / Ok(false) reject (a sumcheck round check, a layer's wiring check or the input check failed)
    pub fn verify(&self, input: &MultiLinearPolynomial<F>, outputs: &MultiLinearPolynomial<F>, seed: &[u8; 32], proof: &[F])
        -> Result<bool, &'static str> {
        match unsafe { zk_gkr_verify(self.h, input.h, outputs.h, seed.as_ptr(), proof.as_ptr() as *const u64) } {
            0 => Ok(true),
            -9 | -27 => Ok(false),
            rc => Err(err(rc)),
        }
    }
}
