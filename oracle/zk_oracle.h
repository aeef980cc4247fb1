/*
 * zk_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the iammadab/zk sumcheck / MLE-fold / FFT hot path,
 * written to follow the reference's Rust sources step for step (clone-per-fold,
 * (D+2)*k folds per round, is_zero/is_one shortcuts, recursive FFT with a `pow`
 * per butterfly).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / baseline:
 * the product (zk_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED against every known-answer test the reference holds for
 * this path (tests/test_oracle_kats.py) on BLS12-381 Fr,
 * plus the public Keccak-256 vectors.  UNPINNED (no reference test fixes the
 * value, the reference cannot be built here -- no Rust toolchain): transcript
 * bytes / challenges, forward-FFT values and the root-of-unity convention, and
 * everything on BN254 Fr.  For those the oracle follows the cited source lines
 * plus the ark-ff 0.5.0 conventions restated in SURVEY.md section 8c, and is
 * cross-checked against an independent Python big-int model (oracle/pyref.py).
 *
 * Element layout everywhere: 4 x uint64 little-endian limbs, Montgomery form
 * with R = 2^256, fully reduced (< p) -- ark-ff's Fp<MontBackend<_,4>> in memory.
 */
#ifndef ZK_ORACLE_H
#define ZK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BN254_FR = 0, ORC_BLS12_381_FR = 1, ORC_BLS12_377_FR = 2 };

/* status codes: 0 ok, negative = the reference's Err(&'static str) / panic */
enum {
    ORC_OK = 0,
    ORC_ERR_EVAL_LEN = -1,       /* "evaluation vec len should equal 2^n_vars"  evaluation_form.rs:20 */
    ORC_ERR_EVAL_ARITY = -2,     /* "evaluate must assign to all variables"     evaluation_form.rs:85 */
    ORC_ERR_EMPTY_PRODUCT = -3,  /* product_poly.rs:16 */
    ORC_ERR_ARITY_MISMATCH = -4, /* product_poly.rs:25 */
    ORC_ERR_PANIC_INDEX = -5,    /* reference would panic: u8/usize underflow in index_pair / slice */
    ORC_ERR_FFT_NOT_POW2 = -6,   /* fft/src/lib.rs:28-30 panic */
    ORC_ERR_FFT_NO_ROOT = -7,    /* fft/src/lib.rs:6 unwrap on None */
    ORC_ERR_VERIFY_ROUNDS = -8,  /* verifier.rs:18 */
    ORC_ERR_VERIFY_SUM = -9,     /* verifier.rs:64 */
    ORC_ERR_COEFF_RANGE = -10,   /* coefficient_form.rs:184 */
    ORC_ERR_BAD_FIELD = -20,
    ORC_ERR_ALLOC = -21
};

/* ---- field (ark-ff 0.5.0 PrimeField semantics) ---- */
int orc_field_modulus(int field, uint64_t out[4]);
int orc_field_two_adicity(int field);
/* -p^-1 mod 2^64, R mod p, R^2 mod p (plain integers, 4 LE limbs) and TWO_ADIC_ROOT_OF_UNITY (Montgomery form) as derived at
 * first use from (p, generator, two-adicity); any pointer may be NULL */
int orc_field_constants(int field, uint64_t *inv, uint64_t r1[4], uint64_t r2[4], uint64_t root[4]);
void orc_add(int field, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_sub(int field, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_mul(int field, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_pow(int field, const uint64_t a[4], uint64_t e, uint64_t out[4]);
int orc_inverse(int field, const uint64_t a[4], uint64_t out[4]); /* 0 ok, 1 if a == 0 */
void orc_from_u64(int field, uint64_t v, uint64_t out[4]);                 /* F::from(v) */
void orc_from_canonical(int field, const uint64_t limbs[4], uint64_t out[4]); /* int -> Montgomery (int must be < p) */
void orc_to_canonical(int field, const uint64_t a[4], uint64_t limbs[4]);  /* into_bigint() */
void orc_from_canonical_n(int field, const uint64_t *limbs, uint64_t n, uint64_t *out);
void orc_to_canonical_n(int field, const uint64_t *a, uint64_t n, uint64_t *limbs);
void orc_to_bytes_be(int field, const uint64_t a[4], uint8_t out[32]);     /* into_bigint().to_bytes_be() */
void orc_from_be_bytes_mod_order(int field, const uint8_t *bytes, size_t len, uint64_t out[4]);
int orc_root_of_unity(int field, uint64_t n, uint64_t out[4]);             /* F::get_root_of_unity(n) */

/* synthetic inputs: element i = first of hash(seed,i,attempt) < p, then to Montgomery */
void orc_fill_random(int field, uint64_t seed, uint64_t first_index, uint64_t count, uint64_t *out);

/* ---- polynomial/src/multilinear/pairing_index.rs ---- */
uint64_t orc_insert_bit(uint64_t val, unsigned index, uint64_t bit);          /* :16-20 */
int orc_index_pair(unsigned n_vars, unsigned index, uint64_t *left, uint64_t *right); /* :2-9 */

/* ---- polynomial/src/multilinear/evaluation_form.rs ---- */
int orc_mle_new_check(uint64_t n_vars, uint64_t len);                         /* :15-27 */
int orc_mle_partial_evaluate(int field, uint64_t n_vars, const uint64_t *evals,
                             uint64_t initial_var, const uint64_t *assignments,
                             uint64_t n_assign, uint64_t *out /* 2^(n_vars-n_assign) elems */); /* :40-80 */
int orc_mle_evaluate(int field, uint64_t n_vars, const uint64_t *evals,
                     const uint64_t *point, uint64_t n_point, uint64_t out[4]); /* :83-89 */
void orc_mle_to_bytes(int field, uint64_t n_vars, const uint64_t *evals, uint8_t *out); /* :97-103 */

/* "optimised CPU" baseline row: the same single-variable MSB fold, out of place, no clone / copy,
 * OpenMP across `threads` cores.  Same values as orc_mle_partial_evaluate(.., 0, [r]); returns the thread count used. */
int orc_fold_msb_parallel(int field, uint64_t n_vars, const uint64_t *evals, const uint64_t r[4], uint64_t *out, int threads);

/* ---- polynomial/src/multilinear/coefficient_form.rs:340-347 + boolean_hypercube.rs:27-45 (the step before the path) ----
 * to_evaluation_form of the sparse coefficient-form polynomial {key -> coeff}, key bit v <-> variable v
 * (selector_to_index :418-430).  out: 2^n_vars elements in hypercube order (binary strings, variable 0 first = MSB). */
int orc_coeff_to_evaluation(int field, uint64_t n_vars, const uint64_t *keys, const uint64_t *coeffs, uint64_t n_terms,
                            uint64_t *out);

/* ---- polynomial/src/product_poly.rs ---- */
int orc_product_new_check(uint64_t k, const uint64_t *n_vars_each);           /* :14-32 */
void orc_prod_reduce(int field, uint64_t k, uint64_t n_vars,
                     const uint64_t *const *tables, uint64_t *out);           /* :66-74 */
void orc_sum(int field, const uint64_t *elems, uint64_t n, uint64_t out[4]);   /* iter().sum::<F>(), prover.rs:53-54 */
int orc_product_evaluate(int field, uint64_t k, uint64_t n_vars, const uint64_t *const *tables,
                         const uint64_t *point, uint64_t n_point, uint64_t out[4]); /* :36-44 */

/* ---- sha3::Keccak256 + transcript/src/lib.rs ---- */
void orc_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);
typedef struct orc_transcript orc_transcript;
orc_transcript *orc_transcript_new(void);                                     /* :10-14 */
void orc_transcript_free(orc_transcript *t);
void orc_transcript_append(orc_transcript *t, const uint8_t *data, size_t len); /* :16-18 */
void orc_transcript_sample_challenge(orc_transcript *t, uint8_t out[32]);     /* :20-25 */
void orc_transcript_sample_field_element(orc_transcript *t, int field, uint64_t out[4]); /* :27-30 */

/* ---- sumcheck/src/prover.rs, sumcheck/src/verifier.rs ---- */
/* reference-faithful prover: per round (D+1) x [fold each factor -> prod_reduce -> sum], then fold at the
 * challenge.  absorb_table != 0 => `prove` (prover.rs:15), else `prove_partial` (prover.rs:24).
 * round_polys_out: n_vars*(D+1) elements; challenges_out: n_vars elements. */
int orc_sumcheck_prove(int field, uint64_t k, uint64_t n_vars, const uint64_t *const *tables,
                       unsigned max_var_degree, const uint64_t sum[4], int absorb_table,
                       uint64_t *round_polys_out, uint64_t *challenges_out);
/* bench "optimised CPU" row (not a restatement): same outputs as orc_sumcheck_prove with absorb_table = 0, fused rounds,
 * OpenMP.  Returns the number of threads used (> 0) or a negative error. */
int orc_sumcheck_prove_fused_parallel(int field, uint64_t k, uint64_t n_vars, const uint64_t *const *tables,
                                      unsigned max_var_degree, const uint64_t sum[4], uint64_t *round_polys_out,
                                      uint64_t *challenges_out, int threads);
/* verify_partial (verifier.rs:38-41 -> :44-78): returns ORC_OK and the subclaim, or the Err code. */
int orc_sumcheck_verify_partial(int field, uint64_t n_rounds, unsigned max_var_degree,
                                const uint64_t sum[4], const uint64_t *round_polys,
                                const uint8_t *table_bytes, size_t table_bytes_len, /* NULL,0 for verify_partial */
                                uint64_t subclaim_sum[4], uint64_t *challenges_out);
/* the same with every round polynomial at its own length (proof.round_polys: Vec<Vec<F>>, verifier.rs:55-58): lens[r]
 * evaluations for round r, stored back to back in round_polys */
int orc_sumcheck_verify_partial_lengths(int field, uint64_t n_rounds, const uint32_t *lens, const uint64_t sum[4],
                                        const uint64_t *round_polys, const uint8_t *table_bytes, size_t table_bytes_len,
                                        uint64_t subclaim_sum[4], uint64_t *challenges_out);
/* verify_internal's own signature (verifier.rs:44-48 takes `transcript: &mut Transcript`): the same rounds on a transcript the
 * caller holds and keeps */
int orc_sumcheck_verify_partial_lengths_on(orc_transcript *tr, int field, uint64_t n_rounds, const uint32_t *lens,
                                           const uint64_t sum[4], const uint64_t *round_polys, uint64_t subclaim_sum[4],
                                           uint64_t *challenges_out);
int orc_sumcheck_verify_lengths(int field, uint64_t k, uint64_t n_vars, const uint64_t *const *tables,
                                uint64_t n_round_polys, const uint32_t *lens, const uint64_t sum[4],
                                const uint64_t *round_polys);
/* verify (verifier.rs:15-33): 1 = Ok(true), 0 = Ok(false), negative = Err */
int orc_sumcheck_verify(int field, uint64_t k, uint64_t n_vars, const uint64_t *const *tables,
                        uint64_t n_round_polys, unsigned max_var_degree, const uint64_t sum[4],
                        const uint64_t *round_polys);

/* ---- fft/src/lib.rs ---- */
int orc_fft(int field, const uint64_t *in, uint64_t n, uint64_t *out);        /* :4-8, recursive, pow per butterfly */
int orc_ifft(int field, const uint64_t *in, uint64_t n, uint64_t *out);       /* :11-19 */
int orc_fft_internal(int field, const uint64_t *in, uint64_t n, const uint64_t omega[4], uint64_t *out); /* :21-46 */
/* same DFT (same omega) by an iterative table-driven algorithm: used only to check larger sizes in
 * reasonable time; validated against orc_fft at small n by tests/test_oracle_kats.py */
int orc_ntt_fast(int field, const uint64_t *in, uint64_t n, int inverse, uint64_t *out);
/* ONE output of fft (inverse = 0) / ifft (inverse != 0) straight from the definition out[k] = sum_j in[j] * omega^(j*k)
 * (fft/src/lib.rs:39-45): pins single outputs of transforms too large for the recursion */
int orc_dft_point(int field, const uint64_t *in, uint64_t n, uint64_t k, int inverse, uint64_t out[4]);

/* ---- checker pieces for the GKR-shaped driver (NO reference crate; definitions: DESIGN.md section 10 and
 * oracle/gkr_ref.py).  C forms of the model's circuit evaluation, statement digest, eq table and wiring predicates, so that a
 * width-2^20 proof can be checked on the CPU in seconds with nothing from the library under test. ---- */
int orc_circuit_layer(int field, uint64_t n_gates, const uint8_t *op, const uint32_t *left, const uint32_t *right,
                      const uint64_t *w, uint64_t *out);
int orc_tree_digest(const uint8_t *data, size_t len, uint8_t out[32]);
int orc_eq_table(int field, const uint64_t *point, uint64_t n_vars, uint64_t *out);
int orc_gkr_wiring_sums(int field, uint64_t n_gates, const uint8_t *op, const uint32_t *left, const uint32_t *right,
                        const uint64_t *e1, const uint64_t *e2, const uint64_t alpha[4], const uint64_t beta[4],
                        const uint64_t *eq_u, const uint64_t *eq_v, uint64_t out_add[4], uint64_t out_mul[4]);

#ifdef __cplusplus
}
#endif
#endif
