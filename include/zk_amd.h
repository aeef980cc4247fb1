/*
 * zk_amd.h -- C ABI of libzk_amd.so: the MI355X (gfx950) implementation of the iammadab/zk sumcheck /
 * MLE-fold / NTT hot path.
 *
 * The reference has no FFI: its boundary for this path is the public Rust API of the
 * `polynomial`, `sumcheck`, `transcript` and `fft` crates.  Each entry point below names the reference item it
 * replaces (file:line relative to the reference checkout); bindings/rust/ shows the shim that maps the
 * reference's type and method names onto these calls, INTEGRATION.md shows how a maintainer wires it in.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  Every call returns int32_t: 0 = ok, negative =
 *    zk_status.  zk_strerror() returns the reference's own `Err(&'static str)` text where one exists.
 *    No exception or panic crosses the boundary: misuse the reference panics on returns ZK_ERR_PANIC_*.
 *  - Field elements cross as `const uint64_t*`: ark-ff 0.5.0 `Fp<MontBackend<_,4>>` in memory -- 4 little-endian
 *    u64 limbs, Montgomery form (R = 2^256), fully reduced.  A `Vec<F>` is passed as-is, no conversion.
 *  - Device memory lives behind opaque handles (zk_mle).  The `*_host` calls are the value-semantics
 *    convenience forms (upload, compute, download) matching the reference's Vec-in / Vec-out signatures.
 *  - A zk_ctx owns one device, one HIP stream and its scratch; it is not thread-safe, distinct contexts are
 *    independent.  All work is stream-ordered; calls that return host data synchronise that stream.
 *  - There is NO CPU fallback: without a usable gfx950 device zk_ctx_create fails with ZK_ERR_NO_DEVICE.
 */
#ifndef ZK_AMD_H
#define ZK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_AMD_ABI_VERSION 6   /* 3: + zk_comm (RCCL / host), zk_shard_prover_run, zk_ntt_sharded, sample_n, zk_ctx_trim, zk_mle_equal
                                  4: + zk_sumcheck_verify_lengths / _verify_partial_lengths (per-round degrees, verifier.rs:55-58)
                                  5: + zk_comm_info, zk_bench_evaluate_device
                                  6: + zk_sumcheck_prove_batch, zk_batch_last_stats */

typedef enum zk_field {
    ZK_FIELD_BN254_FR = 0,     /* north-star field (not a dependency of the reference: SURVEY D2) */
    ZK_FIELD_BLS12_381_FR = 1, /* field of the reference's polynomial/sumcheck tests */
    ZK_FIELD_BLS12_377_FR = 2  /* field of the reference's fft test */
} zk_field;

typedef enum zk_status {
    ZK_OK = 0,
    ZK_ERR_EVAL_LEN = -1,       /* "evaluation vec len should equal 2^n_vars"            evaluation_form.rs:20 */
    ZK_ERR_EVAL_ARITY = -2,     /* "evaluate must assign to all variables"               evaluation_form.rs:85 */
    ZK_ERR_EMPTY_PRODUCT = -3,  /* "cannot create product polynomial from empty ..."     product_poly.rs:16    */
    ZK_ERR_ARITY_MISMATCH = -4, /* "... don't share the same number of variables"        product_poly.rs:25    */
    ZK_ERR_PANIC_INDEX = -5,    /* reference panics: integer underflow in index_pair / slice (evaluation_form.rs:55,75) */
    ZK_ERR_FFT_NOT_POW2 = -6,   /* reference panics: "values must be a power of 2"       fft/src/lib.rs:29     */
    ZK_ERR_FFT_NO_ROOT = -7,    /* reference panics: get_root_of_unity(..).unwrap()      fft/src/lib.rs:6,14   */
    ZK_ERR_VERIFY_ROUNDS = -8,  /* "invalid proof: require 1 round poly for each variable in poly" verifier.rs:18 */
    ZK_ERR_VERIFY_SUM = -9,     /* "verifier check failed: claimed_sum != p(0) + p(1)"   verifier.rs:64        */
    ZK_ERR_COEFF_RANGE = -10,   /* "coefficient map represents more than specificed number of variables" coefficient_form.rs:184 */
    ZK_ERR_BAD_ARG = -20,
    ZK_ERR_BAD_FIELD = -21,
    ZK_ERR_NO_DEVICE = -22,     /* no gfx950 device / HIP runtime failure at context creation */
    ZK_ERR_HIP = -23,           /* a HIP call failed; zk_last_hip_error() has the text */
    ZK_ERR_ALLOC = -24,
    ZK_ERR_UNSUPPORTED = -25,
    ZK_ERR_CONTEXT_MISMATCH = -26,
    ZK_ERR_GKR_REJECT = -27,    /* GKR-shaped driver: a layer's wiring check or the input-layer check failed (no reference text) */
    ZK_ERR_COMM = -28           /* a collective failed (RCCL error text in zk_last_hip_error) or librccl could not be loaded */
} zk_status;

typedef struct zk_ctx zk_ctx;               /* one device + stream + scratch */
typedef struct zk_mle zk_mle;               /* device-resident table of 2^n_vars elements */
typedef struct zk_transcript zk_transcript; /* host-side Keccak-256 Fiat-Shamir sponge */

/* ---- library ---------------------------------------------------------------------------------------- */
int32_t zk_abi_version(void);
const char *zk_strerror(int32_t status);
const char *zk_last_hip_error(void);
int32_t zk_device_count(int32_t *out_count);

/* ---- context ---------------------------------------------------------------------------------------- */
int32_t zk_ctx_create(int32_t field, int32_t device, zk_ctx **out_ctx);
int32_t zk_ctx_destroy(zk_ctx *ctx);
int32_t zk_ctx_synchronize(zk_ctx *ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream; NULL = the legacy default stream) instead of the
 * context's own non-blocking stream; zk_ctx_use_own_stream switches back.  Both synchronise the stream being left. */
int32_t zk_ctx_set_stream(zk_ctx *ctx, void *hip_stream);
int32_t zk_ctx_use_own_stream(zk_ctx *ctx);
int32_t zk_ctx_field(const zk_ctx *ctx, int32_t *out_field);
/* freed tables are kept in a per-context pool (hipMalloc/hipFree of a 512 MiB block costs more than a 2^24 fold); the pool
 * is trimmed automatically when it exceeds half of the device's free memory or an allocation fails; this drops it now. */
int32_t zk_ctx_trim(zk_ctx *ctx);
/* modulus as 4 LE limbs; two-adicity s of p-1 */
int32_t zk_field_modulus(int32_t field, uint64_t out_p[4]);
int32_t zk_field_two_adicity(int32_t field, int32_t *out_s);
/* host-side element helpers for callers without ark-ff (tests, Python): F::from(u64), into_bigint, from_be_bytes_mod_order */
/* F::get_root_of_unity(2^log_n) (fft/src/lib.rs:6,14): g^((p-1)/2^log_n); None -> ZK_ERR_FFT_NO_ROOT */
int32_t zk_field_root_of_unity(int32_t field, uint64_t log_n, uint64_t out[4]);
int32_t zk_fe_from_u64(int32_t field, uint64_t v, uint64_t out[4]);
int32_t zk_fe_from_canonical(int32_t field, const uint64_t limbs[4], uint64_t out[4]);
int32_t zk_fe_to_canonical(int32_t field, const uint64_t a[4], uint64_t out_limbs[4]);
int32_t zk_fe_from_be_bytes_mod_order(int32_t field, const uint8_t *bytes, size_t len, uint64_t out[4]);

/* ---- MultiLinearPolynomial  (polynomial/src/multilinear/evaluation_form.rs) ------------------------------- */
/* ::new(n_vars, evaluations) :15-27 -- len != 2^n_vars -> ZK_ERR_EVAL_LEN.  Copies host -> device. */
int32_t zk_mle_upload(zk_ctx *ctx, uint64_t n_vars, const uint64_t *evals, uint64_t len, zk_mle **out);
/* uninitialised table (for outputs) / synthetic table (bench inputs: SURVEY 8d generator, element i of stream `seed`) */
int32_t zk_mle_alloc(zk_ctx *ctx, uint64_t n_vars, zk_mle **out);
int32_t zk_mle_fill_random(zk_ctx *ctx, zk_mle *t, uint64_t seed, uint64_t first_index);
int32_t zk_mle_clone(zk_ctx *ctx, const zk_mle *t, zk_mle **out);                    /* #[derive(Clone)] :4 */
int32_t zk_mle_free(zk_ctx *ctx, zk_mle *t);
int32_t zk_mle_n_vars(const zk_mle *t, uint64_t *out_n_vars);                         /* ::n_vars :30 */
int32_t zk_mle_download(zk_ctx *ctx, const zk_mle *t, uint64_t *out_evals);           /* ::evaluation_slice :92 */
int32_t zk_mle_device_ptr(const zk_mle *t, void **out_ptr);                           /* raw device pointer (interop) */
/* #[derive(PartialEq)] :4 -- n_vars and every evaluation equal (compared on the device; canonical representation) */
int32_t zk_mle_equal(zk_ctx *ctx, const zk_mle *a, const zk_mle *b, int32_t *out_equal);
/* ::partial_evaluate(initial_var, assignments) :40-80 -> new table of n_vars - n_assign variables */
int32_t zk_mle_partial_evaluate(zk_ctx *ctx, const zk_mle *t, uint64_t initial_var,
                                const uint64_t *assignments, uint64_t n_assign, zk_mle **out);
/* the sumcheck fold: partial_evaluate(0, [r]) written into a preallocated (n_vars-1)-variable table (no allocation) */
int32_t zk_mle_fold_into(zk_ctx *ctx, const zk_mle *t, const uint64_t r[4], zk_mle *out);
/* ::evaluate(assignments) :83-89 -- n_point != n_vars -> ZK_ERR_EVAL_ARITY */
int32_t zk_mle_evaluate(zk_ctx *ctx, const zk_mle *t, const uint64_t *point, uint64_t n_point, uint64_t out[4]);
/* ::to_bytes :97-103 -- 32-byte big-endian canonical integers, concatenated (32 << n_vars bytes) */
int32_t zk_mle_to_bytes(zk_ctx *ctx, const zk_mle *t, uint8_t *out_bytes);
/* value-semantics forms of the two calls above the reference's own tests use */
int32_t zk_mle_partial_evaluate_host(zk_ctx *ctx, uint64_t n_vars, const uint64_t *evals, uint64_t len,
                                     uint64_t initial_var, const uint64_t *assignments, uint64_t n_assign,
                                     uint64_t *out_evals /* 2^(n_vars-n_assign) elements */);

/* ---- the step before the path: CoeffMultilinearPolynomial::to_evaluation_form (coefficient_form.rs:340-347) ------------
 * Sparse coefficient form {key -> coefficient} (key bit v <-> variable v, selector_to_index :418-430; the BTreeMap of
 * coefficient_form.rs:27-30 passed as parallel arrays, duplicate keys are summed like ::new :164-171) -> the dense
 * evaluation table in hypercube order, resident on the device.  key >= 2^n_vars -> ZK_ERR_COEFF_RANGE (:183-186).
 * n_vars == 0 -> ZK_ERR_EVAL_LEN (the reference returns an empty vector there, which cannot be a table). */
int32_t zk_coeff_to_evaluation(zk_ctx *ctx, uint64_t n_vars, const uint64_t *keys, const uint64_t *coeffs, uint64_t n_terms,
                               zk_mle **out);

/* ---- ProductPoly  (polynomial/src/product_poly.rs) ---------------------------------------------------------- */
/* ::new :14-32 -- k == 0 -> ZK_ERR_EMPTY_PRODUCT, unequal arity -> ZK_ERR_ARITY_MISMATCH.  Validation only:
 * a product is passed to the calls below as an array of k table handles. */
int32_t zk_product_check(const zk_mle *const *factors, uint64_t k);
/* ::prod_reduce :66-74 -> new table, element-wise product of the k factors */
int32_t zk_prod_reduce(zk_ctx *ctx, const zk_mle *const *factors, uint64_t k, zk_mle **out);
/* ::evaluate :36-44 */
int32_t zk_product_evaluate(zk_ctx *ctx, const zk_mle *const *factors, uint64_t k, const uint64_t *point,
                            uint64_t n_point, uint64_t out[4]);
/* one prover round's polynomial in evaluation form (sumcheck/src/prover.rs:49-56):
 * out[t] = sum_x prod_f P_f(t, x), t = 0..max_var_degree -- (max_var_degree+1) elements */
int32_t zk_round_sums(zk_ctx *ctx, const zk_mle *const *factors, uint64_t k, uint32_t max_var_degree,
                      uint64_t *out_sums);

/* ---- Transcript  (transcript/src/lib.rs) -- host side ------------------------------------------------------- */
int32_t zk_transcript_new(zk_transcript **out);                                        /* ::new :10-14 */
int32_t zk_transcript_free(zk_transcript *t);
int32_t zk_transcript_append(zk_transcript *t, const uint8_t *data, size_t len);       /* ::append :16-18 */
int32_t zk_transcript_sample_field_element(zk_transcript *t, int32_t field, uint64_t out[4]); /* :27-30 */
int32_t zk_transcript_sample_n_field_elements(zk_transcript *t, int32_t field, uint64_t n, uint64_t *out); /* :32-34, n*4 u64 */
int32_t zk_transcript_sample_challenge(zk_transcript *t, uint8_t out[32]);             /* :20-25 (private in the reference) */
int32_t zk_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);

/* ---- SumcheckProver<MAX_VAR_DEGREE, F>  (sumcheck/src/prover.rs) ---------------------------------------------- */
/* ::prove :15-20 (absorb_table != 0: the whole table is serialised and absorbed first) and ::prove_partial :24-30
 * (absorb_table == 0).  The factor tables are consumed as scratch only if `consume` != 0 (the reference takes the
 * polynomial by value); otherwise they are left intact.
 * out_round_polys: n_vars * (max_var_degree+1) elements (SumcheckProof.round_polys, row-major);
 * out_challenges: n_vars elements (the Vec<F> prove_partial returns). */
int32_t zk_sumcheck_prove(zk_ctx *ctx, zk_mle *const *factors, uint64_t k, uint32_t max_var_degree,
                          const uint64_t sum[4], int32_t absorb_table, int32_t consume,
                          uint64_t *out_round_polys, uint64_t *out_challenges);
/* n_proofs INDEPENDENT ::prove_partial calls (prover.rs:24-30: one per ProductPoly; the reference's callers -- a GKR prover's layers,
 * SURVEY 8d config 4 -- simply call it n_proofs times) of ONE shape (k factors, max_var_degree) and ONE size, proved side by side:
 * each proof keeps its own transcript and is bit-identical to what zk_sumcheck_prove(absorb_table = 0) returns for the same inputs,
 * but every round of all proofs is ONE kernel launch (up to 8 proofs per launch; larger batches run in groups), so the per-round
 * latency chain is paid once per group instead of once per proof.
 * factors: n_proofs * k handles, proof after proof; sums: n_proofs * 4 u64; out_round_polys: n_proofs * n_vars * (max_var_degree+1)
 * elements; out_challenges: n_proofs * n_vars elements (proof-major).  consume as in zk_sumcheck_prove (a handle listed twice
 * anywhere in the batch is proved out of place).  Errors: those of zk_sumcheck_prove; proofs of different n_vars ->
 * ZK_ERR_ARITY_MISMATCH. */
int32_t zk_sumcheck_prove_batch(zk_ctx *ctx, uint64_t n_proofs, zk_mle *const *factors, uint64_t k, uint32_t max_var_degree,
                                const uint64_t *sums, int32_t consume, uint64_t *out_round_polys, uint64_t *out_challenges);
/* what the calling thread's last zk_sumcheck_prove_batch did: launches issued once for all proofs of a group (merged) / launches
 * issued proof by proof because the kernel has no batched form for that shape (replayed).  Diagnostics only. */
int32_t zk_batch_last_stats(uint64_t *out_merged, uint64_t *out_replayed);
/* value-semantics form: k host tables of 2^n_vars elements each */
int32_t zk_sumcheck_prove_host(zk_ctx *ctx, const uint64_t *const *tables, uint64_t k, uint64_t n_vars,
                               uint32_t max_var_degree, const uint64_t sum[4], int32_t absorb_table,
                               uint64_t *out_round_polys, uint64_t *out_challenges);

/* Stepwise form of the same loop for a table sharded across devices.  Rank g of `world` (a power of
 * two) holds the shard {idx : idx mod world == g} as an (n - log2 world)-variable table with local index idx / world,
 * so every pair (j, j + 2^(m-1)) of the first n - log2 world rounds is local.  Every rank calls these in lockstep on
 * its own context; everything is asynchronous on the context's stream (use zk_ctx_set_stream to share the stream the
 * collective runs on):
 *   round_begin : (fold at the previous challenge +) local round sums -> digit lanes: per element 8 uint64 lanes, lane i
 *                 = 32-bit digit i of the Montgomery representative, zero-extended.  The caller all-reduces (sum) the
 *                 (D+1)*8 lanes across ranks -- the ONE collective of the round; integer lane sums of <= 2^16 ranks
 *                 cannot overflow and carry exactly the field sum (RCCL has no mod-p reduction).
 *   round_finish: carry-propagate + reduce the summed lanes mod p, absorb the round polynomial, squeeze the challenge
 *                 (identical on every rank: same transcript, same bytes).
 * At any point after a round_finish (typically once the local tables are <= 2^10 elements, and at the latest after the
 * last local round): tail_ptr applies the pending challenge and exposes this rank's k shard tables as one device buffer
 * [k][2^s]; the caller all-gathers them rank-major into [world][k][2^s] and tail_rounds rebuilds the (s + log2 world)-
 * variable tables and runs ALL remaining rounds on every rank redundantly -- no further collective.
 * results downloads the proof (total_rounds = n rounds).  `sum` is the GLOBAL claimed sum (prover.rs:42). */
typedef struct zk_shard_prover zk_shard_prover;
int32_t zk_shard_prover_create(zk_ctx *ctx, zk_mle *const *factors, uint64_t k, uint32_t max_var_degree,
                               const uint64_t sum[4], uint32_t world, zk_shard_prover **out);
int32_t zk_shard_prover_destroy(zk_shard_prover *sp);
int32_t zk_shard_prover_rounds(zk_shard_prover *sp, uint64_t *out_local, uint64_t *out_total, uint64_t *out_done);
int32_t zk_shard_prover_lanes_ptr(zk_shard_prover *sp, void **out_device_ptr, uint64_t *out_n_lanes);
int32_t zk_shard_prover_round_begin(zk_shard_prover *sp);
int32_t zk_shard_prover_round_finish(zk_shard_prover *sp);
int32_t zk_shard_prover_tail_ptr(zk_shard_prover *sp, void **out_device_ptr, uint64_t *out_n_elems);
int32_t zk_shard_prover_tail_rounds(zk_shard_prover *sp, const void *gathered_device /* [world][k][2^s] elements */);
int32_t zk_shard_prover_results(zk_shard_prover *sp, uint64_t *out_round_polys, uint64_t *out_challenges);
/* ---- communicator: the exchange steps of the sharded paths, driven from inside the library --------------------------
 * One zk_comm per rank (= per process = per GPU).  Two kinds:
 *   RCCL (the product path, xGMI): zk_comm_unique_id on one rank, the 128 bytes distributed by the host's own means (the
 *     Rust side: any channel it already has; Python: torch.distributed), then zk_comm_create_rccl on every rank
 *     (ncclCommInitRank); or zk_comm_wrap_rccl around a ncclComm_t the host already owns.  librccl.so.1 is dlopen'ed on
 *     first use, so single-GPU users never load it.  Collectives are enqueued on the context's stream: no host waits.
 *   host callbacks (tests / hosts with their own transport): the library stages the buffer through pinned memory,
 *     synchronises, and calls back; the callbacks return 0 on success.  Same control flow, different transport.
 *       allreduce(user, buf, n)               : buf[i] <- sum over ranks of buf[i], n uint64 lanes (cannot overflow)
 *       allgather(user, send, n, recv)        : recv[r*n .. r*n+n) <- rank r's send
 *       alltoall (user, send, recv, n)        : recv[r*n .. ) <- rank r's send[me*n .. ), n uint64 per peer */
typedef struct zk_comm zk_comm;
typedef int32_t (*zk_host_allreduce_fn)(void *user, uint64_t *buf, uint64_t n);
typedef int32_t (*zk_host_allgather_fn)(void *user, const uint64_t *send, uint64_t n, uint64_t *recv);
typedef int32_t (*zk_host_alltoall_fn)(void *user, const uint64_t *send, uint64_t *recv, uint64_t n_per_peer);
int32_t zk_comm_unique_id(uint8_t out_id[128]);
int32_t zk_comm_create_rccl(zk_ctx *ctx, const uint8_t id[128], uint32_t world, uint32_t rank, zk_comm **out);
int32_t zk_comm_wrap_rccl(zk_ctx *ctx, void *nccl_comm, uint32_t world, uint32_t rank, zk_comm **out);
int32_t zk_comm_create_host(zk_ctx *ctx, uint32_t world, uint32_t rank, zk_host_allreduce_fn allreduce,
                            zk_host_allgather_fn allgather, zk_host_alltoall_fn alltoall, void *user, zk_comm **out);
int32_t zk_comm_destroy(zk_comm *comm);
/* what the transport itself reports: RCCL's version (ncclGetVersion; 0 for the host transport) and the communicator's own rank
 * count and rank (ncclCommCount / ncclCommUserRank).  Any out pointer may be NULL. */
int32_t zk_comm_info(zk_comm *comm, int32_t *out_rccl_version, uint32_t *out_ranks, uint32_t *out_rank);
/* The WHOLE sharded prover (prove_partial, sumcheck/src/prover.rs:24-30,44-68) on a freshly created zk_shard_prover:
 * per local round {round_begin -> all-reduce of the (D+1)*8 lanes -> round_finish} while more than `gather_below` local
 * variables remain, then tail_ptr -> all-gather -> tail_rounds.  With an RCCL comm everything is enqueued on the context's
 * stream with ZERO host synchronisations (the all-reduce is the round's one collective); fetch the proof with
 * zk_shard_prover_results.  Every rank calls it with the same arguments. */
int32_t zk_shard_prover_run(zk_shard_prover *sp, zk_comm *comm, uint32_t gather_below);
/* The same run with a breakdown by HIP events on the stream (each record stalls the stream a few microseconds: a breakdown,
 * not a timing): out_ms[0] local kernels of the exchanging rounds, [1] the per-round all-reduces, [2] the all-gather of the
 * shard tails (with the pending fold), [3] the replicated tail rounds.  Synchronises. */
int32_t zk_shard_prover_run_phases(zk_shard_prover *sp, zk_comm *comm, uint32_t gather_below, double out_ms[4]);
/* Failure rule of the two collective entry points (zk_shard_prover_run*, zk_ntt_sharded): a rank whose local step fails
 * between collectives aborts a communicator it created (ncclCommAbort, so its peers' pending collectives return an error
 * instead of hanging) and returns the error; the zk_comm is dead afterwards -- every later call on it returns ZK_ERR_COMM --
 * and must be destroyed.  A wrapped communicator (zk_comm_wrap_rccl) is only marked dead: aborting it is its owner's call. */
/* fft / ifft (fft/src/lib.rs:4-19) of an N = world * 2^m point vector held as index-mod-world shards ("strided": rank r
 * holds x[r + world*j]); ONE all-to-all.  forward: strided in -> "sliced" out (rank s holds X[k], k mod M in its slice of
 * M / world, as world rows of M / world); inverse: sliced in -> strided out, scaled 1/N.  shard is not modified;
 * out != shard, same size.  m >= log2 world.  Asynchronous with an RCCL comm. */
int32_t zk_ntt_sharded(zk_ctx *ctx, zk_comm *comm, const zk_mle *shard, int32_t inverse, zk_mle *out);

/* raw device buffers on the context (pooled) and synchronous copies: for hosts that drive the exchange themselves */
int32_t zk_ctx_device_alloc(zk_ctx *ctx, uint64_t bytes, void **out_device_ptr);
int32_t zk_ctx_device_free(zk_ctx *ctx, void *device_ptr, uint64_t bytes);
int32_t zk_ctx_memcpy_dtoh(zk_ctx *ctx, void *dst_host, const void *src_device, uint64_t bytes);
int32_t zk_ctx_memcpy_htod(zk_ctx *ctx, void *dst_device, const void *src_host, uint64_t bytes);

/* ---- sum of products + GKR-shaped driver (NO reference crate; formats are this library's) ----
 * The reference's building block for GKR is prove_partial / verify_partial (sumcheck/src/prover.rs:24-30,
 * sumcheck/src/verifier.rs:38-41; intent: polynomial/src/multilinear/evaluation_form.rs:45-48).  A GKR layer polynomial is a
 * SUM of products of MLEs, so prove_partial is offered on sum_i prod_{f in term i} T_f: factors listed flat, term after
 * term, term_k[i] factors in term i (each <= max_var_degree, <= 4 terms, <= 8 factors in all, no table listed twice).
 * Round polynomials, transcript and challenges are exactly prove_partial's (one term == zk_sumcheck_prove with
 * absorb_table = 0).  out_final (optional, k*4 u64): every factor evaluated at the challenge point, i.e. the fold after
 * the last round that the reference computes and drops (prover.rs:64). */
int32_t zk_sumcheck_prove_terms(zk_ctx *ctx, zk_mle *const *factors, const uint64_t *term_k, uint64_t n_terms,
                                uint32_t max_var_degree, const uint64_t sum[4], int32_t consume,
                                uint64_t *out_round_polys, uint64_t *out_challenges, uint64_t *out_final);
/* eq(point, .) as a table: out[idx] = prod_v (bit_v(idx) ? point[v] : 1 - point[v]), variable 0 = index MSB */
int32_t zk_eq_table(zk_ctx *ctx, const uint64_t *point, uint64_t n_vars, zk_mle **out);

/* Layered arithmetic circuit, fan-in 2.  Layers are appended from the OUTPUT layer (0) towards the inputs; layer i has
 * 2^log_out gates over the 2^log_in values of layer i+1 (log_in of layer i == log_out of layer i+1; the last layer reads the
 * input table).  Gate z = (op[z], left[z], right[z]), op 0 = add, 1 = mul; indices follow the MLE convention
 * (variable 0 = index MSB).  1 <= log_in <= 30. */
typedef struct zk_circuit zk_circuit;
int32_t zk_circuit_create(zk_ctx *ctx, zk_circuit **out);
int32_t zk_circuit_add_layer(zk_circuit *c, uint64_t log_out, uint64_t log_in, const uint8_t *op, const uint32_t *left,
                             const uint32_t *right);
int32_t zk_circuit_free(zk_circuit *c);
int32_t zk_circuit_depth(const zk_circuit *c, uint64_t *out);
int32_t zk_circuit_layer_dims(const zk_circuit *c, uint64_t layer, uint64_t *out_log_out, uint64_t *out_log_in);
int32_t zk_circuit_proof_elems(const zk_circuit *c, uint64_t *out);   /* sum over layers of 6*log_in + 2 */
/* evaluate the circuit on the device: new table of 2^log_out(0) outputs */
int32_t zk_gkr_evaluate(const zk_circuit *c, const zk_mle *input, zk_mle **out_outputs);
/* GKR prover: per layer two prove_partial-style sumchecks (Libra's phase 1 over x, phase 2 over y), chained by their
 * sub-claims, on ONE transcript (Transcript semantics of transcript/src/lib.rs) that first absorbs seed (32 bytes: the caller's
 * domain separator / session id) and the library's own digests of the circuit, the inputs and the outputs (Keccak-256 tree
 * hash) -- the statement is bound before the output point is drawn.  out_proof: zk_circuit_proof_elems
 * elements, per layer [round polys #1 (log_in*3) | round polys #2 (log_in*3) | W(u) | W(v)].  Returns the outputs table. */
int32_t zk_gkr_prove(const zk_circuit *c, const zk_mle *input, const uint8_t seed[32], zk_mle **out_outputs,
                     uint64_t *out_proof);
/* ZK_OK = accept; ZK_ERR_VERIFY_SUM = a sumcheck round check failed; ZK_ERR_GKR_REJECT = wiring / input check failed.
 * The round checks of all layers are made first (host, while the device sums the wiring predicates), the wiring checks after
 * them: a proof with defects of both kinds reports the round check. */
int32_t zk_gkr_verify(const zk_circuit *c, const zk_mle *input, const zk_mle *outputs, const uint8_t seed[32],
                      const uint64_t *proof);

/* ---- pieces of the multi-GPU four-step NTT (orchestration: zk_amd/distributed.py ShardedNtt) -------------
 * The fft crate's transform (fft/src/lib.rs:21-46) over a vector sharded by index mod W: local zk_ntt per rank, then
 * the inter-rank twiddles, one all-to-all, and W-point transforms across the ranks' rows. */
/* t[j] *= scale * base^j */
int32_t zk_mle_mul_powers(zk_ctx *ctx, zk_mle *table, const uint64_t base[4], const uint64_t scale[4]);
/* in / out read as W = 2^log_w rows of L elements: out[k][j] = sum_r w^(r*k) in[r][j], w = get_root_of_unity(W)
 * (its inverse, unscaled, when inverse != 0); log_w <= 10 */
int32_t zk_dft_across(zk_ctx *ctx, const zk_mle *in, zk_mle *out, uint64_t log_w, int32_t inverse);

/* ---- SumcheckVerifier  (sumcheck/src/verifier.rs) -- host-side protocol logic; oracle check on device ------------ */
/* ::verify_partial :38-41 -> SubClaim{sum, challenges} */
int32_t zk_sumcheck_verify_partial(int32_t field, uint64_t n_rounds, uint32_t max_var_degree,
                                   const uint64_t sum[4], const uint64_t *round_polys,
                                   uint64_t out_subclaim_sum[4], uint64_t *out_challenges);
/* ::verify :15-33 -> *out_ok = 1 for Ok(true), 0 for Ok(false); Err -> negative status */
int32_t zk_sumcheck_verify(zk_ctx *ctx, const zk_mle *const *factors, uint64_t k, uint64_t n_round_polys,
                           uint32_t max_var_degree, const uint64_t sum[4], const uint64_t *round_polys,
                           int32_t *out_ok);

/* The same two with every round polynomial at ITS OWN length: SumcheckProof.round_polys is a Vec<Vec<F>> and
 * verify_internal hands each round to UnivariatePolynomial::interpolate as it comes (verifier.rs:55-58,
 * univariate_poly.rs:43-49), so a proof whose rounds carry different numbers of evaluations is legal input.
 * evals_per_round[r] (<= 256; 0 = the zero polynomial, 1 = a constant) evaluations for round r, stored back to back in
 * round_polys (sum of evals_per_round elements). */
int32_t zk_sumcheck_verify_partial_lengths(int32_t field, uint64_t n_rounds, const uint32_t *evals_per_round,
                                           const uint64_t sum[4], const uint64_t *round_polys,
                                           uint64_t out_subclaim_sum[4], uint64_t *out_challenges);
int32_t zk_sumcheck_verify_lengths(zk_ctx *ctx, const zk_mle *const *factors, uint64_t k, uint64_t n_round_polys,
                                   const uint32_t *evals_per_round, const uint64_t sum[4], const uint64_t *round_polys,
                                   int32_t *out_ok);

/* ---- fft crate  (fft/src/lib.rs) ------------------------------------------------------------------------------ */
/* fft :4-8 / ifft :11-19 on a device vector of 2^log_n elements (zk_mle doubles as the vector handle);
 * natural order in, natural order out, omega = F::get_root_of_unity(n). */
int32_t zk_ntt(zk_ctx *ctx, const zk_mle *in, int32_t inverse, zk_mle *out);
/* value-semantics forms: fft(Vec<F>) -> Vec<F>; n == 0 or n > 2^two_adicity -> ZK_ERR_FFT_NO_ROOT,
 * n not a power of two -> ZK_ERR_FFT_NO_ROOT (get_root_of_unity returns None first, fft/src/lib.rs:6) */
int32_t zk_fft_host(zk_ctx *ctx, const uint64_t *in, uint64_t n, uint64_t *out);
int32_t zk_ifft_host(zk_ctx *ctx, const uint64_t *in, uint64_t n, uint64_t *out);
/* fft_internal(values, omega) :21-46 with a caller-chosen omega: n not a power of two -> ZK_ERR_FFT_NOT_POW2 */
int32_t zk_fft_internal_host(zk_ctx *ctx, const uint64_t *in, uint64_t n, const uint64_t omega[4], uint64_t *out);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------------------------ */
/* time `reps` launches of the MSB fold of `t` into `out` with HIP events on the context's stream; average ms/launch */
int32_t zk_bench_fold(zk_ctx *ctx, const zk_mle *t, const uint64_t r[4], zk_mle *out, int32_t reps, double *out_ms);
/* the same with one event every `group` launches: out_ms_each[i] = average launch duration inside group i,
   ceil(reps / group) values (at most 65536); group = 1 brackets every launch (an event record costs the stream ~4 us) */
int32_t zk_bench_fold_samples(zk_ctx *ctx, const zk_mle *t, const uint64_t r[4], zk_mle *out, int32_t reps, int32_t group,
                              double *out_ms_each);
/* the same for zk_ntt (all passes of one transform), average ms per transform */
int32_t zk_bench_ntt(zk_ctx *ctx, const zk_mle *in, int32_t inverse, zk_mle *out, int32_t reps, double *out_ms);
/* wall clock of `reps` zk_sumcheck_prove calls (prove_partial semantics: absorb_table = 0, the tables are left intact), each
   measured around the whole call with std::chrono -- every launch, the transcript, the download of the proof and the one host
   wait -- as SURVEY 8(d) prescribes for the prover; out_ms_each[reps].  What a compiled host sees: no binding overhead. */
int32_t zk_bench_prove_partial(zk_ctx *ctx, zk_mle *const *factors, uint64_t k, uint32_t max_var_degree, const uint64_t sum[4],
                               int32_t reps, double *out_ms_each);
/* the same for zk_mle_evaluate (the reference's own criterion bench, polynomial/benches/polynomial_evaluation.rs): per-call ms */
int32_t zk_bench_evaluate(zk_ctx *ctx, const zk_mle *t, const uint64_t *point, uint64_t n_point, int32_t reps, double *out_ms_each);
/* device time of the same call: `reps` back-to-back enqueues of evaluate's launches between two HIP events on the context's stream,
   no host wait in between; average ms per evaluate (bench.py roofline_evaluate) */
int32_t zk_bench_evaluate_device(zk_ctx *ctx, const zk_mle *t, const uint64_t *point, uint64_t n_point, int32_t reps, double *out_ms);
/* register-resident modular-multiply throughput (no memory traffic): variant 0 = fe_mul chain. Returns modmul/s */
int32_t zk_bench_modmul(zk_ctx *ctx, int32_t variant, int32_t iters, double *out_modmul_per_s);
/* plain 16-B/lane streaming copy of `bytes` bytes: achieved GB/s (calibrates the HBM ceiling on this device) */
int32_t zk_bench_copy(zk_ctx *ctx, uint64_t bytes, int32_t reps, double *out_gbps);

/* ---- environment switches (zk_amd/csrc/env.hpp) ------------------------------------------------------------------
   Read once per process; every one is a tuning / A-B / debug override and every setting produces bit-identical results.  A value
   that is not a whole decimal number inside the accepted range is ignored (the default applies) with one line on stderr.

   name                    default   accepted     meaning
   ZK_SKIP1_MIN_PAIRS      65536     1 .. 2^40    rounds with at least this many pairs leave S(1) to the tail (S(1) = claim - S(0))
   ZK_LEAD_MIN_PAIRS       65536     1 .. 2^40    ... accumulate the leading coefficient instead of S(D) (K = D shapes)
   ZK_QUAD_MAX_PAIRS       32768     0 .. 2^40    rounds up to this size run four lanes per pair index (k_round_quad); 0 = never
   ZK_PIPE_MAX_PAIRS       4096      0 .. 2^40    rounds up to this size are prepared before their challenge exists; 0 = never
   ZK_ROUND_MIN_BLOCKS     512 / 256 1 .. 2048    smallest grid of the fused round kernels (256 for the GKR shape)
   ZK_FINISH_PIPE          1         0 .. 1       0: the classic single-workgroup finisher instead of the pipelined one
   ZK_SPONGE_IN_TAIL       1         0 .. 1       0: the initial sponge state goes to the device with a launch of its own instead of as an argument of round 0's tail
   ZK_ROUND0_DOT29         1         0 .. 2       0: round 0 of the two-table degree-2 shapes on the wide accumulator; 2: force dot29
   ZK_ROUND_GLDS           1         0 .. 1       0: the big rounds on k_round0_dot29 / k_round_kd instead of the LDS-DMA kernels (k_round0_glds, k_round_fused_glds)
   ZK_ROUND_GLDS_MIN_PAIRS per kernel 64 .. 2^40  smallest round (pairs, a multiple of 64) on the LDS-DMA kernels; unset: 2^21 / 2^20 (round 0: 2 tables / 2 + term), 2^16 / 2^19 (fused: 3 tables / 2 + term)
   ZK_ROUND_GLDS_NT_MIN_PAIRS 2^20   64 .. 2^40   fused LDS-DMA rounds at least this big store their half tables nontemporal
   ZK_EVAL_FOLDS           off       flag         variable-by-variable evaluate (one fold launch per variable)
   ZK_EVAL_STREAM_MIN      21        0 .. 1000    smallest table (variables) that takes the streaming evaluate kernel
   ZK_EVAL_STREAM_LEAVE    8 / 9     7 .. 12      variables the streaming launch leaves (log2 of its grid): 8 at 21 variables, 9 above
   ZK_EVAL_WEIGHT          3         0 .. 3       bit 0 / bit 1: k_eval_low / k_eval_stream weight their outputs (no second bulk launch)
   ZK_ZETA_GLOBAL          off       flag         to_evaluation_form by global passes of three index bits (round 4's path)
   ZK_ZETA_DEVICE_SORT_MIN 4096      0 .. 2^40    to_evaluation_form: term lists at least this long are ordered on the device (0: always)
   ZK_NTT_FULL_TABLE_MAX_LOG 24      0 .. 24      largest inter-pass twiddle table (log2 entries) kept in HBM; smaller: composed per element (slower, less traffic)
   ZK_TO_BYTES_THREADS     affinity  1 .. 4       host threads copying to_bytes chunks to the caller (default: CPUs allowed, at most 4)
   ZK_PUBLISH_IN_FINISHER  1         0 .. 1       0: the proof block always goes to pinned host memory by a launch of its own (k_publish_host)
   ZK_CLAIM_IN_ROUND       1         0 .. 1       0: the tails evaluate the SKIP1 claim S_prev(r_prev) themselves instead of reading it from the round kernel's claim workgroup
   ZK_SHARD_SKIP1          1         0 .. 1       0: the sharded prover's round kernels form every sum (no S(1) / S(D) derivation behind the all-reduce)
   ZK_SHARD_FAKE_ALLREDUCE_US 0      0 .. 1000    measurement aid: every all-reduce of the sharded loop is followed by a spin of this many microseconds
                                                  on its stream (a one-rank communicator standing in for a node's latency)
   ZK_PIPE_DEBUG / ZK_HOST_DEBUG  off  flag       phase stamps of the pipelined rounds / host enqueue + wait times on stderr
   ZK_BATCH_DEBUG                 off  flag       zk_sumcheck_prove_batch: report on stderr every launch replayed proof by proof   */

#ifdef __cplusplus
}
#endif
#endif /* ZK_AMD_H */
