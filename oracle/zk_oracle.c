/*
 * zk_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See zk_oracle.h.
 *
 * Plain C11 + unsigned __int128.  Every function cites the reference lines it restates
 * (paths relative to the reference checkout).  Field arithmetic restates ark-ff 0.5.0's
 * Fp<MontBackend<_,4>> (not vendored in the reference): 4x64-bit Montgomery, R = 2^256,
 * results always fully reduced.  All Montgomery constants are DERIVED here from the
 * modulus alone (no table shared with the product library).
 */
#include "zk_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * field parameters
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    u64 p[4];
    u64 inv;     /* -p^-1 mod 2^64 */
    u64 r1[4];   /* R mod p   (= F::one() in memory) */
    u64 r2[4];   /* R^2 mod p */
    unsigned bits;
    unsigned two_adicity;
    u64 generator; /* ark-ff GENERATOR (multiplicative generator, small integer) */
    u64 root[4];   /* TWO_ADIC_ROOT_OF_UNITY = g^((p-1)/2^s), Montgomery form */
    int ready;
} fparams;

static fparams g_fields[3] = {
    /* BN254 Fr: ark-bn254 is NOT a dependency of the reference; legal instantiation of F: PrimeField */
    {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
     0, {0}, {0}, 254, 28, 5, {0}, 0},
    /* BLS12-381 Fr: the field of the reference's polynomial/sumcheck tests (evaluation_form.rs:109) */
    {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
     0, {0}, {0}, 255, 32, 7, {0}, 0},
    /* BLS12-377 Fr: the field of the reference's fft test (fft/src/lib.rs:75) */
    {{0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL},
     0, {0}, {0}, 253, 47, 22, {0}, 0},
};

static int geq4(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static u64 add4(const u64 a[4], const u64 b[4], u64 out[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a[i] + b[i];
        out[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}
static u64 sub4(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        out[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static int is_zero4(const u64 a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static int eq4(const u64 a[4], const u64 b[4]) {
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
}

/* plain modular ops on fully reduced values */
static void f_add(const fparams *F, const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[4];
    u64 c = add4(a, b, t);
    if (c || geq4(t, F->p)) sub4(t, F->p, t);
    memcpy(out, t, 32);
}
static void f_sub(const fparams *F, const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[4];
    if (sub4(a, b, t)) add4(t, F->p, t);
    memcpy(out, t, 32);
}
/* Montgomery product a*b*R^-1 mod p (CIOS) */
static void f_mul(const fparams *F, const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 cur = (u128)a[j] * b[i] + t[j] + carry;
            t[j] = (u64)cur;
            carry = cur >> 64;
        }
        u128 cur = (u128)t[4] + carry;
        t[4] = (u64)cur;
        t[5] = (u64)(cur >> 64);
        u64 m = t[0] * F->inv;
        cur = (u128)m * F->p[0] + t[0];
        carry = cur >> 64;
        for (int j = 1; j < 4; ++j) {
            cur = (u128)m * F->p[j] + t[j] + carry;
            t[j - 1] = (u64)cur;
            carry = cur >> 64;
        }
        cur = (u128)t[4] + carry;
        t[3] = (u64)cur;
        t[4] = t[5] + (u64)(cur >> 64);
    }
    if (t[4] || geq4(t, F->p)) sub4(t, F->p, t);
    memcpy(out, t, 32);
}
static void f_pow(const fparams *F, const u64 a[4], u64 e, u64 out[4]) {
    u64 acc[4], base[4];
    memcpy(acc, F->r1, 32);
    memcpy(base, a, 32);
    while (e) {
        if (e & 1) f_mul(F, acc, base, acc);
        f_mul(F, base, base, base);
        e >>= 1;
    }
    memcpy(out, acc, 32);
}
/* a^e for a 256-bit exponent given as 4 limbs */
static void f_pow4(const fparams *F, const u64 a[4], const u64 e[4], u64 out[4]) {
    u64 acc[4];
    memcpy(acc, F->r1, 32);
    for (int i = 255; i >= 0; --i) {
        f_mul(F, acc, acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) f_mul(F, acc, a, acc);
    }
    memcpy(out, acc, 32);
}
static void f_from_canonical(const fparams *F, const u64 limbs[4], u64 out[4]) { f_mul(F, limbs, F->r2, out); }
static void f_to_canonical(const fparams *F, const u64 a[4], u64 out[4]) {
    const u64 one[4] = {1, 0, 0, 0};
    f_mul(F, a, one, out);
}

static const fparams *field_get(int field) {
    if (field < 0 || field > 2) return NULL;
    fparams *F = &g_fields[field];
    if (F->ready) return F;
    /* -p^-1 mod 2^64 by Newton iteration */
    u64 x = 1;
    for (int i = 0; i < 6; ++i) x *= 2 - F->p[0] * x;
    F->inv = (u64)0 - x;
    /* R mod p and R^2 mod p by 256 / 512 modular doublings of 1 */
    u64 v[4] = {1, 0, 0, 0};
    for (int i = 0; i < 512; ++i) {
        u64 c = add4(v, v, v);
        if (c || geq4(v, F->p)) sub4(v, F->p, v);
        if (i == 255) memcpy(F->r1, v, 32);
    }
    memcpy(F->r2, v, 32);
    /* TWO_ADIC_ROOT_OF_UNITY = GENERATOR^t, p-1 = 2^s * t */
    u64 t[4], one[4] = {1, 0, 0, 0}, g[4] = {F->generator, 0, 0, 0}, gm[4];
    sub4(F->p, one, t);
    for (unsigned i = 0; i < F->two_adicity; ++i) {
        t[0] = (t[0] >> 1) | (t[1] << 63);
        t[1] = (t[1] >> 1) | (t[2] << 63);
        t[2] = (t[2] >> 1) | (t[3] << 63);
        t[3] >>= 1;
    }
    f_from_canonical(F, g, gm);
    f_pow4(F, gm, t, F->root);
    F->ready = 1;
    return F;
}

int orc_field_modulus(int field, u64 out[4]) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    memcpy(out, F->p, 32);
    return ORC_OK;
}
/* the Montgomery constants and the two-adic root exactly as this file derived them (tests pin them to the literals the
 * arkworks / zkcrypto / halo2curves field definitions publish): -p^-1 mod 2^64, R mod p, R^2 mod p as plain integers,
 * TWO_ADIC_ROOT_OF_UNITY in Montgomery form */
int orc_field_constants(int field, u64 *inv, u64 r1[4], u64 r2[4], u64 root[4]) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (inv) *inv = F->inv;
    if (r1) memcpy(r1, F->r1, 32);
    if (r2) memcpy(r2, F->r2, 32);
    if (root) memcpy(root, F->root, 32);
    return ORC_OK;
}
int orc_field_two_adicity(int field) {
    const fparams *F = field_get(field);
    return F ? (int)F->two_adicity : ORC_ERR_BAD_FIELD;
}
void orc_add(int field, const u64 a[4], const u64 b[4], u64 out[4]) { f_add(field_get(field), a, b, out); }
void orc_sub(int field, const u64 a[4], const u64 b[4], u64 out[4]) { f_sub(field_get(field), a, b, out); }
void orc_mul(int field, const u64 a[4], const u64 b[4], u64 out[4]) { f_mul(field_get(field), a, b, out); }
void orc_pow(int field, const u64 a[4], u64 e, u64 out[4]) { f_pow(field_get(field), a, e, out); }
int orc_inverse(int field, const u64 a[4], u64 out[4]) {
    const fparams *F = field_get(field);
    if (is_zero4(a)) return 1;
    u64 e[4], two[4] = {2, 0, 0, 0};
    sub4(F->p, two, e); /* Fermat: a^(p-2) */
    f_pow4(F, a, e, out);
    return 0;
}
void orc_from_u64(int field, u64 v, u64 out[4]) {
    const fparams *F = field_get(field);
    u64 l[4] = {v, 0, 0, 0};
    f_from_canonical(F, l, out);
}
void orc_from_canonical(int field, const u64 limbs[4], u64 out[4]) { f_from_canonical(field_get(field), limbs, out); }
void orc_to_canonical(int field, const u64 a[4], u64 limbs[4]) { f_to_canonical(field_get(field), a, limbs); }
/* the same for n elements at once (test harness convenience: Python big-int lists <-> element arrays) */
void orc_from_canonical_n(int field, const u64 *limbs, u64 n, u64 *out) {
    const fparams *F = field_get(field);
    for (u64 i = 0; i < n; ++i) f_from_canonical(F, limbs + 4 * i, out + 4 * i);
}
void orc_to_canonical_n(int field, const u64 *a, u64 n, u64 *limbs) {
    const fparams *F = field_get(field);
    for (u64 i = 0; i < n; ++i) f_to_canonical(F, a + 4 * i, limbs + 4 * i);
}

/* elem.into_bigint().to_bytes_be()  (evaluation_form.rs:100, sumcheck/src/lib.rs:26, prover.rs:42) */
void orc_to_bytes_be(int field, const u64 a[4], uint8_t out[32]) {
    u64 c[4];
    f_to_canonical(field_get(field), a, c);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[31 - (i * 8 + b)] = (uint8_t)(c[i] >> (8 * b));
}
/* F::from_be_bytes_mod_order (transcript/src/lib.rs:29): int(bytes, big endian) mod p.
 * Horner over bytes: acc = acc*256 + byte, everything in the field. */
void orc_from_be_bytes_mod_order(int field, const uint8_t *bytes, size_t len, u64 out[4]) {
    const fparams *F = field_get(field);
    u64 acc[4] = {0, 0, 0, 0}, c256[4], l[4] = {256, 0, 0, 0};
    f_from_canonical(F, l, c256);
    for (size_t i = 0; i < len; ++i) {
        u64 b[4], bl[4] = {bytes[i], 0, 0, 0};
        f_from_canonical(F, bl, b);
        f_mul(F, acc, c256, acc);
        f_add(F, acc, b, acc);
    }
    memcpy(out, acc, 32);
}
/* F::get_root_of_unity(n) (fft/src/lib.rs:6,14): ark-ff 0.5.0 FftField -- n must be a power of two
 * <= 2^TWO_ADICITY; omega = TWO_ADIC_ROOT_OF_UNITY squared (s - log2 n) times.  (BN254 Fr also declares a
 * small subgroup 3^2 in ark-bn254; for power-of-two n that branch reduces to the same value.) */
int orc_root_of_unity(int field, u64 n, u64 out[4]) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (n == 0 || (n & (n - 1))) return ORC_ERR_FFT_NO_ROOT;
    unsigned lg = 0;
    while ((1ULL << lg) < n) ++lg;
    if (lg > F->two_adicity) return ORC_ERR_FFT_NO_ROOT;
    u64 w[4];
    memcpy(w, F->root, 32);
    for (unsigned i = lg; i < F->two_adicity; ++i) f_mul(F, w, w, w);
    memcpy(out, w, 32);
    return ORC_OK;
}

/* synthetic inputs -- build-owned generator, identical on host oracle and device */
static u64 splitmix64(u64 x) {
    u64 z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void orc_fill_random(int field, u64 seed, u64 first_index, u64 count, u64 *out) {
    const fparams *F = field_get(field);
    u64 topmask = (F->bits % 64) ? ((1ULL << (F->bits % 64)) - 1) : ~0ULL;
    for (u64 n = 0; n < count; ++n) {
        u64 h0 = splitmix64(seed ^ splitmix64(first_index + n));
        u64 l[4];
        for (u64 attempt = 0;; ++attempt) {
            for (int j = 0; j < 4; ++j) l[j] = splitmix64(h0 + 4 * attempt + (u64)j);
            l[3] &= topmask;
            if (!geq4(l, F->p)) break;
        }
        f_from_canonical(F, l, out + 4 * n);
    }
}

/* ------------------------------------------------------------------------------------------
 * polynomial/src/multilinear/pairing_index.rs
 * ---------------------------------------------------------------------------------------- */
static u64 mask_n(unsigned n) { return (n >= 64) ? ~0ULL : ((1ULL << n) - 1); } /* :24-26 */
u64 orc_insert_bit(u64 val, unsigned index, u64 bit) {                            /* :16-20 */
    u64 high = val >> index;
    u64 low = val & mask_n(index);
    return (high << (index + 1)) | (bit << index) | low;
}
int orc_index_pair(unsigned n_vars, unsigned index, u64 *left, u64 *right) {      /* :2-9 */
    if (n_vars == 0) return ORC_ERR_PANIC_INDEX;           /* n_vars - 1 underflows (u8) */
    unsigned base = n_vars - 1;
    if (index > base) return ORC_ERR_PANIC_INDEX;          /* base - index underflows (u8) */
    u64 pairs = 1ULL << base;
    for (u64 val = 0; val < pairs; ++val) {
        u64 l = orc_insert_bit(val, base - index, 0);
        left[val] = l;
        right[val] = l | (1ULL << (base - index));
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * polynomial/src/multilinear/evaluation_form.rs
 * ---------------------------------------------------------------------------------------- */
int orc_mle_new_check(u64 n_vars, u64 len) {                                      /* :15-27 */
    if (n_vars >= 64 || len != (1ULL << n_vars)) return ORC_ERR_EVAL_LEN;
    return ORC_OK;
}

/* :40-80.  Clone the table (:49); per assignment walk index_pair(n_vars - i, initial_var) (:55) writing pair
 * ordinal j in place (:60-70, with the is_zero / is_one shortcuts); truncate + copy (:75-79). */
int orc_mle_partial_evaluate(int field, u64 n_vars, const u64 *evals, u64 initial_var,
                             const u64 *assignments, u64 n_assign, u64 *out) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (n_assign > n_vars) return ORC_ERR_PANIC_INDEX;      /* self.n_vars - assignments.len() underflows (:75) */
    for (u64 i = 0; i < n_assign; ++i) {                    /* pre-flight the u8 underflow panics of :55 / pairing_index :3,:6 */
        u64 nv = n_vars - i;
        if (nv == 0 || initial_var > nv - 1) return ORC_ERR_PANIC_INDEX;
    }
    u64 len = 1ULL << n_vars;
    u64 *w = (u64 *)malloc(len * 32);
    if (!w) return ORC_ERR_ALLOC;
    memcpy(w, evals, len * 32); /* :49 clone */
    for (u64 i = 0; i < n_assign; ++i) {
        const u64 *a = assignments + 4 * i;
        unsigned nv = (unsigned)(n_vars - i);
        unsigned base = nv - 1, pos = base - (unsigned)initial_var;
        u64 pairs = 1ULL << base;
        int a_zero = is_zero4(a), a_one = eq4(a, F->r1);
        for (u64 j = 0; j < pairs; ++j) {
            u64 lp = orc_insert_bit(j, pos, 0), rp = lp | (1ULL << pos);
            u64 l[4], r[4];
            memcpy(l, w + 4 * lp, 32);
            memcpy(r, w + 4 * rp, 32);
            if (a_zero) {
                memcpy(w + 4 * j, l, 32);
            } else if (a_one) {
                memcpy(w + 4 * j, r, 32);
            } else {
                u64 d[4], m[4];
                f_sub(F, l, r, d);      /* left - right            */
                f_mul(F, a, d, m);      /* assignment * (..)       */
                f_sub(F, l, m, w + 4 * j); /* left - assignment*(..) */
            }
        }
    }
    memcpy(out, w, (1ULL << (n_vars - n_assign)) * 32); /* :76-79 */
    free(w);
    return ORC_OK;
}

/* optimised-CPU baseline: out[j] = lo - r*(lo - hi), j < 2^(n-1), all cores (evaluation_form.rs:68 only) */
int orc_fold_msb_parallel(int field, u64 n_vars, const u64 *evals, const u64 r[4], u64 *out, int threads) {
    const fparams *F = field_get(field);
    if (!F || n_vars == 0) return ORC_ERR_BAD_FIELD;
    const long half = (long)(1ULL << (n_vars - 1));
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#endif
    (void)threads;
#pragma omp parallel for schedule(static)
    for (long j = 0; j < half; ++j) {
        u64 d[4], m[4];
        f_sub(F, evals + 4 * j, evals + 4 * (j + half), d);
        f_mul(F, r, d, m);
        f_sub(F, evals + 4 * j, m, out + 4 * j);
    }
    return used;
}

int orc_mle_evaluate(int field, u64 n_vars, const u64 *evals, const u64 *point, u64 n_point, u64 out[4]) { /* :83-89 */
    if (n_point != n_vars) return ORC_ERR_EVAL_ARITY;
    if (n_vars == 0) {              /* partial_evaluate(0, []) returns the table; element 0 */
        memcpy(out, evals, 32);
        return ORC_OK;
    }
    return orc_mle_partial_evaluate(field, n_vars, evals, 0, point, n_point, out);
}

void orc_mle_to_bytes(int field, u64 n_vars, const u64 *evals, uint8_t *out) {      /* :97-103 */
    u64 len = 1ULL << n_vars;
    for (u64 i = 0; i < len; ++i) orc_to_bytes_be(field, evals + 4 * i, out + 32 * i);
}

/* ------------------------------------------------------------------------------------------
 * polynomial/src/multilinear/coefficient_form.rs:340-347 (to_evaluation_form)
 * ---------------------------------------------------------------------------------------- */
/* For every hypercube point in BooleanHyperCube order (boolean_hypercube.rs:27-45: binary_string(index, n), first char
 * = variable 0) push evaluate_slice(point) (coefficient_form.rs:39-68).  At a 0/1 point a term c * prod_{v in key} x_v
 * (key bit v <-> variable v, :418-430) contributes c exactly when every variable of the key is 1. */
int orc_coeff_to_evaluation(int field, u64 n_vars, const u64 *keys, const u64 *coeffs, u64 n_terms, u64 *out) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (n_vars >= 64) return ORC_ERR_EVAL_LEN;
    for (u64 t = 0; t < n_terms; ++t)
        if (keys[t] >> n_vars) return ORC_ERR_COEFF_RANGE;                           /* :183-186 */
    if (n_vars == 0) return ORC_OK;   /* the hypercube iterator yields nothing for bit_size 0 (:31): empty vector */
    u64 len = 1ULL << n_vars;
    for (u64 idx = 0; idx < len; ++idx) {
        u64 mask = 0;                  /* variables set to one at this point: char v of the binary string = bit (n-1-v) */
        for (u64 v = 0; v < n_vars; ++v)
            if ((idx >> (n_vars - 1 - v)) & 1) mask |= 1ULL << v;
        u64 acc[4] = {0, 0, 0, 0};
        for (u64 t = 0; t < n_terms; ++t)
            if ((keys[t] & mask) == keys[t]) f_add(F, acc, coeffs + 4 * t, acc);
        memcpy(out + 4 * idx, acc, 32);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * polynomial/src/product_poly.rs
 * ---------------------------------------------------------------------------------------- */
int orc_product_new_check(u64 k, const u64 *n_vars_each) {                          /* :14-32 */
    if (k == 0) return ORC_ERR_EMPTY_PRODUCT;
    for (u64 i = 1; i < k; ++i)
        if (n_vars_each[i] != n_vars_each[0]) return ORC_ERR_ARITY_MISMATCH;
    return ORC_OK;
}
void orc_prod_reduce(int field, u64 k, u64 n_vars, const u64 *const *tables, u64 *out) { /* :66-74 */
    const fparams *F = field_get(field);
    u64 len = 1ULL << n_vars;
    memcpy(out, tables[0], len * 32);                                                /* :67 to_vec */
    for (u64 f = 1; f < k; ++f)
        for (u64 i = 0; i < len; ++i) f_mul(F, out + 4 * i, tables[f] + 4 * i, out + 4 * i); /* :70 */
}
/* .iter().sum::<F>() of a slice (sumcheck/src/prover.rs:53-54; the tests' "claimed sum" of a product table, sumcheck/src/lib.rs:56) */
void orc_sum(int field, const u64 *elems, u64 n, u64 out[4]) {
    const fparams *F = field_get(field);
    u64 s[4] = {0, 0, 0, 0};
    for (u64 j = 0; j < n; ++j) f_add(F, s, elems + 4 * j, s);
    memcpy(out, s, 32);
}
int orc_product_evaluate(int field, u64 k, u64 n_vars, const u64 *const *tables, const u64 *point,
                         u64 n_point, u64 out[4]) {                                 /* :36-44 */
    const fparams *F = field_get(field);
    if (n_point != n_vars) return ORC_ERR_EVAL_ARITY;
    u64 prod[4];
    memcpy(prod, F->r1, 32);
    for (u64 f = 0; f < k; ++f) {
        u64 v[4];
        int rc = orc_mle_evaluate(field, n_vars, tables[f], point, n_point, v);
        if (rc) return rc;
        f_mul(F, prod, v, prod);
    }
    memcpy(out, prod, 32);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * sha3 0.10.8 Keccak256 (original Keccak padding 0x01, rate 136) -- NOT NIST SHA3-256
 * ---------------------------------------------------------------------------------------- */
static const u64 KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43,
                                   25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static void keccak_f1600(u64 s[25]) {
    for (int round = 0; round < 24; ++round) {
        u64 c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(s[x + 5 * y], KECCAK_ROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= KECCAK_RC[round];
    }
}
typedef struct {
    u64 s[25];
    uint8_t buf[136];
    size_t fill;
} keccak_ctx;
static void keccak_init(keccak_ctx *c) { memset(c, 0, sizeof *c); }
static void keccak_absorb_block(keccak_ctx *c, const uint8_t *blk) {
    for (int i = 0; i < 17; ++i) {
        u64 w = 0;
        for (int b = 0; b < 8; ++b) w |= (u64)blk[8 * i + b] << (8 * b);
        c->s[i] ^= w;
    }
    keccak_f1600(c->s);
}
static void keccak_update(keccak_ctx *c, const uint8_t *data, size_t len) {
    while (len) {
        size_t take = 136 - c->fill;
        if (take > len) take = len;
        memcpy(c->buf + c->fill, data, take);
        c->fill += take;
        data += take;
        len -= take;
        if (c->fill == 136) {
            keccak_absorb_block(c, c->buf);
            c->fill = 0;
        }
    }
}
static void keccak_finalize_reset(keccak_ctx *c, uint8_t out[32]) {
    memset(c->buf + c->fill, 0, 136 - c->fill);
    c->buf[c->fill] ^= 0x01;
    c->buf[135] ^= 0x80;
    keccak_absorb_block(c, c->buf);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(c->s[i] >> (8 * b));
    keccak_init(c);
}
void orc_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    keccak_ctx c;
    keccak_init(&c);
    keccak_update(&c, data, len);
    keccak_finalize_reset(&c, out);
}

/* ------------------------------------------------------------------------------------------
 * transcript/src/lib.rs
 * ---------------------------------------------------------------------------------------- */
struct orc_transcript {
    keccak_ctx h;
};
orc_transcript *orc_transcript_new(void) {                                          /* :10-14 */
    orc_transcript *t = (orc_transcript *)malloc(sizeof *t);
    if (t) keccak_init(&t->h);
    return t;
}
void orc_transcript_free(orc_transcript *t) { free(t); }
void orc_transcript_append(orc_transcript *t, const uint8_t *data, size_t len) {    /* :16-18 */
    keccak_update(&t->h, data, len);
}
void orc_transcript_sample_challenge(orc_transcript *t, uint8_t out[32]) {          /* :20-25 */
    keccak_finalize_reset(&t->h, out); /* finalize_reset */
    keccak_update(&t->h, out, 32);     /* hasher.update(result_hash) */
}
void orc_transcript_sample_field_element(orc_transcript *t, int field, u64 out[4]) { /* :27-30 */
    uint8_t h[32];
    orc_transcript_sample_challenge(t, h);
    orc_from_be_bytes_mod_order(field, h, 32, out);
}

/* ------------------------------------------------------------------------------------------
 * sumcheck/src/prover.rs
 * ---------------------------------------------------------------------------------------- */
static void append_elems(orc_transcript *t, int field, const u64 *e, u64 n) {       /* sumcheck/src/lib.rs:23-29 */
    uint8_t b[32];
    for (u64 i = 0; i < n; ++i) {
        orc_to_bytes_be(field, e + 4 * i, b);
        orc_transcript_append(t, b, 32);
    }
}

int orc_sumcheck_prove(int field, u64 k, u64 n_vars, const u64 *const *tables, unsigned D,
                       const u64 sum[4], int absorb_table, u64 *round_polys_out, u64 *challenges_out) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (k == 0) return ORC_ERR_EMPTY_PRODUCT;
    u64 len = 1ULL << n_vars;
    orc_transcript *tr = orc_transcript_new();                                       /* prover.rs:16 / :28 */
    u64 **cur = (u64 **)malloc(k * sizeof(u64 *));
    u64 **fold = (u64 **)malloc(k * sizeof(u64 *));
    u64 *prod = (u64 *)malloc((len / 2 + 1) * 32);
    if (!tr || !cur || !fold || !prod) return ORC_ERR_ALLOC;
    for (u64 f = 0; f < k; ++f) {
        cur[f] = (u64 *)malloc(len * 32);
        fold[f] = (u64 *)malloc((len / 2 + 1) * 32);
        if (!cur[f] || !fold[f]) return ORC_ERR_ALLOC;
        memcpy(cur[f], tables[f], len * 32);
    }
    if (absorb_table) {                                                              /* prover.rs:17 poly.to_bytes() */
        uint8_t *bytes = (uint8_t *)malloc(len * 32);
        if (!bytes) return ORC_ERR_ALLOC;
        for (u64 f = 0; f < k; ++f) {                                                /* product_poly.rs:77-83 */
            orc_mle_to_bytes(field, n_vars, cur[f], bytes);
            orc_transcript_append(tr, bytes, len * 32);
        }
        free(bytes);
    }
    append_elems(tr, field, sum, 1);                                                 /* prover.rs:42 */
    int rc = ORC_OK;
    for (u64 round = 0; round < n_vars && rc == ORC_OK; ++round) {                   /* prover.rs:44 */
        u64 m = n_vars - round, half = 1ULL << (m - 1);
        u64 *rp = round_polys_out + round * (D + 1) * 4;
        for (unsigned i = 0; i <= D && rc == ORC_OK; ++i) {                          /* prover.rs:49 */
            u64 a[4];
            orc_from_u64(field, i, a);                                               /* F::from(i) */
            for (u64 f = 0; f < k; ++f) {                                            /* product_poly.rs:48-63 */
                rc = orc_mle_partial_evaluate(field, m, cur[f], 0, a, 1, fold[f]);
                if (rc) break;
            }
            if (rc) break;
            orc_prod_reduce(field, k, m - 1, (const u64 *const *)fold, prod);        /* .prod_reduce() */
            u64 s[4] = {0, 0, 0, 0};
            for (u64 j = 0; j < half; ++j) f_add(F, s, prod + 4 * j, s);             /* .iter().sum::<F>() */
            memcpy(rp + 4 * i, s, 32);
        }
        if (rc) break;
        append_elems(tr, field, rp, D + 1);                                          /* prover.rs:59 */
        u64 *ch = challenges_out + 4 * round;
        orc_transcript_sample_field_element(tr, field, ch);                          /* prover.rs:62 */
        for (u64 f = 0; f < k; ++f) {                                                /* prover.rs:64 */
            rc = orc_mle_partial_evaluate(field, m, cur[f], 0, ch, 1, fold[f]);
            if (rc) break;
            memcpy(cur[f], fold[f], half * 32);
        }
    }
    for (u64 f = 0; f < k; ++f) {
        free(cur[f]);
        free(fold[f]);
    }
    free(cur);
    free(fold);
    free(prod);
    orc_transcript_free(tr);
    return rc;
}

/* "optimised CPU" row of the bench -- NOT a restatement: the same round polynomials and challenges as
 * orc_sumcheck_prove (prove_partial semantics) computed the way a tuned CPU prover would: per round ONE pass over the
 * pairs that forms all D+1 products from (lo, hi) and a second parallel pass that folds in place; OpenMP over all cores.
 * Field sums are exact, so per-thread partial sums added in any order give the reference's canonical result. */
int orc_sumcheck_prove_fused_parallel(int field, u64 k, u64 n_vars, const u64 *const *tables, unsigned D, const u64 sum[4],
                                      u64 *round_polys_out, u64 *challenges_out, int threads) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (k == 0 || k > 8 || D > 15) return ORC_ERR_EMPTY_PRODUCT;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#endif
    (void)threads;
    const u64 len = 1ULL << n_vars;
    u64 *cur[8];
    for (u64 f = 0; f < k; ++f) {
        cur[f] = (u64 *)malloc(len * 32);
        if (!cur[f]) return ORC_ERR_ALLOC;
        memcpy(cur[f], tables[f], len * 32);
    }
    orc_transcript *tr = orc_transcript_new();
    append_elems(tr, field, sum, 1);
    u64 *part = (u64 *)calloc((size_t)used * 16 * 4, 8);
    for (u64 round = 0; round < n_vars; ++round) {
        const long half = (long)(1ULL << (n_vars - round - 1));
        memset(part, 0, (size_t)used * 16 * 32);
#pragma omp parallel
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            u64 *acc = part + (size_t)tid * 16 * 4;
#pragma omp for schedule(static)
            for (long j = 0; j < half; ++j) {
                u64 prod[16][4], v[4], diff[4];
                for (u64 f = 0; f < k; ++f) {
                    const u64 *lo = cur[f] + 4 * j, *hi = cur[f] + 4 * (j + half);
                    f_sub(F, hi, lo, diff);
                    memcpy(v, lo, 32);
                    for (unsigned t = 0; t <= D; ++t) {
                        if (t == 1) memcpy(v, hi, 32);
                        else if (t > 1) f_add(F, v, diff, v);
                        if (f == 0) memcpy(prod[t], v, 32);
                        else f_mul(F, prod[t], v, prod[t]);
                    }
                }
                for (unsigned t = 0; t <= D; ++t) f_add(F, acc + 4 * t, prod[t], acc + 4 * t);
            }
        }
        u64 *rp = round_polys_out + round * (D + 1) * 4;
        for (unsigned t = 0; t <= D; ++t) {
            u64 s[4] = {0, 0, 0, 0};
            for (int w = 0; w < used; ++w) f_add(F, s, part + ((size_t)w * 16 + t) * 4, s);
            memcpy(rp + 4 * t, s, 32);
        }
        append_elems(tr, field, rp, D + 1);
        u64 *ch = challenges_out + 4 * round;
        orc_transcript_sample_field_element(tr, field, ch);
        for (u64 f = 0; f < k; ++f) {
            u64 *T = cur[f];
#pragma omp parallel for schedule(static)
            for (long j = 0; j < half; ++j) {   /* in place: index j is written after j and j+half are read by the same thread */
                u64 d[4], m[4];
                f_sub(F, T + 4 * j, T + 4 * (j + half), d);
                f_mul(F, ch, d, m);
                f_sub(F, T + 4 * j, m, T + 4 * j);
            }
        }
    }
    for (u64 f = 0; f < k; ++f) free(cur[f]);
    free(part);
    orc_transcript_free(tr);
    return used;
}

/* ------------------------------------------------------------------------------------------
 * polynomial/src/univariate_poly.rs (:29-80, :157-209) -- only what the verifier needs
 * ---------------------------------------------------------------------------------------- */
#define ORC_MAX_DEG 256
/* interpolate ys over xs = 0..n-1 (univariate_poly.rs:43-49 -> :54-80), dense coefficients low -> high */
static void uni_interpolate(const fparams *F, int field, const u64 *ys, unsigned n, u64 *coef /* n elems */) {
    memset(coef, 0, (size_t)n * 32);
    for (unsigned bi = 0; bi < n; ++bi) {
        u64 basis[ORC_MAX_DEG + 1][4];
        unsigned blen = 1;
        memcpy(basis[0], F->r1, 32);
        u64 x[4];
        orc_from_u64(field, bi, x);
        for (unsigned xi = 0; xi < n; ++xi) {
            if (xi == bi) continue;
            u64 xv[4], negx[4], zero[4] = {0, 0, 0, 0}, den[4], deninv[4];
            orc_from_u64(field, xi, xv);
            f_sub(F, zero, xv, negx);
            f_sub(F, x, xv, den);
            orc_inverse(field, den, deninv);
            /* basis *= (X - xv) * deninv */
            u64 nb[ORC_MAX_DEG + 1][4];
            memset(nb, 0, sizeof(u64) * 4 * (blen + 1));
            for (unsigned i = 0; i < blen; ++i) {
                u64 t0[4], t1[4];
                f_mul(F, negx, deninv, t0);
                f_mul(F, basis[i], t0, t0);
                f_add(F, nb[i], t0, nb[i]);
                f_mul(F, basis[i], deninv, t1);
                f_add(F, nb[i + 1], t1, nb[i + 1]);
            }
            ++blen;
            memcpy(basis, nb, sizeof(u64) * 4 * blen);
        }
        for (unsigned i = 0; i < blen; ++i) {
            u64 t[4];
            f_mul(F, basis[i], ys + 4 * bi, t);
            f_add(F, coef + 4 * i, t, coef + 4 * i);
        }
    }
}
static void uni_evaluate(const fparams *F, const u64 *coef, unsigned n, const u64 x[4], u64 out[4]) { /* :29-40 Horner */
    u64 acc[4] = {0, 0, 0, 0};
    for (int i = (int)n - 1; i >= 0; --i) {
        f_mul(F, acc, x, acc);
        f_add(F, acc, coef + 4 * i, acc);
    }
    memcpy(out, acc, 32);
}

/* ------------------------------------------------------------------------------------------
 * sumcheck/src/verifier.rs
 * ---------------------------------------------------------------------------------------- */
/* verify_internal with each round polynomial at ITS OWN length (proof.round_polys is a Vec<Vec<F>>; :55-58 interpolates
 * whatever length the round carries: 0 evaluations -> the zero polynomial, 1 -> a constant).  round_polys = the rounds'
 * evaluations back to back. */
/* verify_internal itself takes `transcript: &mut Transcript` (verifier.rs:44-48): this is that signature -- the rounds are
 * checked on a transcript the caller already holds (verify absorbs the table first, :22; a driver that chains several
 * sumchecks on ONE transcript keeps passing the same one) */
int orc_sumcheck_verify_partial_lengths_on(orc_transcript *tr, int field, u64 n_rounds, const uint32_t *lens, const u64 sum[4],
                                           const u64 *round_polys, u64 subclaim_sum[4], u64 *challenges_out) {  /* :44-78 */
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (!tr) return ORC_ERR_ALLOC;
    for (u64 r = 0; r < n_rounds; ++r)
        if (lens[r] > ORC_MAX_DEG) return ORC_ERR_ALLOC;
    append_elems(tr, field, sum, 1);                                                 /* :50 */
    u64 claimed[4], zero[4] = {0, 0, 0, 0};
    memcpy(claimed, sum, 32);
    int rc = ORC_OK;
    u64 *coef = (u64 *)malloc((size_t)(ORC_MAX_DEG + 1) * 32);
    const u64 *rp = round_polys;
    for (u64 r = 0; r < n_rounds; ++r) {
        const unsigned len = lens[r];
        append_elems(tr, field, rp, len);                                            /* :56 */
        uni_interpolate(F, field, rp, len, coef);                                    /* :58 */
        u64 p0[4], p1[4], s[4];
        uni_evaluate(F, coef, len, zero, p0);                                        /* :61 */
        uni_evaluate(F, coef, len, F->r1, p1);                                       /* :62 */
        f_add(F, p0, p1, s);
        if (!eq4(claimed, s)) {                                                      /* :64 */
            rc = ORC_ERR_VERIFY_SUM;
            break;
        }
        u64 *ch = challenges_out + 4 * r;
        orc_transcript_sample_field_element(tr, field, ch);                          /* :69 */
        uni_evaluate(F, coef, len, ch, claimed);                                     /* :70 */
        rp += (size_t)len * 4;
    }
    free(coef);
    if (rc == ORC_OK) memcpy(subclaim_sum, claimed, 32);
    return rc;
}
int orc_sumcheck_verify_partial_lengths(int field, u64 n_rounds, const uint32_t *lens, const u64 sum[4],
                                        const u64 *round_polys, const uint8_t *table_bytes, size_t table_bytes_len,
                                        u64 subclaim_sum[4], u64 *challenges_out) {  /* :38-41 / :15-23: a fresh transcript */
    if (!field_get(field)) return ORC_ERR_BAD_FIELD;
    orc_transcript *tr = orc_transcript_new();
    if (!tr) return ORC_ERR_ALLOC;
    if (table_bytes) orc_transcript_append(tr, table_bytes, table_bytes_len);        /* :22 */
    const int rc = orc_sumcheck_verify_partial_lengths_on(tr, field, n_rounds, lens, sum, round_polys, subclaim_sum, challenges_out);
    orc_transcript_free(tr);
    return rc;
}
int orc_sumcheck_verify_partial(int field, u64 n_rounds, unsigned D, const u64 sum[4],
                                const u64 *round_polys, const uint8_t *table_bytes, size_t table_bytes_len,
                                u64 subclaim_sum[4], u64 *challenges_out) {         /* every round D + 1 evaluations */
    if (D + 1 > ORC_MAX_DEG) return ORC_ERR_ALLOC;
    uint32_t *lens = (uint32_t *)malloc((size_t)(n_rounds + 1) * sizeof(uint32_t));
    if (!lens) return ORC_ERR_ALLOC;
    for (u64 r = 0; r < n_rounds; ++r) lens[r] = D + 1;
    int rc = orc_sumcheck_verify_partial_lengths(field, n_rounds, lens, sum, round_polys, table_bytes, table_bytes_len,
                                                 subclaim_sum, challenges_out);
    free(lens);
    return rc;
}

int orc_sumcheck_verify_lengths(int field, u64 k, u64 n_vars, const u64 *const *tables, u64 n_round_polys,
                                const uint32_t *lens, const u64 sum[4], const u64 *round_polys) { /* :15-33 */
    if (n_round_polys != n_vars) return ORC_ERR_VERIFY_ROUNDS;                       /* :17-19 */
    u64 len = 1ULL << n_vars;
    uint8_t *bytes = (uint8_t *)malloc(k * len * 32);
    u64 *ch = (u64 *)malloc((n_vars + 1) * 32);
    if (!bytes || !ch) return ORC_ERR_ALLOC;
    for (u64 f = 0; f < k; ++f) orc_mle_to_bytes(field, n_vars, tables[f], bytes + f * len * 32);
    u64 sub[4], ev[4];
    int rc = orc_sumcheck_verify_partial_lengths(field, n_vars, lens, sum, round_polys, bytes, k * len * 32, sub, ch);
    if (rc == ORC_OK) {
        rc = orc_product_evaluate(field, k, n_vars, tables, ch, n_vars, ev);         /* :27-29 */
        if (rc == ORC_OK) rc = eq4(ev, sub) ? 1 : 0;                                 /* :31 */
    }
    free(bytes);
    free(ch);
    return rc;
}

int orc_sumcheck_verify(int field, u64 k, u64 n_vars, const u64 *const *tables, u64 n_round_polys,
                        unsigned D, const u64 sum[4], const u64 *round_polys) {     /* :15-33 */
    if (n_round_polys != n_vars) return ORC_ERR_VERIFY_ROUNDS;                       /* :17-19 */
    u64 len = 1ULL << n_vars;
    uint8_t *bytes = (uint8_t *)malloc(k * len * 32);
    u64 *ch = (u64 *)malloc((n_vars + 1) * 32);
    if (!bytes || !ch) return ORC_ERR_ALLOC;
    for (u64 f = 0; f < k; ++f) orc_mle_to_bytes(field, n_vars, tables[f], bytes + f * len * 32);
    u64 sub[4], ev[4];
    int rc = orc_sumcheck_verify_partial(field, n_vars, D, sum, round_polys, bytes, k * len * 32, sub, ch);
    if (rc == ORC_OK) {
        rc = orc_product_evaluate(field, k, n_vars, tables, ch, n_vars, ev);         /* :27-29 */
        if (rc == ORC_OK) rc = eq4(ev, sub) ? 1 : 0;                                 /* :31 */
    }
    free(bytes);
    free(ch);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * fft/src/lib.rs
 * ---------------------------------------------------------------------------------------- */
int orc_fft_internal(int field, const u64 *in, u64 n, const u64 omega[4], u64 *out) { /* :21-46 */
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    if (n == 1) {                                                                    /* :22-24 */
        memcpy(out, in, 32);
        return ORC_OK;
    }
    if (n == 0 || (n & (n - 1))) return ORC_ERR_FFT_NOT_POW2;                        /* :28-30 */
    u64 h = n / 2;
    u64 *even = (u64 *)malloc(h * 32), *odd = (u64 *)malloc(h * 32);               /* :48-61 split_even_odd */
    u64 *ee = (u64 *)malloc(h * 32), *oe = (u64 *)malloc(h * 32);
    if (!even || !odd || !ee || !oe) return ORC_ERR_ALLOC;
    for (u64 i = 0; i < n; ++i) memcpy(((i & 1) ? odd : even) + 4 * (i / 2), in + 4 * i, 32);
    u64 w2[4];
    f_mul(F, omega, omega, w2);                                                      /* omega.square() */
    int rc = orc_fft_internal(field, even, h, w2, ee);                               /* :36 */
    if (rc == ORC_OK) rc = orc_fft_internal(field, odd, h, w2, oe);                  /* :37 */
    if (rc == ORC_OK) {
        for (u64 i = 0; i < h; ++i) {                                                /* :40-43 */
            u64 wi[4], wj[4], t[4];
            f_pow(F, omega, i, wi);
            f_pow(F, omega, i + h, wj);
            f_mul(F, wi, oe + 4 * i, t);
            f_add(F, ee + 4 * i, t, out + 4 * i);
            f_mul(F, wj, oe + 4 * i, t);
            f_add(F, ee + 4 * i, t, out + 4 * (i + h));
        }
    }
    free(even);
    free(odd);
    free(ee);
    free(oe);
    return rc;
}
int orc_fft(int field, const u64 *in, u64 n, u64 *out) {                             /* :4-8 */
    u64 w[4];
    int rc = orc_root_of_unity(field, n, w);
    if (rc) return rc;
    return orc_fft_internal(field, in, n, w, out);
}
int orc_ifft(int field, const u64 *in, u64 n, u64 *out) {                            /* :11-19 */
    const fparams *F = field_get(field);
    u64 w[4], wi[4], nm[4], ninv[4];
    int rc = orc_root_of_unity(field, n, w);
    if (rc) return rc;
    orc_inverse(field, w, wi);
    rc = orc_fft_internal(field, in, n, wi, out);
    if (rc) return rc;
    orc_from_u64(field, n, nm);
    orc_inverse(field, nm, ninv);                                                    /* F::from(n).inverse() */
    for (u64 i = 0; i < n; ++i) f_mul(F, out + 4 * i, ninv, out + 4 * i);            /* :17 */
    return ORC_OK;
}

/* Iterative DIT with a twiddle table: same DFT out[i] = sum_j in[j] * omega^(i*j), natural order both sides. */
int orc_ntt_fast(int field, const u64 *in, u64 n, int inverse, u64 *out) {
    const fparams *F = field_get(field);
    u64 w[4];
    int rc = orc_root_of_unity(field, n, w);
    if (rc) return rc;
    if (inverse) orc_inverse(field, w, w);
    unsigned lg = 0;
    while ((1ULL << lg) < n) ++lg;
    for (u64 i = 0; i < n; ++i) { /* bit-reversal copy */
        u64 r = 0;
        for (unsigned b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        memcpy(out + 4 * r, in + 4 * i, 32);
    }
    u64 *tw = (u64 *)malloc((n / 2 + 1) * 32);
    if (!tw) return ORC_ERR_ALLOC;
    memcpy(tw, F->r1, 32);
    for (u64 i = 1; i < n / 2; ++i) f_mul(F, tw + 4 * (i - 1), w, tw + 4 * i);
    for (unsigned s = 1; s <= lg; ++s) {
        u64 m = 1ULL << s, h = m / 2, step = n / m;
        for (u64 base = 0; base < n; base += m)
            for (u64 j = 0; j < h; ++j) {
                u64 t[4], u[4];
                f_mul(F, tw + 4 * (j * step), out + 4 * (base + j + h), t);
                memcpy(u, out + 4 * (base + j), 32);
                f_add(F, u, t, out + 4 * (base + j));
                f_sub(F, u, t, out + 4 * (base + j + h));
            }
    }
    free(tw);
    if (inverse) {
        u64 nm[4], ninv[4];
        orc_from_u64(field, n, nm);
        orc_inverse(field, nm, ninv);
        for (u64 i = 0; i < n; ++i) f_mul(F, out + 4 * i, ninv, out + 4 * i);
    }
    return ORC_OK;
}

/* One output of the transform by its definition (fft/src/lib.rs:39-45 computes out[k] = sum_j in[j] * omega^(j*k)):
 * checks single outputs of transforms too large for the recursion above.  OpenMP over chunks of j; exact field sums, so
 * the association does not matter. */
int orc_dft_point(int field, const u64 *in, u64 n, u64 k, int inverse, u64 out[4]) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    u64 w[4], wk[4];
    int rc = orc_root_of_unity(field, n, w);
    if (rc) return rc;
    if (inverse) orc_inverse(field, w, w);
    f_pow(F, w, k % n, wk);
    const u64 chunk = 1ULL << 12;
    const u64 n_chunks = (n + chunk - 1) / chunk;
    u64 *part = (u64 *)calloc(n_chunks, 32);
    if (!part) return ORC_ERR_ALLOC;
#pragma omp parallel for schedule(static)
    for (u64 c = 0; c < n_chunks; ++c) {
        u64 pw[4], acc[4] = {0, 0, 0, 0}, t[4];
        f_pow(F, wk, c * chunk, pw);
        const u64 end = (c + 1) * chunk < n ? (c + 1) * chunk : n;
        for (u64 j = c * chunk; j < end; ++j) {
            f_mul(F, in + 4 * j, pw, t);
            f_add(F, acc, t, acc);
            f_mul(F, pw, wk, pw);
        }
        memcpy(part + 4 * c, acc, 32);
    }
    u64 acc[4] = {0, 0, 0, 0};
    for (u64 c = 0; c < n_chunks; ++c) f_add(F, acc, part + 4 * c, acc);
    free(part);
    if (inverse) {
        u64 nm[4], ninv[4];
        orc_from_u64(field, n, nm);
        orc_inverse(field, nm, ninv);
        f_mul(F, acc, ninv, acc);
    }
    memcpy(out, acc, 32);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Checker pieces for the GKR-shaped driver.  The reference has NO gkr crate: the circuit format, the statement
 * digest and the wiring check are this repository's own definitions (oracle/gkr_ref.py is the big-int
 * model of the same protocol); what follows restates those definitions in C so that a width-2^20 proof can be checked on the
 * CPU in seconds, from the reference's primitives (field ops, MLE variable order evaluation_form.rs:40-80, Keccak-256).
 * ---------------------------------------------------------------------------------------- */
/* one layer of a fan-in-2 circuit: out[z] = op[z] ? w[left[z]] * w[right[z]] : w[left[z]] + w[right[z]] */
int orc_circuit_layer(int field, u64 n_gates, const uint8_t *op, const uint32_t *left, const uint32_t *right, const u64 *w, u64 *out) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
#pragma omp parallel for schedule(static)
    for (u64 z = 0; z < n_gates; ++z) {
        if (op[z]) f_mul(F, w + 4 * (u64)left[z], w + 4 * (u64)right[z], out + 4 * z);
        else f_add(F, w + 4 * (u64)left[z], w + 4 * (u64)right[z], out + 4 * z);
    }
    return ORC_OK;
}
/* the driver's statement digest (gkr_ref.tree_digest): Keccak-256 over 128-byte leaves (the last may be shorter; no data = one
 * empty leaf), then 4-ary nodes Keccak256(child digests concatenated) until one digest is left */
int orc_tree_digest(const uint8_t *data, size_t len, uint8_t out[32]) {
    size_t n = len ? (len + 127) / 128 : 1;
    uint8_t *cur = (uint8_t *)malloc(n * 32);
    if (!cur) return ORC_ERR_ALLOC;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const size_t off = i * 128, l = len > off ? (len - off < 128 ? len - off : 128) : 0;
        orc_keccak256(data + (l ? off : 0), l, cur + 32 * i);
    }
    while (n > 1) {
        const size_t m = (n + 3) / 4;
        uint8_t *nxt = (uint8_t *)malloc(m * 32);
        if (!nxt) {
            free(cur);
            return ORC_ERR_ALLOC;
        }
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < m; ++i) {
            const size_t c = n - 4 * i < 4 ? n - 4 * i : 4;
            orc_keccak256(cur + 128 * i, 32 * c, nxt + 32 * i);
        }
        free(cur);
        cur = nxt;
        n = m;
    }
    memcpy(out, cur, 32);
    free(cur);
    return ORC_OK;
}
/* eq(point, .) as a table, variable 0 = index MSB: out[idx] = prod_v (bit_v(idx) ? point[v] : 1 - point[v]) */
int orc_eq_table(int field, const u64 *point, u64 n_vars, u64 *out) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    memcpy(out, F->r1, 32);
    for (u64 v = 0; v < n_vars; ++v) {   /* after v variables: 2^v entries; variable v becomes the new LEAST significant bit */
        const u64 cnt = 1ULL << v;
        u64 om[4];
        f_sub(F, F->r1, point + 4 * v, om);
        for (u64 i = cnt; i-- > 0;) {
            u64 e[4];
            memcpy(e, out + 4 * i, 32);
            f_mul(F, e, om, out + 4 * (2 * i));
            f_mul(F, e, point + 4 * v, out + 4 * (2 * i + 1));
        }
    }
    return ORC_OK;
}
/* the verifier's wiring predicates of one layer (gkr_ref.gkr_verify): with E = alpha * e1 + beta * e2 (e2 may be NULL),
 * add_e = sum over add gates of E[z] eq_u[left[z]] eq_v[right[z]], mul_e likewise over the mul gates */
int orc_gkr_wiring_sums(int field, u64 n_gates, const uint8_t *op, const uint32_t *left, const uint32_t *right, const u64 *e1,
                        const u64 *e2, const u64 alpha[4], const u64 beta[4], const u64 *eq_u, const u64 *eq_v, u64 out_add[4],
                        u64 out_mul[4]) {
    const fparams *F = field_get(field);
    if (!F) return ORC_ERR_BAD_FIELD;
    const int nt = omp_get_max_threads();
    u64 *part = (u64 *)calloc((size_t)nt * 8, 8);
    if (!part) return ORC_ERR_ALLOC;
#pragma omp parallel
    {
        u64 a[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
#pragma omp for schedule(static)
        for (u64 z = 0; z < n_gates; ++z) {
            u64 E[4], t[4];
            f_mul(F, alpha, e1 + 4 * z, E);
            if (e2) {
                f_mul(F, beta, e2 + 4 * z, t);
                f_add(F, E, t, E);
            }
            f_mul(F, E, eq_u + 4 * (u64)left[z], t);
            f_mul(F, t, eq_v + 4 * (u64)right[z], t);
            if (op[z]) f_add(F, m, t, m);
            else f_add(F, a, t, a);
        }
        const int id = omp_get_thread_num();
        memcpy(part + 8 * id, a, 32);
        memcpy(part + 8 * id + 4, m, 32);
    }
    u64 a[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
    for (int i = 0; i < nt; ++i) {
        f_add(F, a, part + 8 * i, a);
        f_add(F, m, part + 8 * i + 4, m);
    }
    free(part);
    memcpy(out_add, a, 32);
    memcpy(out_mul, m, 32);
    return ORC_OK;
}
