// field.cuh -- 256-bit prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Element = ark-ff 0.5.0 Fp<MontBackend<_,4>> in memory: 4 x u64 LE limbs = 8 x u32 LE limbs, Montgomery form
// with R = 2^256, always fully reduced (< p).  This is the representation the reference's hot path computes on
// (polynomial/src/multilinear/evaluation_form.rs:57-70 `left - r*(left-right)`; product_poly.rs:70 `*=`;
// sumcheck/src/prover.rs:53-54 `.sum::<F>()`), so results are bit-identical limb for limb.
//
// The modulus is a RUNTIME parameter (FieldParams, passed by value -> SGPRs): one code object serves BN254 Fr,
// BLS12-381 Fr and BLS12-377 Fr (the reference is generic over F: PrimeField).
//
// Multiplication is product-scanning (Comba) over 32-bit limbs on v_mad_u64_u32: each limb product is ONE
// quarter-rate multiply-add into a 64-bit column accumulator plus one full-rate v_addc for the column's third
// word.  The 512-bit product (mul_wide) and the Montgomery reduction (redc) are separate so the sumcheck round
// kernel can add many unreduced products and reduce once.  No MFMA: these are integer modular ops.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#else
#define ZK_HD inline
#define ZK_D inline
#endif

namespace zk {

struct FieldParams {
    uint32_t p[8];    // modulus
    uint32_t r1[8];   // R mod p      (Montgomery one)
    uint32_t r2[8];   // R^2 mod p    (to-Montgomery multiplier)
    uint32_t inv;     // -p^-1 mod 2^32
    uint32_t bits;    // bit length of p
    uint32_t p29[9];  // the modulus in nine 29-bit limbs (unsaturated multiplier, see fe_mul29)
    uint32_t inv29;   // -p^-1 mod 2^29
    uint32_t r2_29[9];   // prepared multiplier of R^2       (fe_mul29(x, .) = x*R: canonical -> Montgomery)
    uint32_t r2s_29[9];  // prepared multiplier of R^2 * 2^5 (fe_mul29(x, .) = x*R*2^5: canonical -> prepared form)
};

struct Fe {
    uint32_t v[8];
};

// ---- 96-bit column accumulator: (lo,hi) 64-bit + ex ------------------------------------------------------
struct Acc {
    uint64_t lh;
    uint32_t ex;
};

// acc += a*b
ZK_HD void mac(Acc &acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    // one quarter-rate mad whose carry-out feeds the third word directly
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc.lh), "+v"(acc.ex)
        : "v"(a), "v"(b)
        : "vcc");
#else
    uint64_t s = acc.lh + (uint64_t)a * b;
    acc.ex += (s < acc.lh) ? 1u : 0u;
    acc.lh = s;
#endif
}
// acc += sum_{i<N} a[i] * b[-i]  (b points at the FIRST product's second operand and is walked downwards: one column of a
// product-scanning multiplication).  ONE asm statement for the whole column: between separate asm statements the compiler
// pads with s_nop (it cannot see that the next statement does not read the vcc this one wrote) -- 64 of them per 256 x 256-bit
// product, 30 % of the instructions of the sums-only round kernel's loop.
template <int N>
ZK_HD void mac_col(Acc &acc, const uint32_t *a, const uint32_t *b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
#define ZK_MAC_PAIR(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
    if constexpr (N == 1) {
        asm(ZK_MAC_PAIR("%2", "%3")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(b[0])
            : "vcc");
    } else if constexpr (N == 2) {
        asm(ZK_MAC_PAIR("%2", "%4") ZK_MAC_PAIR("%3", "%5")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[-1])
            : "vcc");
    } else if constexpr (N == 3) {
        asm(ZK_MAC_PAIR("%2", "%5") ZK_MAC_PAIR("%3", "%6") ZK_MAC_PAIR("%4", "%7")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[-1]), "v"(b[-2])
            : "vcc");
    } else if constexpr (N == 4) {
        asm(ZK_MAC_PAIR("%2", "%6") ZK_MAC_PAIR("%3", "%7") ZK_MAC_PAIR("%4", "%8") ZK_MAC_PAIR("%5", "%9")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[-1]), "v"(b[-2]), "v"(b[-3])
            : "vcc");
    } else if constexpr (N == 5) {
        asm(ZK_MAC_PAIR("%2", "%7") ZK_MAC_PAIR("%3", "%8") ZK_MAC_PAIR("%4", "%9") ZK_MAC_PAIR("%5", "%10") ZK_MAC_PAIR("%6", "%11")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[-1]), "v"(b[-2]), "v"(b[-3]), "v"(b[-4])
            : "vcc");
    } else if constexpr (N == 6) {
        asm(ZK_MAC_PAIR("%2", "%8") ZK_MAC_PAIR("%3", "%9") ZK_MAC_PAIR("%4", "%10") ZK_MAC_PAIR("%5", "%11") ZK_MAC_PAIR("%6", "%12") ZK_MAC_PAIR("%7", "%13")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[-1]), "v"(b[-2]), "v"(b[-3]), "v"(b[-4]), "v"(b[-5])
            : "vcc");
    } else if constexpr (N == 7) {
        asm(ZK_MAC_PAIR("%2", "%9") ZK_MAC_PAIR("%3", "%10") ZK_MAC_PAIR("%4", "%11") ZK_MAC_PAIR("%5", "%12") ZK_MAC_PAIR("%6", "%13") ZK_MAC_PAIR("%7", "%14") ZK_MAC_PAIR("%8", "%15")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(b[0]), "v"(b[-1]), "v"(b[-2]), "v"(b[-3]), "v"(b[-4]), "v"(b[-5]), "v"(b[-6])
            : "vcc");
    } else {
        static_assert(N == 8, "a column of an 8 x 8 limb product has at most 8 terms");
        asm(ZK_MAC_PAIR("%2", "%10") ZK_MAC_PAIR("%3", "%11") ZK_MAC_PAIR("%4", "%12") ZK_MAC_PAIR("%5", "%13") ZK_MAC_PAIR("%6", "%14") ZK_MAC_PAIR("%7", "%15") ZK_MAC_PAIR("%8", "%16") ZK_MAC_PAIR("%9", "%17")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[-1]), "v"(b[-2]), "v"(b[-3]), "v"(b[-4]), "v"(b[-5]), "v"(b[-6]), "v"(b[-7])
            : "vcc");
    }
#undef ZK_MAC_PAIR
#else
    for (int i = 0; i < N; ++i) mac(acc, a[i], b[-i]);
#endif
}
// the same with the b operands wave-uniform (modulus limbs): they stay in SGPRs (one scalar source per instruction)
template <int N>
ZK_HD void mac_col_s(Acc &acc, const uint32_t *a, const uint32_t *b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
#define ZK_MAC_PAIR(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
    if constexpr (N == 1) {
        asm(ZK_MAC_PAIR("%2", "%3")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "s"(b[0])
            : "vcc");
    } else if constexpr (N == 2) {
        asm(ZK_MAC_PAIR("%2", "%4") ZK_MAC_PAIR("%3", "%5")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "s"(b[0]), "s"(b[-1])
            : "vcc");
    } else if constexpr (N == 3) {
        asm(ZK_MAC_PAIR("%2", "%5") ZK_MAC_PAIR("%3", "%6") ZK_MAC_PAIR("%4", "%7")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "s"(b[0]), "s"(b[-1]), "s"(b[-2])
            : "vcc");
    } else if constexpr (N == 4) {
        asm(ZK_MAC_PAIR("%2", "%6") ZK_MAC_PAIR("%3", "%7") ZK_MAC_PAIR("%4", "%8") ZK_MAC_PAIR("%5", "%9")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(b[0]), "s"(b[-1]), "s"(b[-2]), "s"(b[-3])
            : "vcc");
    } else if constexpr (N == 5) {
        asm(ZK_MAC_PAIR("%2", "%7") ZK_MAC_PAIR("%3", "%8") ZK_MAC_PAIR("%4", "%9") ZK_MAC_PAIR("%5", "%10") ZK_MAC_PAIR("%6", "%11")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "s"(b[0]), "s"(b[-1]), "s"(b[-2]), "s"(b[-3]), "s"(b[-4])
            : "vcc");
    } else if constexpr (N == 6) {
        asm(ZK_MAC_PAIR("%2", "%8") ZK_MAC_PAIR("%3", "%9") ZK_MAC_PAIR("%4", "%10") ZK_MAC_PAIR("%5", "%11") ZK_MAC_PAIR("%6", "%12") ZK_MAC_PAIR("%7", "%13")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "s"(b[0]), "s"(b[-1]), "s"(b[-2]), "s"(b[-3]), "s"(b[-4]), "s"(b[-5])
            : "vcc");
    } else if constexpr (N == 7) {
        asm(ZK_MAC_PAIR("%2", "%9") ZK_MAC_PAIR("%3", "%10") ZK_MAC_PAIR("%4", "%11") ZK_MAC_PAIR("%5", "%12") ZK_MAC_PAIR("%6", "%13") ZK_MAC_PAIR("%7", "%14") ZK_MAC_PAIR("%8", "%15")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "s"(b[0]), "s"(b[-1]), "s"(b[-2]), "s"(b[-3]), "s"(b[-4]), "s"(b[-5]), "s"(b[-6])
            : "vcc");
    } else {
        static_assert(N == 8, "a column of an 8 x 8 limb product has at most 8 terms");
        asm(ZK_MAC_PAIR("%2", "%10") ZK_MAC_PAIR("%3", "%11") ZK_MAC_PAIR("%4", "%12") ZK_MAC_PAIR("%5", "%13") ZK_MAC_PAIR("%6", "%14") ZK_MAC_PAIR("%7", "%15") ZK_MAC_PAIR("%8", "%16") ZK_MAC_PAIR("%9", "%17")
            : "+v"(acc.lh), "+v"(acc.ex)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "s"(b[0]), "s"(b[-1]), "s"(b[-2]), "s"(b[-3]), "s"(b[-4]), "s"(b[-5]), "s"(b[-6]), "s"(b[-7])
            : "vcc");
    }
#undef ZK_MAC_PAIR
#else
    for (int i = 0; i < N; ++i) mac(acc, a[i], b[-i]);
#endif
}
// acc += a*b with b wave-uniform (modulus limbs, the fold challenge): b stays in an SGPR
ZK_HD void mac_s(Acc &acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc.lh), "+v"(acc.ex)
        : "v"(a), "s"(b)
        : "vcc");
#else
    mac(acc, a, b);
#endif
}
// acc += x (32-bit)
ZK_HD void acc_add32(Acc &acc, uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    // x * 1 through the same mad + addc pair as a product: 12 cycles where the add_co / addc / addc chain the compiler emits is
    // 12 cycles plus two s_nop 1 (gfx950 pads every VALU-written carry before the VALU that reads it)
    asm("v_mad_u64_u32 %0, vcc, %2, 1, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc.lh), "+v"(acc.ex)
        : "v"(x)
        : "vcc");
    return;
#endif
    uint32_t c0, c1;
    uint32_t lo = __builtin_addc((uint32_t)acc.lh, x, 0u, &c0);
    uint32_t hi = __builtin_addc((uint32_t)(acc.lh >> 32), 0u, c0, &c1);
    acc.lh = ((uint64_t)hi << 32) | lo;
    acc.ex += c1;
}
// take the low word, shift the accumulator right by 32
ZK_HD uint32_t acc_shift(Acc &acc) {
    uint32_t lo = (uint32_t)acc.lh;
    acc.lh = (acc.lh >> 32) | ((uint64_t)acc.ex << 32);
    acc.ex = 0;
    return lo;
}

// ---- add / sub -------------------------------------------------------------------------------------------
// r = a + b (8 limbs), returns carry.  __builtin_addc/__builtin_subc lower to v_add_co/v_addc_co chains.
ZK_HD uint32_t add8(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t co;
        r[i] = __builtin_addc(a[i], b[i], c, &co);
        c = co;
    }
    return c;
}
// r = a - b (8 limbs), returns borrow (0/1)
ZK_HD uint32_t sub8(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t co;
        r[i] = __builtin_subc(a[i], b[i], c, &co);
        c = co;
    }
    return c;
}

ZK_HD Fe fe_add(const Fe &a, const Fe &b, const FieldParams &P) {
    Fe s, d, r;
    uint32_t carry = add8(s.v, a.v, b.v);
    uint32_t borrow = sub8(d.v, s.v, P.p);
    bool use_d = carry | (borrow ^ 1u);   // s >= p  (p < 2^255, so carry is only possible in theory)
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = use_d ? d.v[i] : s.v[i];
    return r;
}
ZK_HD Fe fe_sub(const Fe &a, const Fe &b, const FieldParams &P) {
    Fe d, e, r;
    uint32_t borrow = sub8(d.v, a.v, b.v);
    add8(e.v, d.v, P.p);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = borrow ? e.v[i] : d.v[i];
    return r;
}
ZK_HD Fe fe_neg(const Fe &a, const FieldParams &P) {
    Fe z = {{0, 0, 0, 0, 0, 0, 0, 0}};
    return fe_sub(z, a, P);
}
ZK_HD bool fe_eq(const Fe &a, const Fe &b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
    return d == 0;
}
ZK_HD bool fe_is_zero(const Fe &a) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.v[i];
    return d == 0;
}
ZK_HD Fe fe_zero() {
    Fe z = {{0, 0, 0, 0, 0, 0, 0, 0}};
    return z;
}
ZK_HD Fe fe_one(const FieldParams &P) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = P.r1[i];
    return r;
}

// ---- multiplication -------------------------------------------------------------------------------------
// t[0..16) = a * b   (full 512-bit product, product scanning)
template <int C>
ZK_HD void mul_wide_col(uint32_t t[16], Acc &acc, const uint32_t a[8], const uint32_t b[8]) {
    constexpr int lo = C < 8 ? 0 : C - 7, hi = C < 8 ? C : 7;
    mac_col<hi - lo + 1>(acc, a + lo, b + (C - lo));
    t[C] = acc_shift(acc);
    if constexpr (C + 1 < 15) mul_wide_col<C + 1>(t, acc, a, b);
}
ZK_HD void mul_wide(uint32_t t[16], const uint32_t a[8], const uint32_t b[8]) {
    Acc acc = {0, 0};
    mul_wide_col<0>(t, acc, a, b);
    t[15] = (uint32_t)acc.lh;
}

// the 16 columns of a word-serial Montgomery reduction: columns 0..7 fix the multipliers m[c], columns 8..15 give the result
template <int C>
ZK_HD void redc_cols(const uint32_t *t, uint32_t (&m)[8], uint32_t *out8, Acc &acc, const FieldParams &P) {
    acc_add32(acc, t[C]);
    if constexpr (C < 8) {
        if constexpr (C > 0) mac_col_s<C>(acc, m, P.p + C);   // sum_{i<C} m[i] * p[C-i]
        m[C] = (uint32_t)acc.lh * P.inv;
        mac_s(acc, m[C], P.p[0]);
        (void)acc_shift(acc);   // low word is zero by construction
    } else {
        if constexpr (C < 15) mac_col_s<15 - C>(acc, m + (C - 7), P.p + 7);   // sum_{i=C-7..7} m[i] * p[C-i]
        out8[C - 8] = acc_shift(acc);
    }
    if constexpr (C + 1 < 16) redc_cols<C + 1>(t, m, out8, acc, P);
}
// Montgomery reduction of a 512-bit value t < p*R (+ slack, see `extra`): returns t * R^-1 mod p, fully reduced.
// `top` is an optional 17th limb of weight 2^512 already folded by the caller (see redc_wide).
ZK_HD Fe redc(const uint32_t t[16], const FieldParams &P) {
    uint32_t m[8];
    Acc acc = {0, 0};
    Fe s;
    redc_cols<0>(t, m, s.v, acc, P);
    // result = s + carry*2^256 < 2p when t < p*R
    uint32_t carry = (uint32_t)acc.lh;
    Fe d, r;
    uint32_t borrow = sub8(d.v, s.v, P.p);
    bool use_d = carry | (borrow ^ 1u);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = use_d ? d.v[i] : s.v[i];
    return r;
}

ZK_HD Fe fe_mul(const Fe &a, const Fe &b, const FieldParams &P) {
    uint32_t t[16];
    mul_wide(t, a.v, b.v);
    return redc(t, P);
}
ZK_HD Fe fe_sqr(const Fe &a, const FieldParams &P) { return fe_mul(a, a, P); }

// ---- conversions ----------------------------------------------------------------------------------------
ZK_HD Fe fe_from_canonical(const Fe &x, const FieldParams &P) {   // x < p as plain integer -> Montgomery
    Fe r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.v[i] = P.r2[i];
    return fe_mul(x, r2, P);
}
ZK_HD Fe fe_to_canonical(const Fe &a, const FieldParams &P) {     // Montgomery -> plain integer (into_bigint)
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        t[i] = a.v[i];
        t[i + 8] = 0;
    }
    return redc(t, P);
}
ZK_HD Fe fe_from_u32(uint32_t x, const FieldParams &P) {          // F::from(x)
    Fe v = {{x, 0, 0, 0, 0, 0, 0, 0}};
    return fe_from_canonical(v, P);
}

// s (9 limbs) -= (p << k) while s >= (p << k), for k = KMAX..0: brings any s < 2^(KMAX+1) * p below p.
template <int KMAX>
ZK_HD void ladder9(uint32_t s[9], const FieldParams &P) {
#pragma unroll
    for (int k = KMAX; k >= 0; --k) {
        uint32_t d[9];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            uint32_t pk;
            if (k == 0) pk = (i < 8) ? P.p[i] : 0u;
            else pk = ((i < 8) ? (P.p[i] << k) : 0u) | ((i > 0) ? (P.p[i - 1] >> (32 - k)) : 0u);
            uint32_t bo;
            d[i] = __builtin_subc(s[i], pk, borrow, &bo);
            borrow = bo;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) s[i] = borrow ? s[i] : d[i];
    }
}

// ---- wide accumulation for the sumcheck round sums -------------------------------------------------------
// Sum of up to ~2^32 512-bit products in 17 limbs; reduced once with redc_wide.
struct WideAcc {
    uint32_t v[17];
};
ZK_HD void wide_zero(WideAcc &w) {
#pragma unroll
    for (int i = 0; i < 17; ++i) w.v[i] = 0;
}
// w += a*b  (unreduced; product scanning straight into the running sum)
template <int C>
ZK_HD void wide_mac_col(WideAcc &w, Acc &acc, const uint32_t a[8], const uint32_t b[8]) {
    constexpr int lo = C < 8 ? 0 : C - 7, hi = C < 8 ? C : 7;
    acc_add32(acc, w.v[C]);
    mac_col<hi - lo + 1>(acc, a + lo, b + (C - lo));
    w.v[C] = acc_shift(acc);
    if constexpr (C + 1 < 15) wide_mac_col<C + 1>(w, acc, a, b);
}
ZK_HD void wide_mac(WideAcc &w, const uint32_t a[8], const uint32_t b[8]) {
    Acc acc = {0, 0};
    wide_mac_col<0>(w, acc, a, b);
    acc_add32(acc, w.v[15]);
    w.v[15] = acc_shift(acc);
    w.v[16] += (uint32_t)acc.lh;
}
constexpr int kMaxLazy = 32;   // products accumulated unreduced between two Montgomery reductions (the contract of redc_wide, below)
// w * R^-1 mod p, fully reduced, for any w < 2^544 whose top limb is small: the sum of at most kMaxLazy (= 32) products of
// values < p, so top = w.v[16] <= floor(32 p^2 / 2^512) (1, 6 and 0 for the three shipped fields).  redc of the low 512 bits
// gives s < R + p; the top limb contributes top * (R mod p) < top * p.  The 9-limb sum is therefore < R + p + top*p, and
// ladder9<4> (conditional subtraction of 16p, 8p, 4p, 2p, p) brings anything below 32p under p.  So the contract is
//     R/p + 1 + top_max < 32        (R/p = 5.3, 2.2, 13.7 for BN254 / BLS12-381 / BLS12-377 Fr)
// which host_field.hpp's field_info() checks for every field it registers (a field that breaks it is refused, not mis-reduced).
ZK_HD Fe redc_wide(const WideAcc &w, const FieldParams &P) {
    uint32_t m[8];
    Acc acc = {0, 0};
    uint32_t s[9];
    redc_cols<0>(w.v, m, s, acc, P);
    s[8] = (uint32_t)acc.lh;
    // s += top * (R mod p)
    {
        uint64_t c = 0;
        const uint32_t top = w.v[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c += (uint64_t)top * P.r1[i] + s[i];
            s[i] = (uint32_t)c;
            c >>= 32;
        }
        s[8] += (uint32_t)c;
    }
    ladder9<4>(s, P);
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = s[i];
    return r;
}

// x mod p for any 256-bit x (8 LE limbs), as a canonical (non-Montgomery) integer.  p > 2^251 for every supported
// field, so x < 32p and the ladder 16p..p suffices.  Used for F::from_be_bytes_mod_order (transcript/src/lib.rs:29).
ZK_HD Fe fe_reduce_u256(const uint32_t x[8], const FieldParams &P) {
    uint32_t s[9];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = x[i];
    s[8] = 0;
    ladder9<4>(s, P);
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = s[i];
    return r;
}

// ---- multiplication by a prepared operand on unsaturated 29-bit limbs ---------------------------------------------------
// On gfx950 a carry add (v_addc_co_u32) costs as much as a multiply (v_mad_u64_u32): both ~4 cycles per wave
// (tools/mb/mb_alu.hip).  With nine 29-bit limbs every column sum of the fused multiply + Montgomery reduction
// (<= 18 products < 2^58 plus a carry) fits a 64-bit accumulator, so the whole product needs NO carry instructions:
// 162 multiply-adds, 9 low multiplies, 17 shifts and the limb split / merge -- about 260 instructions against ~420 for
// the saturated form.  The reduction then divides by 2^261 instead of 2^256; the operand that is wave-uniform or comes
// from a table (the fold challenge, NTT twiddles) is therefore prepared once as c * 2^5 mod p, split into 29-bit limbs:
//   fe_mul29(a, prepare(c)) = a * (c * 2^5) * 2^-261 = a * c * 2^-256 mod p = fe_mul(a, c),  bit for bit.
struct Mul29 {
    uint32_t l[9];
};
ZK_HD void split29(const uint32_t a[8], uint32_t out[9]) {
    constexpr uint32_t M = (1u << 29) - 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint32_t v = a[w] >> sh;
        if (sh > 3 && w + 1 < 8) v |= a[w + 1] << (32 - sh);   // the limb straddles two words
        out[i] = (i < 8) ? (v & M) : v;                          // limb 8 = bits 232..255
    }
}
ZK_HD Mul29 mul29_prepare(const Fe &c, const FieldParams &P) {
    Fe t = c;
#pragma unroll
    for (int i = 0; i < 5; ++i) t = fe_add(t, t, P);   // c * 2^5 mod p
    Mul29 m;
    split29(t.v, m.l);
    return m;
}
// LAZY = true leaves out the final conditional subtraction: the result is in [0, 2p) for ANY 256-bit a (a*c*2^-261 + p < 2p
// since c < p < 2^255 makes the first term < 2^250) -- the NTT keeps its values in [0, 2p) inside a transform and reduces once.
// TWO = true adds a second product before the ONE reduction: a*c + a2*c2 (a dot product of length two for the price of one and
// a half multiplications).  The columns stay carry-free: at most 18 products of two 29-bit limbs plus 9 of the reduction,
// 27 * 2^58 < 2^63; the value bound doubles to 2^251 + p, still below 2p.
// (the core works on the left operands' limbs: fe_mul29_t splits an element, fe_mul_tt splits it shifted by five bits)
template <bool LAZY = false, bool TWO = false>
ZK_HD Fe mul29_core(const uint32_t (&x)[9], const uint32_t (&y)[9], const Mul29 &c, const FieldParams &P, const Mul29 *c2 = nullptr) {
    constexpr uint32_t M = (1u << 29) - 1;
    uint32_t m[9], r[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)x[i] * c.l[k - i];
        if constexpr (TWO) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)y[i] * c2->l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P.p29[k - i];
        m[k] = ((uint32_t)acc * P.inv29) & M;
        acc += (uint64_t)m[k] * P.p29[0];   // low 29 bits are now zero
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)x[i] * c.l[k - i];
        if constexpr (TWO) {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)y[i] * c2->l[k - i];
        }
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * P.p29[k - i];
        r[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
    // merge the 29-bit limbs back into eight 32-bit words (value < 2p < 2^256)
    Fe s;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int bit = 32 * w, i = bit / 29, sh = bit - 29 * i;   // word w starts inside limb i at offset sh
        uint32_t v = r[i] >> sh;
        v |= r[i + 1] << (29 - sh);
        if (29 - sh + 29 < 32 && i + 2 < 9) v |= r[i + 2] << (58 - sh);
        s.v[w] = v;
    }
    if constexpr (LAZY) return s;
    Fe d, o;
    const uint32_t borrow = sub8(d.v, s.v, P.p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = borrow ? s.v[i] : d.v[i];
    return o;
}
template <bool LAZY = false, bool TWO = false>
ZK_HD Fe fe_mul29_t(const Fe &a, const Mul29 &c, const FieldParams &P, const Fe *a2 = nullptr, const Mul29 *c2 = nullptr) {
    uint32_t x[9], y[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    split29(a.v, x);
    if constexpr (TWO) split29(a2->v, y);
    return mul29_core<LAZY, TWO>(x, y, c, P, c2);
}
// Product of two TABLE values on the carry-free core (ProductPoly::prod_reduce, product_poly.rs:66-74: neither operand is
// uniform, so nothing can be prepared ahead).  The core divides by 2^261; the missing factor 2^5 goes into the left operand's
// SPLIT: a < p < 2^255, so a * 2^5 < 2^260 still fits nine 29-bit limbs -- the limbs are cut five bits lower, no doublings:
//   fe_mul_tt(a, b) = (a * 2^5) * b * 2^-261 = a * b * 2^-256 mod p = fe_mul(a, b), bit for bit
// (value bound (2^260 * 2^255) / 2^261 + p < 2p, one conditional subtraction; column sums as in fe_mul29).  About 290
// instructions against ~420 for the saturated product.
ZK_HD void split29_shl5(const uint32_t a[8], uint32_t out[9]) {
    constexpr uint32_t M = (1u << 29) - 1;
    out[0] = (a[0] << 5) & M;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int bit = 29 * i - 5, w = bit >> 5, sh = bit & 31;
        uint32_t v = a[w] >> sh;
        if (sh > 3 && w + 1 < 8) v |= a[w + 1] << (32 - sh);
        out[i] = v & M;   // limb 8 = bits 227..255 of a (29 bits)
    }
}
ZK_HD Fe fe_mul_tt(const Fe &a, const Fe &b, const FieldParams &P) {
    uint32_t x[9], y[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Mul29 c;
    split29_shl5(a.v, x);
    split29(b.v, c.l);
    return mul29_core<false, false>(x, y, c, P);
}
// a*c + a2*c2 (both right operands prepared), one reduction
ZK_HD Fe fe_dot2_29(const Fe &a, const Mul29 &c, const Fe &a2, const Mul29 &c2, const FieldParams &P) {
    return fe_mul29_t<false, true>(a, c, P, &a2, &c2);
}
ZK_HD Fe fe_mul29(const Fe &a, const Mul29 &c, const FieldParams &P) { return fe_mul29_t<false>(a, c, P); }

// ---- 32-byte element I/O (two 16-byte accesses: global_load_dwordx4 / global_store_dwordx4) ---------------
#if defined(__HIPCC__)
ZK_D Fe fe_load(const uint64_t *base, uint64_t idx) {
    const uint4 *q = reinterpret_cast<const uint4 *>(base + 4 * idx);
    uint4 a = q[0], b = q[1];
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
ZK_D void fe_store(uint64_t *base, uint64_t idx, const Fe &r) {
    uint4 *q = reinterpret_cast<uint4 *>(base + 4 * idx);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
#endif

}  // namespace zk
