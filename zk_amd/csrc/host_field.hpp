// host_field.hpp -- host-side field tables and conversions for libzk_amd (product code, not the oracle).
//
// Supplies the FieldParams handed to every kernel and the few host-side element operations the protocol layer
// needs between kernels (transcript/src/lib.rs:27-30 from_be_bytes_mod_order; sumcheck/src/lib.rs:23-29
// into_bigint().to_bytes_be(); fft/src/lib.rs:6,14,17 get_root_of_unity / inverse).  Arithmetic itself is the
// shared field.cuh code compiled for the host.
#pragma once
#include <stdint.h>
#include <string.h>

#include "field.cuh"

namespace zk {

struct FieldInfo {
    FieldParams P;
    uint32_t two_adicity;
    uint32_t generator;   // ark-ff GENERATOR
    Fe two_adic_root;     // GENERATOR^((p-1)/2^s), Montgomery form
    bool ready;
};

inline const uint64_t *field_modulus_limbs(int field) {
    // moduli as 4 LE u64 limbs (ark-bn254 / ark-bls12-381 / ark-bls12-377 Fr, v0.5.0)
    static const uint64_t M[3][4] = {
        {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
        {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
        {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL},
    };
    return (field >= 0 && field < 3) ? M[field] : nullptr;
}

inline Fe fe_from_u64limbs(const uint64_t l[4]) {
    Fe r;
    for (int i = 0; i < 4; ++i) {
        r.v[2 * i] = (uint32_t)l[i];
        r.v[2 * i + 1] = (uint32_t)(l[i] >> 32);
    }
    return r;
}
inline void fe_to_u64limbs(const Fe &a, uint64_t l[4]) {
    for (int i = 0; i < 4; ++i) l[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
}

// x^e for a 256-bit exponent (8 x u32 limbs), MSB-first square and multiply
inline Fe fe_pow_limbs(const Fe &x, const uint32_t e[8], const FieldParams &P) {
    Fe acc = fe_one(P);
    for (int i = 255; i >= 0; --i) {
        acc = fe_sqr(acc, P);
        if ((e[i / 32] >> (i % 32)) & 1u) acc = fe_mul(acc, x, P);
    }
    return acc;
}
inline Fe fe_pow_u64(const Fe &x, uint64_t e, const FieldParams &P) {
    Fe acc = fe_one(P), base = x;
    while (e) {
        if (e & 1) acc = fe_mul(acc, base, P);
        base = fe_sqr(base, P);
        e >>= 1;
    }
    return acc;
}
// a^-1 by Fermat (a != 0)
inline Fe fe_inverse(const Fe &a, const FieldParams &P) {
    uint32_t e[8], two[8] = {2, 0, 0, 0, 0, 0, 0, 0};
    sub8(e, P.p, two);
    return fe_pow_limbs(a, e, P);
}

inline const FieldInfo *field_info(int field) {
    static FieldInfo info[3];
    static const uint32_t adicity[3] = {28, 32, 47};
    static const uint32_t gen[3] = {5, 7, 22};
    if (field < 0 || field > 2) return nullptr;
    FieldInfo &I = info[field];
    if (I.ready) return &I;
    FieldParams &P = I.P;
    Fe pm = fe_from_u64limbs(field_modulus_limbs(field));
    memcpy(P.p, pm.v, 32);
    // -p^-1 mod 2^32 (Newton)
    uint32_t x = 1;
    for (int i = 0; i < 5; ++i) x *= 2u - P.p[0] * x;
    P.inv = 0u - x;
    // bit length
    int top = 7;
    while (top > 0 && P.p[top] == 0) --top;
    P.bits = 32 * top + (32 - __builtin_clz(P.p[top]));
    // R mod p, R^2 mod p: 256 / 512 modular doublings of 1
    uint32_t v[8] = {1, 0, 0, 0, 0, 0, 0, 0}, d[8];
    for (int i = 0; i < 512; ++i) {
        uint32_t c = add8(v, v, v);
        uint32_t b = sub8(d, v, P.p);
        if (c | (b ^ 1u)) memcpy(v, d, 32);
        if (i == 255) memcpy(P.r1, v, 32);
    }
    memcpy(P.r2, v, 32);
    // 29-bit limb view of p and -p^-1 mod 2^29 for the unsaturated multiplier
    split29(P.p, P.p29);
    P.inv29 = P.inv & ((1u << 29) - 1);
    {
        Fe r2;
        memcpy(r2.v, P.r2, 32);
        const Mul29 k0 = mul29_prepare(r2, P);
        Fe r2s = r2;
        for (int i = 0; i < 5; ++i) r2s = fe_add(r2s, r2s, P);
        const Mul29 k1 = mul29_prepare(r2s, P);
        memcpy(P.r2_29, k0.l, sizeof k0.l);
        memcpy(P.r2s_29, k1.l, sizeof k1.l);
    }
    // redc_wide's contract (field.cuh): R/p + 1 + top_max < 32 with top_max = floor(kMaxLazy * p^2 / 2^512).  Power-of-two upper
    // bounds of both terms suffice for the shipped fields (11, 13 and 17); a modulus that fails is refused here.
    {
        const int rp_log = 257 - (int)P.bits;                                  // R / p < 2^(257 - bits)
        int lazy_log = 0;
        while ((1 << lazy_log) < kMaxLazy) ++lazy_log;
        const int top_log = lazy_log + 2 * (int)P.bits - 512;                  // kMaxLazy * p^2 / 2^512 < 2^top_log
        const uint64_t rp = rp_log >= 0 && rp_log < 32 ? (1ull << rp_log) : ~0ull;
        const uint64_t top = top_log < 0 ? 0 : (top_log < 32 ? (1ull << top_log) : ~0ull);
        if (rp_log < 0 || rp + 1 + top >= 32 || rp + 1 + top < rp) return nullptr;
    }
    I.two_adicity = adicity[field];
    I.generator = gen[field];
    // TWO_ADIC_ROOT_OF_UNITY = g^t with p - 1 = 2^s * t
    uint32_t t[8], one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    sub8(t, P.p, one);
    for (uint32_t s = 0; s < I.two_adicity; ++s) {
        for (int i = 0; i < 7; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
        t[7] >>= 1;
    }
    I.two_adic_root = fe_pow_limbs(fe_from_u32(I.generator, P), t, P);
    I.ready = true;
    return &I;
}

// F::get_root_of_unity(n) for n = 2^log_n (fft/src/lib.rs:6): TWO_ADIC_ROOT squared (s - log_n) times
inline bool field_root_of_unity(const FieldInfo &I, uint32_t log_n, Fe &out) {
    if (log_n > I.two_adicity) return false;
    Fe w = I.two_adic_root;
    for (uint32_t i = log_n; i < I.two_adicity; ++i) w = fe_sqr(w, I.P);
    out = w;
    return true;
}

// elem.into_bigint().to_bytes_be() -- 32 bytes, big endian, canonical
inline void fe_to_bytes_be(const Fe &a, const FieldParams &P, uint8_t out[32]) {
    Fe c = fe_to_canonical(a, P);
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) out[31 - (4 * i + b)] = (uint8_t)(c.v[i] >> (8 * b));
}
// F::from_be_bytes_mod_order: int(bytes) mod p, as a Montgomery element.  Horner in base 256.
inline Fe fe_from_be_bytes_mod_order(const uint8_t *bytes, size_t len, const FieldParams &P) {
    Fe acc = fe_zero();
    const Fe c256 = fe_from_u32(256, P);
    for (size_t i = 0; i < len; ++i) acc = fe_add(fe_mul(acc, c256, P), fe_from_u32(bytes[i], P), P);
    return acc;
}

}  // namespace zk
