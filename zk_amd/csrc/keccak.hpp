// keccak.hpp -- Keccak-256 sponge + the reference's Fiat-Shamir transcript (product code; host and device).
//
// transcript/src/lib.rs uses sha3 0.10.8 `Keccak256`: Keccak-f[1600], rate 136 bytes, original Keccak padding
// (domain byte 0x01 ... 0x80) -- not NIST SHA3-256.  `Sponge` is usable from host code and from a single GPU
// lane (the on-device transcript keeps the prover's round loop free of host round trips).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZKK_HD __host__ __device__
#else
#define ZKK_HD
#endif

namespace zk {

ZKK_HD inline uint64_t keccak_rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

ZKK_HD inline void keccak_f1600(uint64_t a[25]) {
    const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    for (int round = 0; round < 24; ++round) {
        // theta
        uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];
        uint64_t c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
        uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];
        uint64_t c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
        uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        uint64_t d0 = c4 ^ keccak_rotl(c1, 1), d1 = c0 ^ keccak_rotl(c2, 1), d2 = c1 ^ keccak_rotl(c3, 1);
        uint64_t d3 = c2 ^ keccak_rotl(c4, 1), d4 = c3 ^ keccak_rotl(c0, 1);
        for (int y = 0; y < 25; y += 5) {
            a[y] ^= d0;
            a[y + 1] ^= d1;
            a[y + 2] ^= d2;
            a[y + 3] ^= d3;
            a[y + 4] ^= d4;
        }
        // rho + pi (in-place lane walk)
        uint64_t cur = a[1], nxt;
        nxt = a[10]; a[10] = keccak_rotl(cur, 1);  cur = nxt;
        nxt = a[7];  a[7]  = keccak_rotl(cur, 3);  cur = nxt;
        nxt = a[11]; a[11] = keccak_rotl(cur, 6);  cur = nxt;
        nxt = a[17]; a[17] = keccak_rotl(cur, 10); cur = nxt;
        nxt = a[18]; a[18] = keccak_rotl(cur, 15); cur = nxt;
        nxt = a[3];  a[3]  = keccak_rotl(cur, 21); cur = nxt;
        nxt = a[5];  a[5]  = keccak_rotl(cur, 28); cur = nxt;
        nxt = a[16]; a[16] = keccak_rotl(cur, 36); cur = nxt;
        nxt = a[8];  a[8]  = keccak_rotl(cur, 45); cur = nxt;
        nxt = a[21]; a[21] = keccak_rotl(cur, 55); cur = nxt;
        nxt = a[24]; a[24] = keccak_rotl(cur, 2);  cur = nxt;
        nxt = a[4];  a[4]  = keccak_rotl(cur, 14); cur = nxt;
        nxt = a[15]; a[15] = keccak_rotl(cur, 27); cur = nxt;
        nxt = a[23]; a[23] = keccak_rotl(cur, 41); cur = nxt;
        nxt = a[19]; a[19] = keccak_rotl(cur, 56); cur = nxt;
        nxt = a[13]; a[13] = keccak_rotl(cur, 8);  cur = nxt;
        nxt = a[12]; a[12] = keccak_rotl(cur, 25); cur = nxt;
        nxt = a[2];  a[2]  = keccak_rotl(cur, 43); cur = nxt;
        nxt = a[20]; a[20] = keccak_rotl(cur, 62); cur = nxt;
        nxt = a[14]; a[14] = keccak_rotl(cur, 18); cur = nxt;
        nxt = a[22]; a[22] = keccak_rotl(cur, 39); cur = nxt;
        nxt = a[9];  a[9]  = keccak_rotl(cur, 61); cur = nxt;
        nxt = a[6];  a[6]  = keccak_rotl(cur, 20); cur = nxt;
        a[1] = keccak_rotl(cur, 44);
        // chi
        for (int y = 0; y < 25; y += 5) {
            uint64_t b0 = a[y], b1 = a[y + 1], b2 = a[y + 2], b3 = a[y + 3], b4 = a[y + 4];
            a[y] = b0 ^ (~b1 & b2);
            a[y + 1] = b1 ^ (~b2 & b3);
            a[y + 2] = b2 ^ (~b3 & b4);
            a[y + 3] = b3 ^ (~b4 & b0);
            a[y + 4] = b4 ^ (~b0 & b1);
        }
        a[0] ^= RC[round];
    }
}

// ---- host permutation -----------------------------------------------------------------------------------------------------
// `prove` / `verify` absorb the tables' byte image (k * 2^n * 32 bytes) through ONE sponge (prover.rs:17, verifier.rs:22): serial
// by construction, so the host permutation bounds those calls.  Same permutation as keccak_f1600 above, written for a CPU:
// rho + pi as 25 independent rotations into a second array (the in-place lane walk above is one 24-step dependency chain), every
// index and rotation a compile-time constant so both arrays live in registers, and BMI1 `andn` for chi where the CPU has it
// (measured, Xeon 2.1 GHz, clang -O3: 778 -> 445 ns per permutation, 306 ns with andn).
namespace keccak_host {
constexpr int kRho[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};   // [x][y]
constexpr uint64_t kRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
template <int N>
inline __attribute__((always_inline)) uint64_t rol(uint64_t v) {
    if constexpr (N == 0) return v;
    else return (v << N) | (v >> (64 - N));
}
template <int I>   // lane I = x + 5y moves to (y, 2x + 3y)
inline __attribute__((always_inline)) void rho_pi(const uint64_t *a, const uint64_t *d, uint64_t *b) {
    constexpr int X = I % 5, Y = I / 5;
    b[Y + 5 * ((2 * X + 3 * Y) % 5)] = rol<kRho[X][Y]>(a[I] ^ d[X]);
    if constexpr (I + 1 < 25) rho_pi<I + 1>(a, d, b);
}
inline __attribute__((always_inline)) void rounds(uint64_t *s) {
    uint64_t a[25];
    for (int i = 0; i < 25; ++i) a[i] = s[i];
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
#pragma GCC unroll 5
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma GCC unroll 5
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol<1>(c[(x + 1) % 5]);
        rho_pi<0>(a, d, b);
#pragma GCC unroll 5
        for (int y = 0; y < 25; y += 5) {
#pragma GCC unroll 5
            for (int x = 0; x < 5; ++x) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
        }
        a[0] ^= kRC[round];
    }
    for (int i = 0; i < 25; ++i) s[i] = a[i];
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("bmi"))) inline void permute_bmi(uint64_t *s) { rounds(s); }
inline bool has_bmi() {
    static const bool v = __builtin_cpu_supports("bmi");
    return v;
}
#else
inline void permute_bmi(uint64_t *s) { rounds(s); }
inline bool has_bmi() { return false; }
#endif
inline void permute_plain(uint64_t *s) { rounds(s); }
}  // namespace keccak_host
// the permutation a sponge runs: the lane-walk form in device code, the CPU form on the host
ZKK_HD inline void keccak_permute(uint64_t a[25]) {
#if defined(__HIP_DEVICE_COMPILE__)
    keccak_f1600(a);
#else
    if (keccak_host::has_bmi()) keccak_host::permute_bmi(a);
    else keccak_host::permute_plain(a);
#endif
}

// Incremental sponge with sha3::Keccak256 semantics (update / finalize_reset).
struct Sponge {
    uint64_t s[25];
    uint8_t buf[136];
    uint32_t fill;

    ZKK_HD void init() {
        for (int i = 0; i < 25; ++i) s[i] = 0;
        fill = 0;
    }
    ZKK_HD void absorb_block(const uint8_t *blk) {
        for (int i = 0; i < 17; ++i) {
            uint64_t w = 0;
            for (int b = 0; b < 8; ++b) w |= (uint64_t)blk[8 * i + b] << (8 * b);
            s[i] ^= w;
        }
        keccak_permute(s);
    }
    ZKK_HD void update(const uint8_t *data, size_t len) {
        while (len) {
            if (fill == 0 && len >= 136) {   // whole blocks straight from the input
                absorb_block(data);
                data += 136;
                len -= 136;
                continue;
            }
            size_t take = 136 - fill;
            if (take > len) take = len;
            for (size_t i = 0; i < take; ++i) buf[fill + i] = data[i];
            fill += (uint32_t)take;
            data += take;
            len -= take;
            if (fill == 136) {
                absorb_block(buf);
                fill = 0;
            }
        }
    }
    ZKK_HD void finalize_reset(uint8_t out[32]) {
        for (uint32_t i = fill; i < 136; ++i) buf[i] = 0;
        buf[fill] ^= 0x01;
        buf[135] ^= 0x80;
        absorb_block(buf);
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(s[i] >> (8 * b));
        init();
    }
    // transcript/src/lib.rs:20-25 sample_challenge: h = finalize_reset(); update(h); return h
    ZKK_HD void sample_challenge(uint8_t out[32]) {
        finalize_reset(out);
        update(out, 32);
    }
};


// The same sponge kept as 64-bit lanes with a WORD cursor: every message the prover absorbs is a whole number of
// 8-byte words (32-byte big-endian elements, 32-byte digests, tables of 32-byte elements), so no byte buffer is needed
// and the state lives in registers of a single GPU lane.  Absorbing the 32-byte big-endian image of a 256-bit integer
// with little-endian u64 limbs l0..l3 XORs bswap(l3), bswap(l2), bswap(l1), bswap(l0) into consecutive lanes.
struct WordSponge {
    uint64_t s[25];
    uint32_t pos;   // next lane to absorb into, 0..16
    uint32_t pad_;

    ZKK_HD static uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }
    ZKK_HD void init() {
        for (int i = 0; i < 25; ++i) s[i] = 0;
        pos = 0;
        pad_ = 0;
    }
    ZKK_HD void absorb_word(uint64_t w) {
        // pos is data dependent: select the lane without dynamic register indexing
#pragma unroll
        for (int i = 0; i < 17; ++i)
            if ((uint32_t)i == pos) s[i] ^= w;
        if (++pos == 17) {
            keccak_permute(s);
            pos = 0;
        }
    }
    // 32-byte big-endian image of the integer with LE u64 limbs l[0..4)
    ZKK_HD void absorb_u256_be(const uint64_t l[4]) {
        absorb_word(bswap64(l[3]));
        absorb_word(bswap64(l[2]));
        absorb_word(bswap64(l[1]));
        absorb_word(bswap64(l[0]));
    }
    // transcript/src/lib.rs:20-25: digest = finalize_reset(); update(digest).  Returns the digest as the LE u64 limbs
    // of int(digest, big endian) -- the integer from_be_bytes_mod_order reduces (transcript/src/lib.rs:29).
    ZKK_HD void sample_challenge_u256(uint64_t out_le_limbs[4]) {
#pragma unroll
        for (int i = 0; i < 17; ++i)
            if ((uint32_t)i == pos) s[i] ^= 0x01ull;
        s[16] ^= 0x8000000000000000ull;
        keccak_permute(s);
        const uint64_t d0 = s[0], d1 = s[1], d2 = s[2], d3 = s[3];
        out_le_limbs[0] = bswap64(d3);
        out_le_limbs[1] = bswap64(d2);
        out_le_limbs[2] = bswap64(d1);
        out_le_limbs[3] = bswap64(d0);
        for (int i = 0; i < 25; ++i) s[i] = 0;
        s[0] = d0;
        s[1] = d1;
        s[2] = d2;
        s[3] = d3;
        pos = 4;
    }
    // continue a byte sponge whose buffered length is a multiple of 8 (always true for the prover's messages)
    ZKK_HD bool from_byte_sponge(const Sponge &b) {
        if (b.fill % 8) return false;
        for (int i = 0; i < 25; ++i) s[i] = b.s[i];
        pos = 0;
        pad_ = 0;
        for (uint32_t w = 0; w < b.fill / 8; ++w) {
            uint64_t x = 0;
            for (int k = 0; k < 8; ++k) x |= (uint64_t)b.buf[8 * w + k] << (8 * k);
            s[w] ^= x;
        }
        pos = b.fill / 8;
        return true;
    }
};

}  // namespace zk
