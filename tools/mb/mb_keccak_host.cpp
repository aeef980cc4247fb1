// mb_keccak_host.cpp -- the host permutation behind `prove` / `verify` (a serial Keccak-256 sponge over k * 2^n * 32 table bytes bounds those
// calls: 1.42 s of the 1.43 s of an n = 24 `prove`): keccak_host::rounds (scalar, BMI andn) against one state held in 25 xmm registers
// with AVX-512VL rotates and ternary logic (95 vector operations per round instead of ~130 scalar ones, no spills).
// Measured (clang -O3): Xeon 2.6 GHz 393 -> 251 ns per permutation; EPYC 9575F (the MI355X box's host) 170 -> 193 ns -- slower where it
// would be used, so the library keeps the scalar form.  Build: clang++ -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_keccak_host.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <immintrin.h>
#include "keccak.hpp"
using namespace zk;
// 25 lanes in 25 xmm registers (low 64 bits used): vprolq / vpternlogq (AVX-512VL), no spills with 32 registers
__attribute__((target("avx512f,avx512vl"))) static void permute_vl(uint64_t *s) {
    __m128i a[25];
    for (int i = 0; i < 25; ++i) a[i] = _mm_cvtsi64_si128((long long)s[i]);
    for (int round = 0; round < 24; ++round) {
        __m128i c[5], d[5], b[25];
#pragma GCC unroll 5
        for (int x = 0; x < 5; ++x) c[x] = _mm_ternarylogic_epi64(_mm_ternarylogic_epi64(a[x], a[x + 5], a[x + 10], 0x96), a[x + 15], a[x + 20], 0x96);
#pragma GCC unroll 5
        for (int x = 0; x < 5; ++x) d[x] = _mm_xor_si128(c[(x + 4) % 5], _mm_rol_epi64(c[(x + 1) % 5], 1));
#define RP(I) { constexpr int X = (I) % 5, Y = (I) / 5; b[Y + 5 * ((2 * X + 3 * Y) % 5)] = _mm_rol_epi64(_mm_xor_si128(a[I], d[X]), keccak_host::kRho[X][Y]); }
        RP(0) RP(1) RP(2) RP(3) RP(4) RP(5) RP(6) RP(7) RP(8) RP(9) RP(10) RP(11) RP(12) RP(13) RP(14) RP(15) RP(16) RP(17) RP(18) RP(19) RP(20) RP(21) RP(22) RP(23) RP(24)
#undef RP
#pragma GCC unroll 5
        for (int y = 0; y < 25; y += 5) {
#pragma GCC unroll 5
            for (int x = 0; x < 5; ++x) a[y + x] = _mm_ternarylogic_epi64(b[y + x], b[y + (x + 1) % 5], b[y + (x + 2) % 5], 0xD2);
        }
        a[0] = _mm_xor_si128(a[0], _mm_cvtsi64_si128((long long)keccak_host::kRC[round]));
    }
    for (int i = 0; i < 25; ++i) s[i] = (uint64_t)_mm_cvtsi128_si64(a[i]);
}
int main() {
    uint64_t s1[25], s2[25];
    for (int i = 0; i < 25; ++i) s1[i] = s2[i] = 0x9E3779B97F4A7C15ull * (i + 1);
    for (int i = 0; i < 1000; ++i) { keccak_host::permute_bmi(s1); permute_vl(s2); }
    printf("equal: %d\n", memcmp(s1, s2, sizeof s1) == 0);
    const int N = 3000000;
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) keccak_host::permute_bmi(s1);
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) permute_vl(s2);
        auto t2 = std::chrono::steady_clock::now();
        printf("scalar+bmi %.1f ns/perm   avx512vl %.1f ns/perm   (%llx %llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / N,
               std::chrono::duration<double, std::nano>(t2 - t1).count() / N, (unsigned long long)s1[0], (unsigned long long)s2[0]);
    }
}
