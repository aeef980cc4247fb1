// mb_limbpar.hip -- a LIMB-PARALLEL Montgomery product for the transcript wave's dependent multiplications.
//
// The prover's serial chain evaluates S(t) = e_0 + r (c_1 + r e_inf) on ONE wave per round: K dependent multiplications by the wave-uniform
// challenge, each a fe_mul29 of ~260 instructions that a lone wave issues in ~0.8 us -- 63 of its 64 lanes compute nothing (or a copy).
// Here one value occupies nine lanes, lane i holding limb i (29 bits), and the product is word-serial Montgomery (operand scanning) with
// the lanes as the limb positions:
//     for j = 0..8:   t_i += a_i * c_j                   one v_mad_u64_u32 for all limbs  (c_j wave-uniform: SGPR)
//                     m = (t_0 * (-p^-1)) mod 2^29        v_readlane + two scalar instructions
//                     t_i += m * p_i                      one v_mad_u64_u32 (p_i: lane i's limb of p)
//                     t_i <- t_{i+1}, t_0 += old t_0 >> 29 DPP row_shl:1 + lane 0's carry (two v_readlane, scalar shift)
// ~11 vector + 4 scalar instructions per step instead of ~29, but every one of them depends on the one before.  Column sums stay below
// 2^63 (18 products of 29 x 29 bits + carries), so no carry chain inside the loop; the result's limbs are normalised to < 2^30 by two
// lane-shift passes (good enough as the next product's left operand) and only the LAST product of a chain is carried out exactly and
// reduced (in one lane, ordinary code).
// Checked bit-exact against fe_mul29 (single products and a two-product Horner chain) before anything is timed; timed as a dependent
// chain on one wave, as the transcript wave runs it.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_limbpar.hip -o tools/mb/bin/mb_limbpar
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_field.hpp"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr uint32_t M29 = (1u << 29) - 1;
__device__ __forceinline__ uint32_t shl1(uint32_t v) {   // lane i reads lane i + 1 of its row (row_shl:1), out of range -> 0
    return __builtin_amdgcn_update_dpp(0u, v, 0x101, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t shr1(uint32_t v) {   // lane i reads lane i - 1 (row_shr:1), lane 0 -> 0
    return __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xF, 0xF, true);
}
// limbs of a 256-bit value (lane 0's Fe) spread over lanes 0..8; every lane of the wave calls it
__device__ __forceinline__ uint32_t spread(const Fe &x_lane0, uint32_t lane) {
    uint32_t l[9];
    split29(x_lane0.v, l);
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t s = __builtin_amdgcn_readlane(l[i], 0);
        mine = lane == (uint32_t)i ? s : mine;
    }
    return mine;
}
// a (limb per lane, < 2^30) times the prepared multiplier c (wave-uniform) -> a * c * 2^-261 mod p (+ k p), limbs < 2^30 per lane
__device__ __forceinline__ uint32_t limbpar_mul(uint32_t a, const Mul29 &c, uint32_t p_limb, uint32_t inv29, uint32_t lane) {
    uint64_t t = 0;
    const uint32_t is0 = lane == 0 ? 0xffffffffu : 0u;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const uint32_t cj = __builtin_amdgcn_readfirstlane(c.l[j]);
        t += (uint64_t)a * cj;
        const uint32_t t0 = __builtin_amdgcn_readlane((uint32_t)t, 0);
        const uint32_t m = (t0 * inv29) & M29;
        t += (uint64_t)p_limb * m;
        const uint32_t lo0 = __builtin_amdgcn_readlane((uint32_t)t, 0), hi0 = __builtin_amdgcn_readlane((uint32_t)(t >> 32), 0);
        const uint64_t carry = (((uint64_t)hi0 << 32) | lo0) >> 29;   // lane 0's column is divisible by 2^29
        const uint32_t nlo = shl1((uint32_t)t), nhi = shl1((uint32_t)(t >> 32));
        t = (((uint64_t)nhi << 32) | nlo) + (((uint64_t)((uint32_t)(carry >> 32) & is0) << 32) | ((uint32_t)carry & is0));
    }
    // two lane-shift passes: limbs < 2^29 + 2^5.  Lane 8 is the top limb: it keeps its high bits (the value is below 2p < 2^256, so it stays
    // below 2^24) and hands nothing on; lanes above it hold no limb
    const uint64_t keep = lane == 8 ? ~0ull : (uint64_t)M29;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const uint64_t up = lane == 8 ? 0ull : t >> 29;
        const uint32_t ulo = shr1((uint32_t)up), uhi = shr1((uint32_t)(up >> 32));
        t = (t & keep) + (((uint64_t)uhi << 32) | ulo);
    }
    return lane < 9 ? (uint32_t)t : 0u;
}
// gather the limbs back into lane-uniform words, carry exactly, reduce once (value < 2p + slack from unnormalised inputs: two subtractions)
__device__ __forceinline__ Fe collect(uint32_t limb, const FieldParams &P) {
    uint64_t acc = 0;
    uint32_t r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        acc += __builtin_amdgcn_readlane(limb, i);
        r[i] = i < 8 ? (uint32_t)acc & M29 : (uint32_t)acc;
        acc >>= 29;
    }
    Fe s;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int bit = 32 * w, i = bit / 29, sh = bit - 29 * i;
        uint32_t v = r[i] >> sh;
        v |= r[i + 1] << (29 - sh);
        if (29 - sh + 29 < 32 && i + 2 < 9) v |= r[i + 2] << (58 - sh);
        s.v[w] = v;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        Fe d;
        const uint32_t borrow = sub8(d.v, s.v, P.p);
#pragma unroll
        for (int i = 0; i < 8; ++i) s.v[i] = borrow ? s.v[i] : d.v[i];
    }
    return s;
}

__device__ __forceinline__ uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ Fe random_fe(uint64_t &s) {
    Fe x;
    for (int i = 0; i < 4; ++i) {
        const uint64_t z = splitmix(s);
        x.v[2 * i] = (uint32_t)z;
        x.v[2 * i + 1] = (uint32_t)(z >> 32);
    }
    x.v[7] &= 0x0fffffffu;   // < 2^252 < p
    return x;
}

// one wave per block; every block checks 64 random cases
__global__ void k_check(FieldParams P, unsigned long long *bad) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t seed = 0xC0FFEE + blockIdx.x;   // wave-uniform
    const uint32_t p_limb = lane < 9 ? P.p29[lane < 9 ? lane : 0] : 0u;
    for (int rep = 0; rep < 64; ++rep) {
        Fe x = random_fe(seed), y = random_fe(seed), z = random_fe(seed);
        if (rep == 1) x = fe_zero();
        if (rep == 2) {
            Fe one = {{1, 0, 0, 0, 0, 0, 0, 0}};
            x = fe_sub(fe_zero(), one, P);   // p - 1
        }
        const Mul29 c = mul29_prepare(y, P);
        // single product
        const Fe want1 = fe_mul29(x, c, P);
        const Fe got1 = collect(limbpar_mul(spread(x, lane), c, p_limb, P.inv29, lane), P);
        // the Horner chain of K = 2:  (z + x * c) * c, the intermediate kept in limb-parallel form (limbs added lane-wise)
        const Fe want2 = fe_mul29(fe_add(z, fe_mul29(x, c, P), P), c, P);
        const uint32_t mid = limbpar_mul(spread(x, lane), c, p_limb, P.inv29, lane) + spread(z, lane);
        const Fe got2 = collect(limbpar_mul(mid, c, p_limb, P.inv29, lane), P);
        if (lane == 0 && (!fe_eq(want1, got1) || !fe_eq(want2, got2))) atomicAdd(bad, 1ull);
    }
}

// a dependent chain on ONE wave: variant 0 = fe_mul29 (lane 0's value; the other lanes compute copies, as in the transcript wave),
// variant 1 = limb-parallel (the value lives in limb form across the chain; spread at the start, collect at the end)
__global__ void k_chain(FieldParams P, Fe x0, Mul29 c, int iters, int variant, Fe *out, int zero) {
    const uint32_t lane = threadIdx.x & 63;
    if (variant == 0) {
        Fe x = x0;
        x.v[0] += lane * (uint32_t)zero;   // (zero == 0: keeps the chain on the vector ALU, as in the transcript wave)
        for (int i = 0; i < iters; ++i) x = fe_mul29(x, c, P);
        if (lane == 0) *out = x;
    } else {
        const uint32_t p_limb = lane < 9 ? P.p29[lane < 9 ? lane : 0] : 0u;
        uint32_t a = spread(x0, lane);
        for (int i = 0; i < iters; ++i) a = limbpar_mul(a, c, p_limb, P.inv29, lane);
        const Fe x = collect(a, P);
        if (lane == 0) *out = x;
    }
}
// what the transcript wave does per round: value in ordinary form -> two dependent products -> ordinary form again
__global__ void k_round_like(FieldParams P, Fe x0, Fe z, Mul29 c, int iters, int variant, Fe *out, int zero) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t p_limb = lane < 9 ? P.p29[lane < 9 ? lane : 0] : 0u;
    Fe x = x0;
    if (variant == 0) x.v[0] += lane * (uint32_t)zero;
    for (int i = 0; i < iters; ++i) {
        if (variant == 0) {
            x = fe_mul29(fe_add(z, fe_mul29(x, c, P), P), c, P);
        } else {
            const uint32_t mid = limbpar_mul(spread(x, lane), c, p_limb, P.inv29, lane) + spread(z, lane);
            x = collect(limbpar_mul(mid, c, p_limb, P.inv29, lane), P);
        }
    }
    if (lane == 0) *out = x;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const FieldInfo *fi = field_info(0);
    const FieldParams &P = fi->P;
    unsigned long long *d_bad, bad = 0;
    Fe *d_out;
    CK(hipMalloc(&d_bad, 8));
    CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_bad, 0, 8));
    k_check<<<256, 64>>>(P, d_bad);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    printf("check: %llu of %d cases (single product + two-product Horner chain) differ from fe_mul29\n", bad, 256 * 64);
    if (bad) return 1;
    const Fe x0 = fe_pow_u64(fi->two_adic_root, 777, P), z = fe_pow_u64(fi->two_adic_root, 999, P);
    const Mul29 c = mul29_prepare(fe_pow_u64(fi->two_adic_root, 31337, P), P);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    Fe res[2];
    for (int round = 0; round < 3; ++round) {
        for (int variant = 0; variant < 2; ++variant) {
            k_chain<<<1, 64>>>(P, x0, c, 16, variant, d_out, 0);
            CK(hipEventRecord(e0));
            k_chain<<<1, 64>>>(P, x0, c, iters, variant, d_out, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&res[variant], d_out, sizeof(Fe), hipMemcpyDeviceToHost));
            printf("chain of %d dependent products on one wave, %s: %.3f us per product\n", iters, variant ? "limb-parallel" : "fe_mul29     ", ms * 1e3 / iters);
        }
        printf("   results %s\n", memcmp(&res[0], &res[1], sizeof(Fe)) == 0 ? "identical" : "DIFFER");
        for (int variant = 0; variant < 2; ++variant) {
            k_round_like<<<1, 64>>>(P, x0, z, c, 16, variant, d_out, 0);
            CK(hipEventRecord(e0));
            k_round_like<<<1, 64>>>(P, x0, z, c, iters, variant, d_out, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&res[variant], d_out, sizeof(Fe), hipMemcpyDeviceToHost));
            printf("per-round shape (ordinary form -> 2 dependent products -> ordinary form), %s: %.3f us per round\n",
                   variant ? "limb-parallel" : "fe_mul29     ", ms * 1e3 / iters);
        }
        printf("   results %s\n", memcmp(&res[0], &res[1], sizeof(Fe)) == 0 ? "identical" : "DIFFER");
    }
    return 0;
}
