// mb_shoup.hip -- a constant-operand (precomputed-quotient, "Shoup") multiplier for TABLE twiddles against fe_mul29_t<true>.
// ("one more multiplier experiment, then stop".)
//
// The NTT's multiplications are all by table values (twiddles), so each twiddle w may carry a second precomputed word
//     w' = floor(w * 2^261 / p)                                  (nine 29-bit limbs, like w itself)
// and x * w mod p needs no Montgomery reduction pass:
//     q  = floor(x * w' / 2^261)        the HIGH half of one 9 x 9 limb product
//     r  = (x * w - q * p) mod 2^261    the LOW halves of two 9 x 9 limb products
// With x < 2^261 and the exact q, r lies in [0, 2p).  Only the columns >= 7 of x * w' are formed (the dropped ones are worth < 2^235
// < 2^261, so the quotient is at most one short) => r in [0, 3p): 53 + 45 + 45 = 143 v_mad_u64_u32 against 162 for fe_mul29 (81 product
// + 81 reduction), carry-free columns both (<= 9 products of 29 x 29 bits per column: < 2^62).  3p < 2^256 holds for BN254 Fr and
// BLS12-377 Fr, not for BLS12-381 Fr (p = 0.906 * 2^255): that field would need the exact quotient (all 81 + 90 products: no gain).
//
// Domain: table values are Montgomery residues; with W = w (CANONICAL) the product x_mont * w is (x w)_mont -- the same element
// fe_mul(x_mont, w_mont) yields, modulo p.  Every lane checks that: r mod p == fe_mul(x, w_mont) for 4096 random (x, w) pairs and
// for x at the top of the lazy range, before anything is timed.
//
// Timed: two independent chains per lane (as zk_bench_modmul), the twiddle operand per lane in VGPRs (a table value, not a wave-uniform
// one), values kept lazy (< 3p resp. < 2p) as the NTT keeps them.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_shoup.hip -o tools/mb/bin/mb_shoup
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_field.hpp"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

struct Shoup29 {
    uint32_t w[9];    // the canonical twiddle, nine 29-bit limbs
    uint32_t wq[9];   // floor(w * 2^261 / p)
};

// x (8 words, any value < 2^256) times the table constant: result in [0, 3p), congruent to x * w mod p
__device__ __forceinline__ Fe shoup_mul(const Fe &xv, const Shoup29 &c, const FieldParams &P) {
    constexpr uint32_t M = (1u << 29) - 1;
    uint32_t x[9];
    split29(xv.v, x);
    // ---- quotient estimate: columns 7..16 of x * wq, carried up; limbs 9..17 are q
    uint64_t acc = 0;
#pragma unroll
    for (int k = 7; k <= 8; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)x[i] * c.wq[k - i];
        acc >>= 29;
    }
    uint32_t q[9];
#pragma unroll
    for (int k = 9; k <= 16; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)x[i] * c.wq[k - i];
        q[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    q[8] = (uint32_t)acc;
    // ---- r = (x * w - q * p) mod 2^261: low nine columns of both products, column differences carried as signed values
    uint32_t r[9];
    int64_t d = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        uint64_t a = 0, b = 0;
#pragma unroll
        for (int i = 0; i <= k; ++i) a += (uint64_t)x[i] * c.w[k - i];
#pragma unroll
        for (int i = 0; i <= k; ++i) b += (uint64_t)q[i] * P.p29[k - i];
        d += (int64_t)(a - b);
        r[k] = (uint32_t)d & M;
        d >>= 29;   // arithmetic
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
    return s;
}

__device__ __forceinline__ uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ Fe random_below_p(uint64_t &s, const FieldParams &P) {
    for (;;) {
        Fe x;
        for (int i = 0; i < 4; ++i) {
            const uint64_t z = splitmix(s);
            x.v[2 * i] = (uint32_t)z;
            x.v[2 * i + 1] = (uint32_t)(z >> 32);
        }
        x.v[7] &= (P.bits >= 256) ? 0xffffffffu : ((1u << (P.bits - 224)) - 1);
        Fe dd;
        if (sub8(dd.v, x.v, P.p)) return x;   // borrow: x < p
    }
}
// fully reduce a value < 4p
__device__ Fe reduce_lazy(Fe s, const FieldParams &P) {
    for (int k = 0; k < 3; ++k) {
        Fe dd;
        if (!sub8(dd.v, s.v, P.p)) s = dd;
    }
    return s;
}

// correctness: table[] holds (w_mont, Shoup29(w)) pairs prepared on the host
__global__ void k_check(const Fe *w_mont, const Shoup29 *tab, int n_tab, FieldParams P, unsigned long long *bad) {
    uint64_t seed = 0x5EED0000ull + blockIdx.x * 1024 + threadIdx.x;
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) % n_tab;
    for (int rep = 0; rep < 16; ++rep) {
        Fe x = random_below_p(seed, P);
        if (rep == 1) {                       // the top of the lazy range: 3p - 1 (BN254: < 2^256)
            Fe m1 = P.p[0] ? x : x;
            (void)m1;
            uint32_t c = 0, t3[8];
            for (int i = 0; i < 8; ++i) {
                const uint64_t v = (uint64_t)P.p[i] * 3 + c;
                t3[i] = (uint32_t)v;
                c = (uint32_t)(v >> 32);
            }
            if (c == 0) {
                for (int i = 0; i < 8; ++i) x.v[i] = t3[i];
                x.v[0] -= 1;
            }
        }
        if (rep == 2) x = fe_zero();
        const Fe want = fe_mul(reduce_lazy(x, P), w_mont[t], P);
        const Fe got = reduce_lazy(shoup_mul(x, tab[t], P), P);
        const Fe lazy = shoup_mul(x, tab[t], P);
        // also: the unreduced result must be below 3p (so that a chain stays in range)
        Fe p3;
        uint32_t c = 0;
        for (int i = 0; i < 8; ++i) {
            const uint64_t v = (uint64_t)P.p[i] * 3 + c;
            p3.v[i] = (uint32_t)v;
            c = (uint32_t)(v >> 32);
        }
        Fe dd;
        const bool below3p = c != 0 || sub8(dd.v, lazy.v, p3.v) != 0;
        if (!fe_eq(want, got) || !below3p) atomicAdd(bad, 1ull);
    }
}

__global__ __launch_bounds__(256) void k_time(uint64_t *out, int iters, FieldParams P, const Shoup29 *tab, const uint32_t *tw29, int n_tab, int variant) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) % n_tab;
    Fe a, b;
    for (int i = 0; i < 8; ++i) a.v[i] = 0x1234567u * (i + 1) + threadIdx.x, b.v[i] = 0x7654321u * (i + 3) + blockIdx.x + 7u * threadIdx.x;
    a.v[7] &= 0x0fffffffu;
    b.v[7] &= 0x0fffffffu;
    if (variant == 0) {
        Mul29 c;
        for (int i = 0; i < 9; ++i) c.l[i] = tw29[t * 16 + i];   // a table value: per-lane VGPR operand
        for (int i = 0; i < iters; ++i) {
            a = fe_mul29_t<true>(a, c, P);
            b = fe_mul29_t<true>(b, c, P);
        }
    } else {
        const Shoup29 c = tab[t];
        for (int i = 0; i < iters; ++i) {
            a = shoup_mul(a, c, P);
            b = shoup_mul(b, c, P);
        }
    }
    if (a.v[0] == 0x12345678u && b.v[3] == 0x9abcdef0u) out[0] = a.v[1];
}

// ---- host: w' = floor(w * 2^261 / p) by binary long division on 5 x u64
static void shoup_prepare(const Fe &w_can, const FieldParams &P, Shoup29 &out) {
    unsigned __int128 r[3] = {0, 0, 0};   // remainder < p < 2^256 (two 128-bit words + spill)
    uint64_t rem[5] = {0, 0, 0, 0, 0}, p[5] = {0, 0, 0, 0, 0}, q[5] = {0, 0, 0, 0, 0};
    (void)r;
    for (int i = 0; i < 4; ++i) {
        rem[i] = (uint64_t)w_can.v[2 * i] | ((uint64_t)w_can.v[2 * i + 1] << 32);
        p[i] = (uint64_t)P.p[2 * i] | ((uint64_t)P.p[2 * i + 1] << 32);
    }
    for (int bit = 0; bit < 261; ++bit) {
        for (int i = 4; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);   // rem *= 2
        rem[0] <<= 1;
        for (int i = 4; i > 0; --i) q[i] = (q[i] << 1) | (q[i - 1] >> 63);         // q *= 2
        q[0] <<= 1;
        bool ge = true;
        for (int i = 4; i >= 0; --i)
            if (rem[i] != p[i]) {
                ge = rem[i] > p[i];
                break;
            }
        if (ge) {
            unsigned __int128 borrow = 0;
            for (int i = 0; i < 5; ++i) {
                const unsigned __int128 d = (unsigned __int128)rem[i] - p[i] - borrow;
                rem[i] = (uint64_t)d;
                borrow = (d >> 64) & 1;
            }
            q[0] |= 1;
        }
    }
    auto limb = [](const uint64_t *v, int i) {
        const int b = 29 * i, w = b >> 6, sh = b & 63;
        uint64_t x = v[w] >> sh;
        if (sh > 35 && w + 1 < 5) x |= v[w + 1] << (64 - sh);
        return (uint32_t)(x & ((1u << 29) - 1));
    };
    uint64_t wc[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) wc[i] = (uint64_t)w_can.v[2 * i] | ((uint64_t)w_can.v[2 * i + 1] << 32);
    for (int i = 0; i < 9; ++i) out.w[i] = limb(wc, i), out.wq[i] = limb(q, i);
}

int main(int argc, char **argv) {
    const int field = argc > 1 ? atoi(argv[1]) : 0;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const FieldInfo *fi = field_info(field);
    const FieldParams &P = fi->P;
    // 3p < 2^256 ?
    {
        uint32_t c = 0;
        for (int i = 0; i < 8; ++i) {
            const uint64_t v = (uint64_t)P.p[i] * 3 + c;
            c = (uint32_t)(v >> 32);
        }
        printf("field %d (%u bits): 3p %s 2^256\n", field, P.bits, c ? ">=" : "<");
        if (c) {
            printf("the truncated quotient (result < 3p) does not fit eight words for this field: not applicable\n");
            return 0;
        }
    }
    const int n_tab = 1024;
    std::vector<Fe> w_mont(n_tab);
    std::vector<Shoup29> tab(n_tab);
    std::vector<uint32_t> tw29((size_t)n_tab * 16, 0);
    Fe g = fi->two_adic_root, cur = fe_one(P);
    for (int i = 0; i < n_tab; ++i) {
        w_mont[i] = cur;
        shoup_prepare(fe_to_canonical(cur, P), P, tab[i]);
        const Mul29 m = mul29_prepare(cur, P);
        for (int l = 0; l < 9; ++l) tw29[(size_t)i * 16 + l] = m.l[l];
        cur = fe_mul(cur, g, P);
    }
    Fe *d_w;
    Shoup29 *d_tab;
    uint32_t *d_tw;
    unsigned long long *d_bad, bad = 0;
    uint64_t *d_out;
    CK(hipMalloc(&d_w, n_tab * sizeof(Fe)));
    CK(hipMalloc(&d_tab, n_tab * sizeof(Shoup29)));
    CK(hipMalloc(&d_tw, tw29.size() * 4));
    CK(hipMalloc(&d_bad, 8));
    CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_w, w_mont.data(), n_tab * sizeof(Fe), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, tab.data(), n_tab * sizeof(Shoup29), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw29.data(), tw29.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_bad, 0, 8));
    k_check<<<64, 256>>>(d_w, d_tab, n_tab, P, d_bad);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    printf("check: %llu of %d products differ from fe_mul (or leave [0, 3p))\n", bad, 64 * 256 * 16);
    if (bad) return 1;
    const int blocks = 2048;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round)
        for (int variant = 0; variant < 2; ++variant) {
            k_time<<<blocks, 256>>>(d_out, 8, P, d_tab, d_tw, n_tab, variant);
            CK(hipEventRecord(e0));
            k_time<<<blocks, 256>>>(d_out, iters, P, d_tab, d_tw, n_tab, variant);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double muls = 2.0 * iters * blocks * 256;
            printf("%s: %.3f ms, %.4e multiplications/s\n", variant ? "shoup29 (143 mads, lazy < 3p)      " : "fe_mul29_t<true> (162 mads, lazy < 2p)", ms,
                   muls / (ms * 1e-3));
        }
    return 0;
}
