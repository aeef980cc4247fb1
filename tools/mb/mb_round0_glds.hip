// mb_round0_glds.hip -- round 0 of the two-table product with its prefetch ring in LDS (LDS-DMA, global_load_lds_dwordx4) instead of VGPRs.
//
// k_round0_dot29 (round_kernels.cuh) keeps two pair indices of loads in flight per lane in 64 VGPRs; at 230 VGPRs it runs two waves per SIMD
// and reaches 0.66 of HBM although its issue time (~150 us at 2^24) and its bytes (~170 us at the copy rate) would each fit.  HISTORY.md
// blames the overlap: "the four 32-byte-per-lane load streams overlap imperfectly with two waves per SIMD".  This harness tests that claim
// with the one lever that was never tried: the prefetched rows land in a per-wave LDS ring (no VGPR destination, fully coalesced 1-KiB
// pieces, counted s_waitcnt vmcnt), a lane reads its elements back with ds_read_b128 right before it needs them, and the freed registers
// buy either a third wave per SIMD (ring of 3 half slots) or a deeper ring at two waves (4 half slots = two pair indices, 8 = four).
//
// A unit = one factor's (lo, hi) pair of one 64-pair-index run = 4 KiB = four 1-KiB DMA pieces.  Units are consumed in order
// (run k factor 0, run k factor 1, run k+1 factor 0, ...); unit u + H is issued into the ring position unit u just left.
// Checks: the three sums (S(0), S(1), leading coefficient) equal the shipped kernel's as field elements.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_round0_glds.hip -o tools/mb/bin/mb_round0_glds
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_field.hpp"
#include "round_kernels.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// four 1-KiB pieces of one unit: lo run at `lo`, hi run at `hi` (wave-uniform byte addresses), lane offset voff = 16 * lane
template <bool NT>
__device__ __forceinline__ void dma_unit(uint64_t lo, uint64_t hi, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %4\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2 nt\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                     "s_mov_b32 m0, %5\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3 nt\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lo), "s"(hi), "s"(lds_dst), "s"(lds_dst + 2048u)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %4\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %5\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lo), "s"(hi), "s"(lds_dst), "s"(lds_dst + 2048u)
                     : "memory");
}
// (wait_vm<N>: round_kernels.cuh)
// wait until at most m of the wave's units (4 DMA pieces each) are still in flight
__device__ __forceinline__ void wait_units(int m) {
    switch (m) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<4>(); break;
        case 2: wait_vm<8>(); break;
        case 3: wait_vm<12>(); break;
        case 4: wait_vm<16>(); break;
        case 5: wait_vm<20>(); break;
        case 6: wait_vm<24>(); break;
        default: wait_vm<28>(); break;
    }
}
__device__ __forceinline__ Fe lds_elem(const uint8_t *unit, int h, uint32_t lane) {
    const uint4 *p = reinterpret_cast<const uint4 *>(unit + h * 2048 + lane * 32);
    const uint4 a = p[0], b = p[1];
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}

template <int WPS, int H, bool NT>
__global__ __launch_bounds__(kBlock, WPS) void k_round0_glds(const uint64_t *__restrict__ t0, const uint64_t *__restrict__ t1, uint64_t q,
                                                              FieldParams P, uint64_t *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t ring[];   // [4 waves][H][4096]
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *my = ring + wave * (H * 4096);
    const uint32_t my_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)my);
    const uint32_t voff = lane * 16;
    const uint64_t runs = q >> 6;                                     // q is a multiple of 64
    const uint64_t r0 = (uint64_t)blockIdx.x * 4 + wave, rs = (uint64_t)gridDim.x * 4;
    const uint64_t K = r0 < runs ? (runs - r0 + rs - 1) / rs : 0;     // this wave's runs (wave-uniform)
    const uint64_t base[2] = {(uint64_t)(uintptr_t)t0, (uint64_t)(uintptr_t)t1};
    const uint64_t hi_off = q * 32;
    uint64_t c0[17], c1[17], cL[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) c0[k] = c1[k] = cL[k] = 0;
    Fe sum[3] = {fe_zero(), fe_zero(), fe_zero()};
    constexpr int E = (H + 1) / 2;   // runs in the unrolled epilogue
    // unit u = 2k + f; issue of unit u: run r0 + (u >> 1) * rs of factor u & 1
    auto issue = [&](uint64_t u, uint32_t pos) __attribute__((always_inline)) {
        const uint64_t lo = base[u & 1] + (r0 + (u >> 1) * rs) * 2048;
        dma_unit<NT>(lo, lo + hi_off, voff, my_lds + pos * 4096);
    };
    if (K >= (uint64_t)E) {   // (host guarantees it for every wave; a wave without work skips straight to the reduction)
        const uint64_t U = 2 * K;
#pragma unroll
        for (int i = 0; i < H; ++i) issue((uint64_t)i, (uint32_t)i);
        uint32_t pos = 0;
        uint32_t la0[9], la1[9], ld0[9];
        int since = 0, lazy = 0;
        auto flush = [&]() __attribute__((always_inline)) {
            dot29_normalise(c0);
            dot29_normalise(c1);
            dot29_normalise(cL);
            WideAcc w;
            dot29_to_wide(c0, w);
            sum[0] = fe_add(sum[0], redc_wide(w, P), P);
            dot29_to_wide(c1, w);
            sum[1] = fe_add(sum[1], redc_wide(w, P), P);
            dot29_to_wide(cL, w);
            sum[2] = fe_add(sum[2], redc_wide(w, P), P);
#pragma unroll
            for (int k = 0; k < 17; ++k) c0[k] = c1[k] = cL[k] = 0;
        };
        auto unit_a = [&](const uint8_t *unit) __attribute__((always_inline)) {
            const Fe lo = lds_elem(unit, 0, lane), hi = lds_elem(unit, 1, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            split29(lo.v, la0);
            split29(hi.v, la1);
            const Fe d = fe_sub(hi, lo, P);
            split29(d.v, ld0);
        };
        auto unit_b = [&](const Fe &lo, const Fe &hi) __attribute__((always_inline)) {
            uint32_t b[9];
            split29(lo.v, b);
            dot29_mac(c0, la0, b);
            split29(hi.v, b);
            dot29_mac(c1, la1, b);
            const Fe d = fe_sub(hi, lo, P);
            split29(d.v, b);
            dot29_mac(cL, ld0, b);
            if (++since == 7) {
                dot29_normalise(c0);
                dot29_normalise(c1);
                dot29_normalise(cL);
                since = 0;
            }
            if (++lazy == kMaxLazy) {
                flush();
                lazy = 0;
                since = 0;
            }
        };
        uint64_t u = 0;
        for (; u + 2 * E < U; u += 2) {   // both units of the run issue a successor
            wait_vm<4 * (H - 1)>();
            const uint8_t *ua = my + pos * 4096;
            unit_a(ua);
            issue(u + H, pos);
            pos = pos + 1 == H ? 0 : pos + 1;
            wait_vm<4 * (H - 1)>();
            const uint8_t *ub = my + pos * 4096;
            const Fe lo = lds_elem(ub, 0, lane), hi = lds_elem(ub, 1, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(u + 1 + H, pos);
            pos = pos + 1 == H ? 0 : pos + 1;
            unit_b(lo, hi);
        }
        // the last E runs: unit with e units after it issues iff e >= H and waits for min(H - 1, e) younger units
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int ea = 2 * (E - i) - 1, eb = ea - 1;
            wait_units(ea < H - 1 ? ea : H - 1);
            const uint8_t *ua = my + pos * 4096;
            unit_a(ua);
            if (ea >= H) issue(u + H, pos);
            pos = pos + 1 == H ? 0 : pos + 1;
            wait_units(eb < H - 1 ? eb : H - 1);
            const uint8_t *ub = my + pos * 4096;
            const Fe lo = lds_elem(ub, 0, lane), hi = lds_elem(ub, 1, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (eb >= H) issue(u + 1 + H, pos);
            pos = pos + 1 == H ? 0 : pos + 1;
            unit_b(lo, hi);
            u += 2;
        }
        flush();
    }
    block_reduce_store<3>(sum, partials, P);
}

// the shipped loop (two pair indices of register prefetch) with nontemporal loads: separates the cache policy from the LDS ring
__device__ __forceinline__ Fe fe_load_nt(const uint64_t *base, uint64_t idx) {
    const u32x4_t *p = reinterpret_cast<const u32x4_t *>(base + 4 * idx);
    const u32x4_t a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1);
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
__global__ __launch_bounds__(kBlock, 2) void k_round0_regs_nt(const uint64_t *__restrict__ t0, const uint64_t *__restrict__ t1, uint64_t q, FieldParams P,
                                                               uint64_t *__restrict__ partials) {
    uint64_t c0[17], c1[17], cL[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) c0[k] = c1[k] = cL[k] = 0;
    const uint64_t *in[2] = {t0, t1};
    Fe cur[2][2], nxt[2][2];
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < q) {
#pragma unroll
        for (int f = 0; f < 2; ++f) cur[f][0] = fe_load_nt(in[f], j), cur[f][1] = fe_load_nt(in[f], j + q);
    }
    if (j + stride < q) {
#pragma unroll
        for (int f = 0; f < 2; ++f) nxt[f][0] = fe_load_nt(in[f], j + stride), nxt[f][1] = fe_load_nt(in[f], j + stride + q);
    }
    int since = 0;
    while (j < q) {
        const uint64_t jn = j + stride, jnn = jn + stride;
        Fe v[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) v[f][0] = cur[f][0], v[f][1] = cur[f][1];
#pragma unroll
        for (int f = 0; f < 2; ++f) cur[f][0] = nxt[f][0], cur[f][1] = nxt[f][1];
        if (jnn < q) {
#pragma unroll
            for (int f = 0; f < 2; ++f) nxt[f][0] = fe_load_nt(in[f], jnn), nxt[f][1] = fe_load_nt(in[f], jnn + q);
        }
        uint32_t a[9], b[9];
        split29(v[0][0].v, a);
        split29(v[1][0].v, b);
        dot29_mac(c0, a, b);
        split29(v[0][1].v, a);
        split29(v[1][1].v, b);
        dot29_mac(c1, a, b);
        const Fe d0 = fe_sub(v[0][1], v[0][0], P), d1 = fe_sub(v[1][1], v[1][0], P);
        split29(d0.v, a);
        split29(d1.v, b);
        dot29_mac(cL, a, b);
        if (++since == 7) {
            dot29_normalise(c0);
            dot29_normalise(c1);
            dot29_normalise(cL);
            since = 0;
        }
        j = jn;
    }
    dot29_normalise(c0);
    dot29_normalise(c1);
    dot29_normalise(cL);
    Fe sum[3];
    WideAcc w;
    dot29_to_wide(c0, w);
    sum[0] = redc_wide(w, P);
    dot29_to_wide(c1, w);
    sum[1] = redc_wide(w, P);
    dot29_to_wide(cL, w);
    sum[2] = redc_wide(w, P);
    block_reduce_store<3>(sum, partials, P);
}

__global__ void k_fill(uint64_t *t, uint64_t n, uint64_t seed, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t s = seed + i * 0x9E3779B97F4A7C15ull;
        Fe x;
        for (int w = 0; w < 4; ++w) {
            s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
            x.v[2 * w] = (uint32_t)s;
            x.v[2 * w + 1] = (uint32_t)(s >> 32);
        }
        x.v[7] &= 0x0fffffffu;   // < 2^252 < p: a valid (Montgomery-form) element
        fe_store(t, i, x);
    }
}
static Fe host_sum(const std::vector<uint64_t> &part, size_t blocks, size_t per, size_t slot, const FieldParams &P) {
    Fe s = fe_zero();
    for (size_t b = 0; b < blocks; ++b) s = fe_add(s, fe_from_u64limbs(part.data() + (b * per + slot) * 4), P);
    return s;
}

struct Variant {
    const char *name;
    void (*launch)(uint32_t grid, const uint64_t *, const uint64_t *, uint64_t, const FieldParams &, uint64_t *);
    int wps, h;
};
template <int WPS, int H, bool NT>
static void go(uint32_t grid, const uint64_t *a, const uint64_t *b, uint64_t q, const FieldParams &P, uint64_t *part) {
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute((const void *)k_round0_glds<WPS, H, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * H * 4096));
        once = true;
    }
    k_round0_glds<WPS, H, NT><<<grid, kBlock, 4 * H * 4096>>>(a, b, q, P, part);
}

int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 20;
    const uint64_t n = 1ull << log_n, q = n >> 1;
    const FieldInfo *fi = field_info(0);
    const FieldParams &P = fi->P;
    uint64_t *T[2], *part;
    for (int f = 0; f < 2; ++f) {
        CK(hipMalloc(&T[f], n * 32));
        k_fill<<<2048, 256>>>(T[f], n, 0x5EED + 77 * f, P);
    }
    CK(hipMalloc(&part, 4096 * 3 * 32));
    CK(hipDeviceSynchronize());
    uint32_t g0 = (uint32_t)((q + 256ull * kMaxLazy - 1) / (256ull * kMaxLazy));
    if (g0 < 512) g0 = 512;
    auto shipped = [&]() {
        FactorPtrs fp = {};
        fp.in[0] = T[0], fp.in[1] = T[1];
        k_round0_dot29<0><<<g0, kBlock>>>(fp, q, P, part);
    };
    shipped();
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> pc((size_t)g0 * 3 * 4);
    CK(hipMemcpy(pc.data(), part, pc.size() * 8, hipMemcpyDeviceToHost));
    Fe want[3];
    for (int t = 0; t < 3; ++t) want[t] = host_sum(pc, g0, 3, t, P);

    const Variant vars[] = {
        {"2 waves/SIMD, ring 2 units (1 pair index)", go<2, 2, false>, 2, 2},
        {"2 waves/SIMD, ring 4 units (2 pair indices)", go<2, 4, false>, 2, 4},
        {"2 waves/SIMD, ring 4 units, nt", go<2, 4, true>, 2, 4},
        {"2 waves/SIMD, ring 2 units, nt", go<2, 2, true>, 2, 2},
        {"2 waves/SIMD, ring 3 units, nt", go<2, 3, true>, 2, 3},
        {"3 waves/SIMD, ring 3 units (1.5 pair indices)", go<3, 3, false>, 3, 3},
        {"3 waves/SIMD, ring 3 units, nt", go<3, 3, true>, 3, 3},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto &&f) {
        f();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.0 / reps;
    };
    const double bytes = 2.0 * n * 32;
    for (int round = 0; round < 2; ++round) {
        const double us = time_it(shipped);
        printf("shipped k_round0_dot29<0>  grid %4u                         : %7.1f us  %5.2f TB/s  %.3f of 8 TB/s\n", g0, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8);
        {
            if (round == 0) {
                CK(hipMemset(part, 0, 4096 * 3 * 32));
                k_round0_regs_nt<<<g0, kBlock>>>(T[0], T[1], q, P, part);
                CK(hipDeviceSynchronize());
                std::vector<uint64_t> pv((size_t)g0 * 3 * 4);
                CK(hipMemcpy(pv.data(), part, pv.size() * 8, hipMemcpyDeviceToHost));
                for (int t = 0; t < 3; ++t)
                    if (!fe_eq(host_sum(pv, g0, 3, t, P), want[t])) {
                        printf("register prefetch + nt: sums DIFFER\n");
                        return 1;
                    }
            }
            const double u3 = time_it([&]() { k_round0_regs_nt<<<g0, kBlock>>>(T[0], T[1], q, P, part); });
            printf("same loop, nontemporal register loads  grid %4u             : %7.1f us  %5.2f TB/s  %.3f\n", g0, u3, bytes / u3 * 1e-6, bytes / u3 * 1e-6 / 8);
        }
        for (const Variant &v : vars) {
            const uint32_t grids[] = {256u * v.wps, 512u * v.wps, 1024u, 2048u};
            for (uint32_t g : grids) {
                const uint64_t waves = (uint64_t)g * 4, runs = q >> 6;
                const uint64_t per = runs / waves;
                if (runs % waves || per < (uint64_t)(v.h + 1) / 2 + 1) continue;   // every wave the same number of runs, epilogue fits
                if (round == 0) {
                    CK(hipMemset(part, 0, 4096 * 3 * 32));
                    v.launch(g, T[0], T[1], q, P, part);
                    CK(hipDeviceSynchronize());
                    std::vector<uint64_t> pv((size_t)g * 3 * 4);
                    CK(hipMemcpy(pv.data(), part, pv.size() * 8, hipMemcpyDeviceToHost));
                    bool ok = true;
                    for (int t = 0; t < 3; ++t) ok = ok && fe_eq(host_sum(pv, g, 3, t, P), want[t]);
                    if (!ok) {
                        printf("%s grid %u: sums DIFFER\n", v.name, g);
                        return 1;
                    }
                }
                const double u2 = time_it([&]() { v.launch(g, T[0], T[1], q, P, part); });
                printf("  %-48s grid %4u: %7.1f us  %5.2f TB/s  %.3f\n", v.name, g, u2, bytes / u2 * 1e-6, bytes / u2 * 1e-6 / 8);
            }
        }
    }
    printf("all variants: S(0), S(1) and the leading coefficient equal the shipped kernel's\n");
    return 0;
}
