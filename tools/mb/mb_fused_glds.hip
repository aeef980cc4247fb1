// mb_fused_glds.hip -- the big fused round (k_round_kd<2,2,FUSED,SKIP1,LEAD>) with its input rows arriving by LDS-DMA and its half tables
// leaving as fully coalesced 1-KiB stores.
//
// mb_round0_glds.hip showed what the sums-only round gains from global_load_lds_dwordx4 (8 % at 2^24: the gain is the access shape -- whole
// 1-KiB nontemporal pieces instead of 32 bytes per lane -- not the prefetch depth).  The fused round moves 1.5x the bytes of round 0 per pair
// index and is 58 % of the n = 24 prover's kernel time; the shipped kernel reads AND writes 32 bytes per lane (fe_load / fe_store).  Here:
//   * a unit = one factor's four rows (j, j + q, j + 2q, j + 3q) of a 64-pair-index run = 8 KiB = eight 1-KiB DMA pieces into a two-unit
//     ring per wave (16 KiB; 64 KiB per workgroup, two workgroups per CU);
//   * lane l reads element pair_owned(l) of each row back from LDS, so the two folded elements leave through pair_scatter
//     (one DPP half swap) as two coalesced 1-KiB stores per output row -- the fold kernel's data movement (kernels.cuh k_fold_msb);
//   * stores count in vmcnt on this part and retire in order with the DMA pieces: every wait below is counted by hand.
// Checks: half tables bit-identical to the shipped kernel's, S(0) and the leading coefficient equal as field elements.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_fused_glds.hip -o tools/mb/bin/mb_fused_glds
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_field.hpp"
#include "round_kernels.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// eight 1-KiB pieces: rows at a0..a3 (wave-uniform byte addresses), lane offset voff = 16 * lane, LDS rows 2 KiB apart from lds_dst
template <bool NT>
__device__ __forceinline__ void dma_rows4(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
#define ZK_DMA4(POL)                                                                                                              \
    asm volatile("s_mov_b32 %0, m0\n\t"                                                                                          \
                 "s_mov_b32 m0, %6\n\t"                                                                                          \
                 "s_nop 0\n\t"                                                                                                   \
                 "global_load_lds_dwordx4 %1, %2" POL "\n\t"                                                                     \
                 "global_load_lds_dwordx4 %1, %2 offset:1024" POL "\n\t"                                                         \
                 "s_mov_b32 m0, %7\n\t"                                                                                          \
                 "s_nop 0\n\t"                                                                                                   \
                 "global_load_lds_dwordx4 %1, %3" POL "\n\t"                                                                     \
                 "global_load_lds_dwordx4 %1, %3 offset:1024" POL "\n\t"                                                         \
                 "s_mov_b32 m0, %8\n\t"                                                                                          \
                 "s_nop 0\n\t"                                                                                                   \
                 "global_load_lds_dwordx4 %1, %4" POL "\n\t"                                                                     \
                 "global_load_lds_dwordx4 %1, %4 offset:1024" POL "\n\t"                                                         \
                 "s_mov_b32 m0, %9\n\t"                                                                                          \
                 "s_nop 0\n\t"                                                                                                   \
                 "global_load_lds_dwordx4 %1, %5" POL "\n\t"                                                                     \
                 "global_load_lds_dwordx4 %1, %5 offset:1024" POL "\n\t"                                                         \
                 "s_mov_b32 m0, %0"                                                                                              \
                 : "=&s"(keep)                                                                                                   \
                 : "v"(voff), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(lds_dst), "s"(lds_dst + 2048u), "s"(lds_dst + 4096u),     \
                   "s"(lds_dst + 6144u)                                                                                          \
                 : "memory")
    if (NT) ZK_DMA4(" nt");
    else ZK_DMA4("");
#undef ZK_DMA4
}
// (wait_vm<N>: round_kernels.cuh)
__device__ __forceinline__ Fe lds_elem(const uint8_t *row, uint32_t elem) {
    const uint4 *p = reinterpret_cast<const uint4 *>(row + elem * 32);
    const uint4 a = p[0], b = p[1];
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}

// q pair indices (a multiple of 64); tables of 4q elements in, 2q out (out may equal in: a wave reads its runs before it writes them
// and no other wave touches them).  partials[block][3]: slot 0 = S(0), slot 2 = the leading coefficient (slot 1 is not written)
template <bool NT, bool STORE_NT>
__global__ __launch_bounds__(kBlock, 2) void k_fused22_glds(const uint64_t *__restrict__ t0, const uint64_t *__restrict__ t1, uint64_t *o0, uint64_t *o1,
                                                             uint64_t q, FieldParams P, const uint64_t *__restrict__ rptr,
                                                             uint64_t *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t ring[];   // [4 waves][2 units][4 rows][2048]
    const Mul29 r = load_challenge29(rptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the challenge loads: nothing of the compiler's is in flight when the DMA counting starts)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *my = ring + wave * 16384;
    const uint32_t my_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)my);
    const uint32_t voff = lane * 16, own = pair_owned(lane);
    const uint64_t runs = q >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * 4 + wave, rs = (uint64_t)gridDim.x * 4;
    const uint64_t K = r0 < runs ? (runs - r0 + rs - 1) / rs : 0;   // this wave's runs (wave-uniform)
    const uint64_t in0 = (uint64_t)(uintptr_t)t0, in1 = (uint64_t)(uintptr_t)t1, rowb = q * 32;
    WideAcc acc0, accL;
    wide_zero(acc0);
    wide_zero(accL);
    Fe sum[3] = {fe_zero(), fe_zero(), fe_zero()};
    auto issue = [&](uint64_t in, uint64_t run, uint32_t pos) __attribute__((always_inline)) {
        const uint64_t a = in + run * 2048;
        dma_rows4<NT>(a, a + rowb, a + 2 * rowb, a + 3 * rowb, voff, my_lds + pos * 8192);
    };
    if (K) {
        issue(in0, r0, 0);
        issue(in1, r0, 1);
        wait_vm<8>();   // unit (run 0, factor 0) has landed; factor 1's eight pieces may still fly
        for (uint64_t k = 0; k < K; ++k) {
            const uint64_t run = r0 + k * rs;
            const bool last = k + 1 == K;
            Fe p0, pL;
            // ---- factor 0 (ring position 0)
            {
                const Fe c0 = lds_elem(my, own), c1 = lds_elem(my + 2048, own), c2 = lds_elem(my + 4096, own), c3 = lds_elem(my + 6144, own);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) issue(in0, run + rs, 0);
                const Fe lo = fe_sub(c0, fe_mul29(fe_sub(c0, c2, P), r, P), P);
                const Fe hi = fe_sub(c1, fe_mul29(fe_sub(c1, c3, P), r, P), P);
                if (STORE_NT) {
                    run_store_nt(o0 + run * 256, lane, lo);
                    run_store_nt(o0 + (run * 64 + q) * 4, lane, hi);
                } else {
                    run_store(o0 + run * 256, lane, lo);
                    run_store(o0 + (run * 64 + q) * 4, lane, hi);
                }
                p0 = lo;
                pL = fe_sub(hi, lo, P);
            }
            // factor 1 of this run has landed when at most [next factor-0 unit: 8] + [the four stores above] are in flight
            if (!last) wait_vm<12>();
            else wait_vm<4>();
            // ---- factor 1 (ring position 1)
            {
                const uint8_t *u1 = my + 8192;
                const Fe c0 = lds_elem(u1, own), c1 = lds_elem(u1 + 2048, own), c2 = lds_elem(u1 + 4096, own), c3 = lds_elem(u1 + 6144, own);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) issue(in1, run + rs, 1);
                const Fe lo = fe_sub(c0, fe_mul29(fe_sub(c0, c2, P), r, P), P);
                const Fe hi = fe_sub(c1, fe_mul29(fe_sub(c1, c3, P), r, P), P);
                if (STORE_NT) {
                    run_store_nt(o1 + run * 256, lane, lo);
                    run_store_nt(o1 + (run * 64 + q) * 4, lane, hi);
                } else {
                    run_store(o1 + run * 256, lane, lo);
                    run_store(o1 + (run * 64 + q) * 4, lane, hi);
                }
                wide_mac(acc0, p0.v, lo.v);
                const Fe d = fe_sub(hi, lo, P);
                wide_mac(accL, pL.v, d.v);
            }
            // the next run's factor 0 has landed when at most [its factor 1: 8] + [this run's eight stores, four of them older than the
            // factor-1 pieces: counted as if all were younger -- waits for no more than needed] are in flight
            if (!last) wait_vm<16>();
        }
        sum[0] = redc_wide(acc0, P);
        sum[2] = redc_wide(accL, P);
    }
    block_reduce_store<3, true>(sum, partials, P);
}

// The same round with ONE unit of ring per wave (8 KiB; 32 KiB per workgroup): the next unit's DMA is issued the moment this unit's rows
// are in registers.  STORE: 0 = pair_scatter + cached 1-KiB stores, 1 = the same nontemporal, 2 = per-lane fe_store (32 bytes per lane,
// no DPP / select instructions), 3 = per-lane nontemporal.  DOT29: the two products on carry-free 29-bit columns (dot29_mac) instead
// of wide_mac.
template <int STORE>
__device__ __forceinline__ void put_run(uint64_t *run, uint32_t lane, const Fe &e) {
    if (STORE == 0) run_store(run, lane, e);
    else if (STORE == 1) run_store_nt(run, lane, e);
    else if (STORE == 2) fe_store(run, lane, e);
    else {
        u32x4_t *q = reinterpret_cast<u32x4_t *>(run + 4 * (uint64_t)lane);
        const u32x4_t lo = {e.v[0], e.v[1], e.v[2], e.v[3]}, hi = {e.v[4], e.v[5], e.v[6], e.v[7]};
        __builtin_nontemporal_store(lo, q);
        __builtin_nontemporal_store(hi, q + 1);
    }
}
template <int WPS, bool NT, int STORE, bool DOT29, int MATH = 2>
__global__ __launch_bounds__(kBlock, WPS) void k_fused22_glds1(const uint64_t *__restrict__ t0, const uint64_t *__restrict__ t1, uint64_t *o0, uint64_t *o1,
                                                                uint64_t q, FieldParams P, const uint64_t *__restrict__ rptr,
                                                                uint64_t *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t ring[];   // [4 waves][4 rows][2048]
    const Mul29 r = load_challenge29(rptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *my = ring + wave * 8192;
    const uint32_t my_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)my);
    const uint32_t voff = lane * 16, own = STORE < 2 ? pair_owned(lane) : lane;
    const uint64_t runs = q >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * 4 + wave, rs = (uint64_t)gridDim.x * 4;
    const uint64_t K = r0 < runs ? (runs - r0 + rs - 1) / rs : 0;
    const uint64_t in0 = (uint64_t)(uintptr_t)t0, in1 = (uint64_t)(uintptr_t)t1, rowb = q * 32;
    WideAcc acc0, accL;
    uint64_t d0[17], dL[17];
    wide_zero(acc0);
    wide_zero(accL);
#pragma unroll
    for (int i = 0; i < 17; ++i) d0[i] = dL[i] = 0;
    Fe sum[3] = {fe_zero(), fe_zero(), fe_zero()};
    auto issue = [&](uint64_t in, uint64_t run) __attribute__((always_inline)) {
        const uint64_t a = in + run * 2048;
        dma_rows4<NT>(a, a + rowb, a + 2 * rowb, a + 3 * rowb, voff, my_lds);
    };
    if (K) {
        issue(in0, r0);
        wait_vm<0>();
        int since = 0;
        for (uint64_t k = 0; k < K; ++k) {
            const uint64_t run = r0 + k * rs;
            const bool last = k + 1 == K;
            Fe p0, pL;
            uint32_t l0[9], lL[9];
            {
                const Fe c0 = lds_elem(my, own), c1 = lds_elem(my + 2048, own), c2 = lds_elem(my + 4096, own), c3 = lds_elem(my + 6144, own);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue(in1, run);
                Fe lo, hi;
                if (MATH >= 1) {
                    lo = fe_sub(c0, fe_mul29(fe_sub(c0, c2, P), r, P), P);
                    hi = fe_sub(c1, fe_mul29(fe_sub(c1, c3, P), r, P), P);
                } else {   // data movement only: the same rows in, the same rows out
#pragma unroll
                    for (int i = 0; i < 8; ++i) lo.v[i] = c0.v[i] ^ c2.v[i], hi.v[i] = c1.v[i] ^ c3.v[i];
                }
                put_run<STORE>(o0 + run * 256, lane, lo);
                put_run<STORE>(o0 + (run * 64 + q) * 4, lane, hi);
                const Fe d = MATH >= 2 ? fe_sub(hi, lo, P) : hi;
                if (DOT29) {
                    split29(lo.v, l0);
                    split29(d.v, lL);
                } else {
                    p0 = lo;
                    pL = d;
                }
            }
            wait_vm<4>();   // the eight pieces are older than the four stores
            {
                const Fe c0 = lds_elem(my, own), c1 = lds_elem(my + 2048, own), c2 = lds_elem(my + 4096, own), c3 = lds_elem(my + 6144, own);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) issue(in0, run + rs);
                Fe lo, hi;
                if (MATH >= 1) {
                    lo = fe_sub(c0, fe_mul29(fe_sub(c0, c2, P), r, P), P);
                    hi = fe_sub(c1, fe_mul29(fe_sub(c1, c3, P), r, P), P);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) lo.v[i] = c0.v[i] ^ c2.v[i], hi.v[i] = c1.v[i] ^ c3.v[i];
                }
                put_run<STORE>(o1 + run * 256, lane, lo);
                put_run<STORE>(o1 + (run * 64 + q) * 4, lane, hi);
                const Fe d = MATH >= 2 ? fe_sub(hi, lo, P) : hi;
                if (MATH < 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc0.v[i] ^= lo.v[i] ^ p0.v[i], accL.v[i] ^= d.v[i] ^ pL.v[i];
                } else if (DOT29) {
                    uint32_t b[9];
                    split29(lo.v, b);
                    dot29_mac(d0, l0, b);
                    split29(d.v, b);
                    dot29_mac(dL, lL, b);
                    if (++since == 7) {
                        dot29_normalise(d0);
                        dot29_normalise(dL);
                        since = 0;
                    }
                } else {
                    wide_mac(acc0, p0.v, lo.v);
                    wide_mac(accL, pL.v, d.v);
                }
            }
            if (!last) wait_vm<4>();
        }
        if (DOT29) {
            dot29_normalise(d0);
            dot29_normalise(dL);
            dot29_to_wide(d0, acc0);
            dot29_to_wide(dL, accL);
        }
        sum[0] = redc_wide(acc0, P);
        sum[2] = redc_wide(accL, P);
    }
    block_reduce_store<3, true>(sum, partials, P);
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

template <bool NT, bool SNT>
static void go(uint32_t grid, const uint64_t *a, const uint64_t *b, uint64_t *oa, uint64_t *ob, uint64_t q, const FieldParams &P, const uint64_t *chal,
               uint64_t *part) {
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute((const void *)k_fused22_glds<NT, SNT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        once = true;
    }
    k_fused22_glds<NT, SNT><<<grid, kBlock, 65536>>>(a, b, oa, ob, q, P, chal, part);
}

template <int WPS, bool NT, int SNT, bool DOT29 = false, int MATH = 2>
static void go1(uint32_t grid, const uint64_t *a, const uint64_t *b, uint64_t *oa, uint64_t *ob, uint64_t q, const FieldParams &P, const uint64_t *chal,
                uint64_t *part) {
    k_fused22_glds1<WPS, NT, SNT, DOT29, MATH><<<grid, kBlock, 32768>>>(a, b, oa, ob, q, P, chal, part);
}

int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 20;
    const uint64_t n = 1ull << log_n, q = n >> 2;
    const FieldInfo *fi = field_info(0);
    const FieldParams &P = fi->P;
    uint64_t *T[2], *H[2], *O[2], *part, *chal;
    for (int f = 0; f < 2; ++f) {
        CK(hipMalloc(&T[f], n * 32));
        CK(hipMalloc(&H[f], n * 16));
        CK(hipMalloc(&O[f], n * 16));
        k_fill<<<2048, 256>>>(T[f], n, 0x5EED + 77 * f, P);
    }
    CK(hipMalloc(&part, 4096 * 3 * 32));
    CK(hipMalloc(&chal, 128));
    const Fe r_fe = fe_pow_u64(fi->two_adic_root, 12345, P);
    const Mul29 rm = mul29_prepare(r_fe, P);
    uint32_t rec[32] = {};
    for (int i = 0; i < 9; ++i) rec[8 + i] = rm.l[i];
    CK(hipMemcpy(chal, rec, sizeof rec, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    uint32_t g0 = (uint32_t)((q + 256ull * kMaxLazy - 1) / (256ull * kMaxLazy));
    if (g0 < 512) g0 = 512;
    auto shipped = [&]() {
        FactorPtrs fp = {};
        fp.in[0] = T[0], fp.in[1] = T[1], fp.out[0] = H[0], fp.out[1] = H[1];
        k_round_kd<2, 2, true, 0, true, true><<<g0, kBlock>>>(fp, q, P, chal, part, ClaimJob{});
    };
    CK(hipMemset(part, 0, 4096 * 3 * 32));
    shipped();
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> pc((size_t)g0 * 3 * 4);
    CK(hipMemcpy(pc.data(), part, pc.size() * 8, hipMemcpyDeviceToHost));
    const Fe want0 = host_sum(pc, g0, 3, 0, P), wantL = host_sum(pc, g0, 3, 2, P);
    std::vector<uint64_t> ha((size_t)(n >> 1) * 4), hb((size_t)(n >> 1) * 4);

    struct Variant {
        const char *name;
        void (*launch)(uint32_t, const uint64_t *, const uint64_t *, uint64_t *, uint64_t *, uint64_t, const FieldParams &, const uint64_t *, uint64_t *);
    };
    const Variant vars[] = {
        {"LDS-DMA rows nt, coalesced stores nt", go<true, true>},
        {"ring 1, scatter nt", go1<2, true, 1>},
        {"ring 1, scatter cached", go1<2, true, 0>},
        {"ring 1, per-lane cached", go1<2, true, 2>},
        {"ring 1, per-lane nt", go1<2, true, 3>},
        {"ring 1, scatter nt, folds only (no products)", go1<2, true, 1, false, 1>},
        {"ring 1, scatter nt, data movement only", go1<2, true, 1, false, 0>},
        {"ring 1, scatter nt, 4 waves, data movement only", go1<4, true, 1, false, 0>},
        {"ring 1, scatter nt, dot29", go1<2, true, 1, true>},
        {"ring 1, per-lane cached, dot29", go1<2, true, 2, true>},
        {"ring 1, per-lane nt, dot29", go1<2, true, 3, true>},
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
    const double bytes = 2.0 * n * 32 * 1.5;
    for (int round = 0; round < 2; ++round) {
        const double us = time_it(shipped);
        printf("shipped k_round_kd<2,2,fused,SKIP1,LEAD>  grid %4u      : %7.1f us  %5.2f TB/s  %.3f of 8 TB/s\n", g0, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8);
        for (const Variant &v : vars) {
            const uint32_t grids[] = {512u, 1024u, 2048u};
            for (uint32_t g : grids) {
                if ((q >> 6) < (uint64_t)g * 4) continue;
                if ((q + (uint64_t)g * 256 - 1) / ((uint64_t)g * 256) > (uint64_t)kMaxLazy) continue;
                if (round == 0) {
                    CK(hipMemset(part, 0, 4096 * 3 * 32));
                    for (int f = 0; f < 2; ++f) CK(hipMemset(O[f], 0, n * 16));
                    v.launch(g, T[0], T[1], O[0], O[1], q, P, chal, part);
                    CK(hipDeviceSynchronize());
                    std::vector<uint64_t> pv((size_t)g * 3 * 4);
                    CK(hipMemcpy(pv.data(), part, pv.size() * 8, hipMemcpyDeviceToHost));
                    bool ok = fe_eq(host_sum(pv, g, 3, 0, P), want0) && fe_eq(host_sum(pv, g, 3, 2, P), wantL);
                    for (int f = 0; f < 2; ++f) {
                        CK(hipMemcpy(ha.data(), H[f], ha.size() * 8, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(hb.data(), O[f], hb.size() * 8, hipMemcpyDeviceToHost));
                        ok = ok && memcmp(ha.data(), hb.data(), ha.size() * 8) == 0;
                    }
                    if (!ok && !strstr(v.name, "only")) {
                        printf("%s grid %u: DIFFERS from the shipped kernel\n", v.name, g);
                        return 1;
                    }
                }
                const double u2 = time_it([&]() { v.launch(g, T[0], T[1], O[0], O[1], q, P, chal, part); });
                printf("  %-40s grid %4u: %7.1f us  %5.2f TB/s  %.3f\n", v.name, g, u2, bytes / u2 * 1e-6, bytes / u2 * 1e-6 / 8);
            }
        }
    }
    printf("all variants: half tables bit-identical, S(0) and the leading coefficient equal the shipped kernel's\n");
    // ---- the same kernels IN PLACE (out = in, as the prover runs them; the tables change from launch to launch: timing only)
    for (int round = 0; round < 3; ++round) {
        const double us = time_it([&]() {
            FactorPtrs fp = {};
            fp.in[0] = T[0], fp.in[1] = T[1], fp.out[0] = T[0], fp.out[1] = T[1];
            k_round_kd<2, 2, true, 0, true, true><<<g0, kBlock>>>(fp, q, P, chal, part, ClaimJob{});
        });
        printf("in place: shipped                                  : %7.1f us  %.3f\n", us, bytes / us * 1e-6 / 8);
        for (const Variant &v : vars) {
            const double u2 = time_it([&]() { v.launch(512, T[0], T[1], T[0], T[1], q, P, chal, part); });
            printf("in place: %-40s : %7.1f us  %.3f\n", v.name, u2, bytes / u2 * 1e-6 / 8);
        }
    }
    return 0;
}
