// mb_rows_pattern.hip -- what the fused round's ACCESS PATTERN alone costs, and whether its power-of-two row strides are to blame.
//
// A fused sumcheck round over two tables reads eight streams (rows j, j + q, j + 2q, j + 3q of each table: q * 32 bytes apart, a power
// of two) and writes four (rows 0 and 1 of each, in place).  With every multiplication removed (mb_fused_glds.hip, "data movement only")
// the kernel still takes 285-292 us at 2^24 = 0.69-0.71 of 8 TB/s, against 0.80 for the fold's three streams.  This harness moves the same
// bytes with the same per-wave structure (LDS-DMA ring of one 8-KiB unit, coalesced nontemporal stores) and varies only the addresses:
// the distance between a table's rows (q * 32 + pad) and between the two tables.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_rows_pattern.hip -o tools/mb/bin/mb_rows_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "host_field.hpp"
#include "round_kernels.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// store cache policies: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1, 6 sc0
template <int POL>
__device__ __forceinline__ void store16(uint4 v, uint4 *p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 w = {v.x, v.y, v.z, v.w};
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(w) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(w) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(w) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(w) : "memory");
}
// NR rows read per table and run, NW rows written (NW <= NR); row r of a table at base + r * row_stride
template <int NR, int NW, int POL = 1>
__global__ __launch_bounds__(kBlock, 2) void k_move(uint64_t t0, uint64_t t1, uint64_t o0, uint64_t o1, uint64_t q, uint64_t row_stride) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *my = glds_ring + wave * 8192;
    const uint32_t my_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)my);
    const uint32_t voff = lane * 16;
    const uint64_t runs = q >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * 4 + wave, rs = (uint64_t)gridDim.x * 4;
    const uint64_t K = r0 < runs ? (runs - r0 + rs - 1) / rs : 0;
    if (!K) return;
    auto issue = [&](uint64_t in, uint64_t run) __attribute__((always_inline)) {
        const uint64_t a = in + run * 2048;
        if (NR == 4) glds_rows4(a, a + row_stride, a + 2 * row_stride, a + 3 * row_stride, voff, my_lds);
        else glds_rows2(a, a + row_stride, voff, my_lds);
    };
    auto unit = [&](uint64_t out, uint64_t run, bool more, uint64_t next_in, uint64_t next_run) __attribute__((always_inline)) {
        uint4 v[NR][2];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint4 *p = reinterpret_cast<const uint4 *>(my + r * 2048) + lane;
            v[r][0] = p[0];
            v[r][1] = p[64];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) issue(next_in, next_run);
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            uint4 a = v[r][0], b = v[r][1];
            if (NR > NW) {   // fold the other rows in so that every load is used
                a.x ^= v[r + NW][0].x, a.y ^= v[r + NW][0].y, a.z ^= v[r + NW][0].z, a.w ^= v[r + NW][0].w;
                b.x ^= v[r + NW][1].x, b.y ^= v[r + NW][1].y, b.z ^= v[r + NW][1].z, b.w ^= v[r + NW][1].w;
            }
            uint4 *o = reinterpret_cast<uint4 *>(out + r * row_stride + run * 2048) + lane;
            store16<POL>(a, o);
            store16<POL>(b, o + 64);
        }
    };
    issue(t0, r0);
    wait_vm<0>();
    for (uint64_t k = 0; k < K; ++k) {
        const uint64_t run = r0 + k * rs;
        const bool last = k + 1 == K;
        unit(o0, run, true, t1, run);
        wait_vm<2 * NW>();
        unit(o1, run, !last, t0, run + rs);
        if (!last) wait_vm<2 * NW>();
    }
}

int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 40;
    const uint64_t n = 1ull << log_n, q = n >> 2;
    uint8_t *buf;
    const uint64_t slack = 64ull << 20;
    CK(hipMalloc(&buf, 4 * n * 32 + 8 * slack));   // two input tables, two output tables, room for the pads
    CK(hipMemset(buf, 1, 4 * n * 32 + 8 * slack));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto &&f) {
        for (int i = 0; i < 5; ++i) f();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.0 / reps;
    };
    const uint64_t base = (uint64_t)(uintptr_t)buf;
    const uint64_t stride = q * 32, tsize = 4 * stride;
    const uint64_t t0 = base, t1 = base + tsize, o0 = base + 2 * tsize + (1 << 20), o1 = o0 + tsize;
    for (int round = 0; round < 3; ++round) {
        const double rd = 2.0 * 4 * q * 32, wr2 = 2.0 * 2 * q * 32, wr1 = 2.0 * q * 32;
        double us = time_it([&]() { k_move<4, 0><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
        printf("8 streams read, none written          : %6.1f us  %.3f of 8 TB/s\n", us, rd / us * 1e-6 / 8);
        us = time_it([&]() { k_move<4, 1><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
        printf("8 streams read, 2 written (out of place): %6.1f us  %.3f\n", us, (rd + wr1) / us * 1e-6 / 8);
        us = time_it([&]() { k_move<4, 2><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
        printf("8 streams read, 4 written (out of place): %6.1f us  %.3f\n", us, (rd + wr2) / us * 1e-6 / 8);
        us = time_it([&]() { k_move<4, 2><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
        printf("8 streams read, 4 written (in place)    : %6.1f us  %.3f\n", us, (rd + wr2) / us * 1e-6 / 8);
        us = time_it([&]() { k_move<4, 2><<<1024, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
        printf("   ... grid 1024                         : %6.1f us  %.3f\n", us, (rd + wr2) / us * 1e-6 / 8);
        us = time_it([&]() { k_move<4, 2><<<8192, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
        printf("   ... grid 8192 (two runs per wave)     : %6.1f us  %.3f\n", us, (rd + wr2) / us * 1e-6 / 8);
        us = time_it([&]() { k_move<4, 2><<<16384, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
        printf("   ... grid 16384 (one run per wave)     : %6.1f us  %.3f\n", us, (rd + wr2) / us * 1e-6 / 8);
        {
            const char *names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0"};
            double ip[7], op[7];
            ip[0] = time_it([&]() { k_move<4, 2, 0><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
            op[0] = time_it([&]() { k_move<4, 2, 0><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            ip[1] = time_it([&]() { k_move<4, 2, 1><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
            op[1] = time_it([&]() { k_move<4, 2, 1><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            ip[2] = time_it([&]() { k_move<4, 2, 2><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
            op[2] = time_it([&]() { k_move<4, 2, 2><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            ip[3] = time_it([&]() { k_move<4, 2, 3><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
            op[3] = time_it([&]() { k_move<4, 2, 3><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            ip[4] = time_it([&]() { k_move<4, 2, 4><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
            op[4] = time_it([&]() { k_move<4, 2, 4><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            ip[5] = time_it([&]() { k_move<4, 2, 5><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
            op[5] = time_it([&]() { k_move<4, 2, 5><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            ip[6] = time_it([&]() { k_move<4, 2, 6><<<512, kBlock, kGldsRingBytes>>>(t0, t1, t0, t1, q, stride); });
            op[6] = time_it([&]() { k_move<4, 2, 6><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            for (int i = 0; i < 7; ++i)
                printf("stores %-11s: in place %6.1f us %.3f | out of place %6.1f us %.3f\n", names[i], ip[i], (rd + wr2) / ip[i] * 1e-6 / 8, op[i],
                       (rd + wr2) / op[i] * 1e-6 / 8);
        }
        // the fold's shape on the same structure: 2 streams read, 1 written per table (tables of 2q elements)
        us = time_it([&]() { k_move<2, 1><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
        printf("4 streams read, 2 written (fold shape)  : %6.1f us  %.3f\n", us, (2.0 * 2 * q * 32 + wr1) / us * 1e-6 / 8);
        for (uint32_t g : {2048u, 8192u, 16384u}) {
            us = time_it([&]() { k_move<2, 1><<<g, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
            printf("   ... fold shape, grid %5u              : %6.1f us  %.3f\n", g, us, (2.0 * 2 * q * 32 + wr1) / us * 1e-6 / 8);
        }
        us = time_it([&]() { k_move<2, 0><<<512, kBlock, kGldsRingBytes>>>(t0, t1, o0, o1, q, stride); });
        printf("4 streams read, none written            : %6.1f us  %.3f\n", us, (2.0 * 2 * q * 32) / us * 1e-6 / 8);
    }
    return 0;
}
