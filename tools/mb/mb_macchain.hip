// mb_macchain.hip -- how fast can a product-scanning column run?  (tuning harness, not product code)
// A column of a 256 x 256-bit product is a chain  acc(64) += a_i * b_j ; ex += carry  (v_mad_u64_u32 + v_addc_co_u32).
// V0: the library's order, one carry register (vcc):   mad, addc, mad, addc, ...
// V1: two carry registers, the addc one slot late:     mad0, mad1, addc0, mad0', addc1, ...   (same accumulator)
// V2: two independent accumulators interleaved:        madA, madB, addcA, addcB, ...
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_macchain.hip -o tools/mb/bin/mb_macchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
#define P0(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
template <int MODE>
__global__ void k(uint32_t seed, int iters, uint64_t *out) {
    uint32_t a0 = seed * 2654435761u + threadIdx.x, a1 = a0 * 3 + 1, a2 = a1 * 5 + 7, a3 = a2 * 7 + 3;
    uint32_t b0 = a3 ^ 0x9E3779B9u, b1 = b0 * 11 + 5, b2 = b1 * 13 + 1, b3 = b2 * 17 + 9;
    uint64_t acc = 0, accB = 0;
    uint32_t ex = 0, exB = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile(P0("%2", "%6") P0("%3", "%7") P0("%4", "%8") P0("%5", "%9") P0("%2", "%9") P0("%3", "%8") P0("%4", "%7") P0("%5", "%6")
                         : "+v"(acc), "+v"(ex) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");
        } else if (MODE == 1) {
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %2, %6, %0\n\t"
                "v_mad_u64_u32 %0, s[10:11], %3, %7, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %8, %0\n\t"
                "v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n\t"
                "v_mad_u64_u32 %0, s[10:11], %5, %9, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %9, %0\n\t"
                "v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n\t"
                "v_mad_u64_u32 %0, s[10:11], %3, %8, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %4, %7, %0\n\t"
                "v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n\t"
                "v_mad_u64_u32 %0, s[10:11], %5, %6, %0\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_addc_co_u32 %1, s[10:11], 0, %1, s[10:11]\n\t"
                : "+v"(acc), "+v"(ex) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc", "s10", "s11");
        } else {
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %4, %8, %0\n\t"
                "v_mad_u64_u32 %2, s[10:11], %4, %11, %2\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_addc_co_u32 %3, s[10:11], 0, %3, s[10:11]\n\t"
                "v_mad_u64_u32 %0, vcc, %5, %9, %0\n\t"
                "v_mad_u64_u32 %2, s[10:11], %5, %10, %2\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_addc_co_u32 %3, s[10:11], 0, %3, s[10:11]\n\t"
                "v_mad_u64_u32 %0, vcc, %6, %10, %0\n\t"
                "v_mad_u64_u32 %2, s[10:11], %6, %9, %2\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_addc_co_u32 %3, s[10:11], 0, %3, s[10:11]\n\t"
                "v_mad_u64_u32 %0, vcc, %7, %11, %0\n\t"
                "v_mad_u64_u32 %2, s[10:11], %7, %8, %2\n\t"
                "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_addc_co_u32 %3, s[10:11], 0, %3, s[10:11]\n\t"
                : "+v"(acc), "+v"(ex), "+v"(accB), "+v"(exB) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
                : "vcc", "s10", "s11");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + ex + accB + exB;
}
int main() {
    uint64_t *out;
    CK(hipMalloc(&out, 8 * 1024 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 100000;
    for (int wpb : {1, 2, 4}) {   // 256 CUs x 4 SIMDs: 1024 workgroups of `wpb` waves = wpb/ ... waves per SIMD ~ wpb * 1024 / 1024
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) k<0><<<1024, 64 * wpb>>>(rep, iters, out);
                else if (mode == 1) k<1><<<1024, 64 * wpb>>>(rep, iters, out);
                else k<2><<<1024, 64 * wpb>>>(rep, iters, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("%d wave(s) per SIMD, V%d: %.2f ns per mad+addc pair per wave  (%.2f ns per pair per SIMD)\n", wpb, mode, ms * 1e6 / iters / 8,
                   ms * 1e6 / iters / 8 / wpb);
        }
    }
    return 0;
}
