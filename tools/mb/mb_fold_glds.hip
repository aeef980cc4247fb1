// mb_fold_glds.hip -- the MSB fold (k_fold_msb, the bench line's kernel) with its two input runs arriving by LDS-DMA.
//
// k_fold_msb already moves whole 1-KiB nontemporal pieces (register loads + one DPP half swap on each side) and sits at the copy rate
// (0.79 of 8 TB/s).  The round kernels gained from global_load_lds_dwordx4 because it fixed their access SHAPE; here the shape is already
// right, so what this harness asks is narrower: does taking the loads out of the VGPR file (deeper ring, no gather on the load side)
// move a kernel that is already at the streaming ceiling?  Variants: ring of H units (unit = lo run + hi run = 4 KiB) per wave.
// Checks: outputs bit-identical to k_fold_msb.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_fold_glds.hip -o tools/mb/bin/mb_fold_glds
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_field.hpp"
#include "kernels.cuh"
#include "round_kernels.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void wait_ops(uint32_t n) {   // n even, <= 28 (wave-uniform)
    switch (n >> 1) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<2>(); break;
        case 2: wait_vm<4>(); break;
        case 3: wait_vm<6>(); break;
        case 4: wait_vm<8>(); break;
        case 5: wait_vm<10>(); break;
        case 6: wait_vm<12>(); break;
        case 7: wait_vm<14>(); break;
        case 8: wait_vm<16>(); break;
        case 9: wait_vm<18>(); break;
        case 10: wait_vm<20>(); break;
        case 11: wait_vm<22>(); break;
        case 12: wait_vm<24>(); break;
        case 13: wait_vm<26>(); break;
        default: wait_vm<28>(); break;
    }
}

// out[j] = in[j] - r (in[j] - in[j + half]), half a multiple of 64; H units of ring per wave, WGS workgroups of 256 per CU intended
template <int H>
__global__ __launch_bounds__(kBlock) void k_fold_glds(const uint64_t *in, uint64_t *out, uint64_t half, FieldParams P, Mul29 r) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t fring[];   // [4 waves][H][4096]
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *my = fring + wave * (H * 4096);
    const uint32_t my_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)my);
    const uint32_t voff = lane * 16, own = pair_owned(lane);
    const uint64_t runs = half >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * 4 + wave, rs = (uint64_t)gridDim.x * 4;
    const uint64_t U = r0 < runs ? (runs - r0 + rs - 1) / rs : 0;
    const uint64_t base = (uint64_t)(uintptr_t)in, hi_off = half * 32;
    if (!U) return;
    auto issue = [&](uint64_t u, uint32_t pos) __attribute__((always_inline)) {
        const uint64_t a = base + (r0 + u * rs) * 2048;
        glds_rows2(a, a + hi_off, voff, my_lds + pos * 4096);
    };
    const uint64_t pre = U < (uint64_t)H ? U : (uint64_t)H;
    for (uint64_t i = 0; i < pre; ++i) issue(i, (uint32_t)i);
    uint32_t pos = 0;
    for (uint64_t u = 0; u < U; ++u) {
        const uint64_t e = U - 1 - u;   // units after this one
        // in flight behind this unit's pieces: the younger units' pieces (4 each) and the stores issued since (2 per unit, at most H units)
        const uint32_t younger = (uint32_t)(e < (uint64_t)(H - 1) ? e : (uint64_t)(H - 1));
        const uint32_t stored = (uint32_t)(u < (uint64_t)H ? u : (uint64_t)H);
        wait_ops(4 * younger + 2 * stored);
        const uint8_t *unit = my + pos * 4096;
        const Fe lo = glds_elem(unit, own), hi = glds_elem(unit + 2048, own);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (u + H < U) issue(u + H, pos);
        pos = pos + 1 == H ? 0 : pos + 1;
        const Fe o = fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P);
        run_store_nt(out + (r0 + u * rs) * 256, lane, o);
    }
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
        x.v[7] &= 0x0fffffffu;
        fe_store(t, i, x);
    }
}

template <int H>
static void go(uint32_t grid, const uint64_t *in, uint64_t *out, uint64_t half, const FieldParams &P, const Mul29 &r) {
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute((const void *)k_fold_glds<H>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * H * 4096));
        once = true;
    }
    k_fold_glds<H><<<grid, kBlock, 4 * H * 4096>>>(in, out, half, P, r);
}

int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 200;
    const uint64_t n = 1ull << log_n, half = n >> 1;
    const FieldInfo *fi = field_info(0);
    const FieldParams &P = fi->P;
    uint64_t *T, *A, *B;
    CK(hipMalloc(&T, n * 32));
    CK(hipMalloc(&A, n * 16));
    CK(hipMalloc(&B, n * 16));
    k_fill<<<2048, 256>>>(T, n, 0x5EED, P);
    const Mul29 r = mul29_prepare(fe_pow_u64(fi->two_adic_root, 12345, P), P);
    uint64_t g0 = half / kBlock;
    if (g0 > 32768) g0 = 32768;   // capi.hip launch_fold: 2 * kMaxGridStream
    auto shipped = [&]() { k_fold_msb<<<(uint32_t)g0, kBlock>>>(T, A, half, P, r); };
    shipped();
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> ha((size_t)half * 4), hb((size_t)half * 4);
    CK(hipMemcpy(ha.data(), A, ha.size() * 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto &&f) {
        for (int i = 0; i < 20; ++i) f();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.0 / reps;
    };
    struct V {
        const char *name;
        void (*launch)(uint32_t, const uint64_t *, uint64_t *, uint64_t, const FieldParams &, const Mul29 &);
        int h;
    };
    const V vars[] = {{"ring 1", go<1>, 1}, {"ring 2", go<2>, 2}, {"ring 3", go<3>, 3}, {"ring 4", go<4>, 4}};
    const double bytes = 1.5 * n * 32;
    for (int round = 0; round < 3; ++round) {
        const double us = time_it(shipped);
        printf("shipped k_fold_msb grid %5llu        : %7.2f us  %5.2f TB/s  %.3f of 8 TB/s\n", (unsigned long long)g0, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8);
        for (const V &v : vars) {
            const uint32_t grids[] = {512, 768, 1024, 1280, 2048, 4096};
            for (uint32_t g : grids) {
                if ((half >> 6) < (uint64_t)g * 4) continue;
                if (round == 0) {
                    CK(hipMemset(B, 0, n * 16));
                    v.launch(g, T, B, half, P, r);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(hb.data(), B, hb.size() * 8, hipMemcpyDeviceToHost));
                    if (memcmp(ha.data(), hb.data(), ha.size() * 8) != 0) {
                        printf("%s grid %u: output DIFFERS\n", v.name, g);
                        return 1;
                    }
                }
                const double u2 = time_it([&]() { v.launch(g, T, B, half, P, r); });
                printf("  LDS-DMA %-8s grid %5u            : %7.2f us  %5.2f TB/s  %.3f\n", v.name, g, u2, bytes / u2 * 1e-6, bytes / u2 * 1e-6 / 8);
            }
        }
    }
    printf("all variants bit-identical to k_fold_msb\n");
    return 0;
}
