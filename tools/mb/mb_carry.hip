// mb_carry.hip -- is the s_nop the compiler puts between v_add_co_u32 and v_addc_co_u32 on gfx950 needed for CORRECTNESS?
// (tuning harness, not product code).  LLVM's hazard recogniser pads every VALU-written VCC before the VALU that reads it as
// carry-in with two wait states on gfx940+; the library's multiply-accumulate asm issues v_mad_u64_u32 / v_addc_co_u32 back to
// back and is bit-exact.  This harness runs an 8-limb add chain as ONE asm statement without padding against the compiler's
// chain on random operands (1 wave per SIMD and 8 waves per SIMD) and counts mismatches, and times both.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_carry.hip -o tools/mb/bin/mb_carry
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
struct U256 { uint32_t v[8]; };
__device__ __forceinline__ U256 add_ref(const U256 &a, const U256 &b) {
    U256 r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t co; r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co); c = co; }
    return r;
}
__device__ __forceinline__ U256 add_asm(const U256 &a, const U256 &b) {
    U256 r;
    asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
        "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %15, %23, vcc"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
        : "vcc");
    return r;
}
__device__ __forceinline__ uint32_t rnd(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33) | ((s >> 20) & 1 ? 0xFFFF0000u : 0); }
template <int MODE>   // 0: count mismatches; 1: time the compiler chain; 2: time the asm chain
__global__ void k(uint64_t seed, int iters, unsigned long long *bad, U256 *sink) {
    uint64_t s = seed + (uint64_t)blockIdx.x * 1000003 + threadIdx.x * 7919;
    U256 a, b;
    for (int i = 0; i < 8; ++i) { a.v[i] = rnd(s); b.v[i] = rnd(s); }
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            // operands with long carry runs: mix of random words, 0xFFFFFFFF and 0
            for (int i = 0; i < 8; ++i) { const uint32_t x = rnd(s); a.v[i] = (x & 3) == 0 ? 0xFFFFFFFFu : ((x & 3) == 1 ? 0u : x); b.v[i] = rnd(s) | ((x >> 5) & 1 ? 0xFFFFFFF0u : 0u); }
            const U256 r0 = add_ref(a, b), r1 = add_asm(a, b);
            for (int i = 0; i < 8; ++i) nbad += r0.v[i] != r1.v[i];
        } else if (MODE == 1) {
            a = add_ref(a, b); b = add_ref(b, a);
        } else {
            a = add_asm(a, b); b = add_asm(b, a);
        }
    }
    if (MODE == 0) { if (nbad) atomicAdd(bad, nbad); }
    else if (a.v[0] == 0x12345 && b.v[7] == 0x54321) sink[0] = a;
}
int main() {
    unsigned long long *bad; U256 *sink;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&sink, sizeof(U256))); CK(hipMemset(bad, 0, 8));
    for (int waves_per_block : {1, 4, 16}) {
        for (int rep = 0; rep < 4; ++rep) k<0><<<4096, 64 * waves_per_block>>>(1234 + rep * 77, 4000, bad, sink);
        CK(hipDeviceSynchronize());
        unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("%2d waves per workgroup: %llu mismatching limbs in %.2e chains\n", waves_per_block, h, 4.0 * 4096 * 64 * waves_per_block * 4000);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 1; mode <= 2; ++mode) {
        for (int w = 0; w < 2; ++w) {
            if (mode == 1) k<1><<<1024, 64>>>(1, 1000, bad, sink); else k<2><<<1024, 64>>>(1, 1000, bad, sink);
            CK(hipEventRecord(e0));
            if (mode == 1) k<1><<<1024, 64>>>(1, 100000, bad, sink); else k<2><<<1024, 64>>>(1, 100000, bad, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s chain, one wave per SIMD: %.2f ns per dependent 8-limb add\n", mode == 1 ? "compiler (padded)" : "asm (unpadded)   ", ms * 1e6 / 200000);
    }
    return 0;
}
