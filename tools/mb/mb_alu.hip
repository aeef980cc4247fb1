// mb_alu.hip -- raw VALU issue rates on gfx950 for the instructions a 256-bit modmul is made of (tuning harness).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 8 independent accumulators, each instruction repeated: body executes 8*REP instrs per loop trip
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ __launch_bounds__(256) void k_mad64(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint64_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mad64_addc(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint64_t acc[8]; uint32_t ex[8];
    for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x + i; ex[i] = i; }
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(ex[i]) : "v"(x), "v"(y) : "vcc");
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + ex[i];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mullo(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint32_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    uint32_t y = b ^ threadIdx.x | 1;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mulhi(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint32_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    uint32_t y = b ^ threadIdx.x | 1;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(256) void k_add(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint32_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    uint32_t y = b ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(256) void k_addc(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint32_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    uint32_t y = b ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(y) : "vcc");
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mad24(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint32_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    uint32_t y = b ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(y));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(256) void k_fma64(uint64_t* out, int iters, double a, double b) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    double y = b + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(y));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0.1234567) out[0] = (uint64_t)s;
}
__global__ __launch_bounds__(256) void k_fma32(uint64_t* out, int iters, float a, float b) {
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    float y = b + threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(y));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0.1234567f) out[0] = (uint64_t)s;
}
__global__ __launch_bounds__(256) void k_mov_dpp(uint64_t* out, int iters, uint32_t a, uint32_t b) {
    uint32_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));
        REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[0] = s;
}


#define GEN_KERNEL32(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(uint64_t* out, int iters, uint32_t a, uint32_t b) { \
    uint32_t acc[8]; for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a; \
    uint32_t y = (b ^ threadIdx.x) | 1; \
    for (int it = 0; it < iters; ++it) { \
        _Pragma("unroll") for (int rep = 0; rep < 4; ++rep) { \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(y)); } } \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i]; if (s == 0x1234567) out[0] = s; }
GEN_KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
GEN_KERNEL32(k_and, "v_and_b32 %0, %0, %1")
GEN_KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 29")
GEN_KERNEL32(k_andor, "v_and_or_b32 %0, %0, %1, %1")
GEN_KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
GEN_KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %1")
GEN_KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
GEN_KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
GEN_KERNEL32(k_sub_co, "v_sub_co_u32 %0, vcc, %0, %1")
#define GEN_KERNEL64(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(uint64_t* out, int iters, uint32_t a, uint32_t b) { \
    uint64_t acc[8]; for (int i = 0; i < 8; ++i) acc[i] = ((uint64_t)(threadIdx.x + i + a) << 32) | b; \
    uint64_t y = ((uint64_t)b << 20) ^ threadIdx.x; \
    for (int it = 0; it < iters; ++it) { \
        _Pragma("unroll") for (int rep = 0; rep < 4; ++rep) { \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(y)); } } \
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i]; if (s == 0x1234567) out[0] = s; }
GEN_KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 29, %0")
GEN_KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %1")

int main() {
    uint64_t* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, blocks = 256 * 8;   // 8 blocks/CU = 8 waves/SIMD
    auto run = [&](const char* name, auto launch, double instr_per_iter) {
        launch(10);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double wave_instr = (double)blocks * 4 * iters * instr_per_iter;            // wave-instructions issued
        double per_simd_per_s = wave_instr / (256.0 * 4) / (ms * 1e-3);
        printf("%-14s %8.3f ms   %.3e wave-instr/s/SIMD  -> %.2f cycles per wave-instr @2.4GHz\n", name, ms, per_simd_per_s, 2.4e9 / per_simd_per_s);
    };
    run("v_add_u32", [&](int it) { k_add<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_addc_co", [&](int it) { k_addc<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_mad_u32_u24", [&](int it) { k_mad24<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_mul_lo_u32", [&](int it) { k_mullo<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_mul_hi_u32", [&](int it) { k_mulhi<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_mad_u64_u32", [&](int it) { k_mad64<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("mad64+addc", [&](int it) { k_mad64_addc<<<blocks, 256>>>(out, it, 1, 2); }, 64);
    run("v_fma_f32", [&](int it) { k_fma32<<<blocks, 256>>>(out, it, 1.f, 2.f); }, 32);
    run("v_fma_f64", [&](int it) { k_fma64<<<blocks, 256>>>(out, it, 1.0, 2.0); }, 32);
    run("v_alignbit_b32", [&](int it) { k_alignbit<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_and_b32", [&](int it) { k_and<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_bfe_u32", [&](int it) { k_bfe<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_and_or_b32", [&](int it) { k_andor<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_lshl_or_b32", [&](int it) { k_lshl_or<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_add3_u32", [&](int it) { k_add3<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_cndmask_b32", [&](int it) { k_cndmask<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_add_co_u32", [&](int it) { k_add_co<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_sub_co_u32", [&](int it) { k_sub_co<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_lshrrev_b64", [&](int it) { k_lshr64<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_lshl_add_u64", [&](int it) { k_lshladd64<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    run("v_mov_dpp", [&](int it) { k_mov_dpp<<<blocks, 256>>>(out, it, 1, 2); }, 32);
    return 0;
}
