// mb_fma52.hip -- can the FP64 FMA pipe beat v_mad_u64_u32 as the wide multiplier of a 256-bit modular product?  (tuning harness)
//
// The review of round 3 asked for a Montgomery multiplication on 52-bit limbs in double precision (v_fma_f64 hi / lo pairs).  Its
// cost is decided by the WIDE PRODUCT core -- the reduction is two more products of the same shape -- so this harness builds and
// times exactly that core in both forms, each bit-exact against a plain 64-bit schoolbook product computed in the same kernel:
//
//   mad29 : 9 x 9 limbs of 29 bits, 81 v_mad_u64_u32 straight into 17 64-bit column sums (what fe_mul29 / k_eval_stream do);
//   fma52 : 5 x 5 limbs of 52 bits held as doubles.  Per limb product (Emmart / Zheng's DPF scheme, round-toward-zero):
//               hi = fma_rz(a, b, 2^104)            -> 2^104 + floor(ab / 2^52) * 2^52   (exact: ulp is 2^52 in that binade)
//               lo = fma_rz(a, b, (2^104 + 2^52) - hi) = 2^52 + (ab mod 2^52)            (exact: ulp is 1)
//           and the two 52-bit halves are summed into 64-bit integer columns from the doubles' bit patterns (the exponent bits
//           add up to a known constant that is subtracted once per column): 2 v_fma_f64 + 1 v_add_f64 + 2 64-bit integer adds
//           per limb product, 25 limb products.
//
// Instruction rates measured on this part (tools/mb/mb_alu.hip, profiles/r04_mb_alu_rates.log): v_fma_f64 4.37 cycles per
// wave-instruction, v_mad_u64_u32 4.95, v_lshl_add_u64 4.59 -- the FP64 pipe is NOT faster than the integer multiplier here.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_fma52.hip -o tools/mb/bin/mb_fma52
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned __int128 u128;

__device__ __forceinline__ uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// reference: 256 x 256 -> 512 bits, 4 x 4 u64 schoolbook
__device__ void ref_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            const u128 t = (u128)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (uint64_t)t;
            carry = (uint64_t)(t >> 64);
        }
        out[i + 4] = carry;
    }
}
// ---- mad29: 9 x 9 limbs, 81 mads into 17 columns ----------------------------------------------------------------------------
__device__ __forceinline__ void split29(const uint64_t a[4], uint32_t l[9]) {
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        uint64_t v = a[w] >> sh;
        if (sh > 35 && w + 1 < 4) v |= a[w + 1] << (64 - sh);
        l[i] = (uint32_t)(i < 8 ? v & ((1u << 29) - 1) : v);   // limb 8 = bits 232..255
    }
}
__device__ __forceinline__ void mul_mad29(const uint32_t x[9], const uint32_t y[9], uint64_t col[17]) {
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) col[i + j] += (uint64_t)x[i] * y[j];
}
// ---- fma52: 5 x 5 limbs as doubles ------------------------------------------------------------------------------------------
__device__ __forceinline__ void split52(const uint64_t a[4], double l[5]) {
    for (int i = 0; i < 5; ++i) {
        const int bit = 52 * i, w = bit >> 6, sh = bit & 63;
        uint64_t v = a[w] >> sh;
        if (sh > 12 && w + 1 < 4) v |= a[w + 1] << (64 - sh);
        v &= (1ull << 52) - 1;
        l[i] = __longlong_as_double((long long)(v | 0x4330000000000000ull)) - 4503599627370496.0;   // (2^52 + v) - 2^52, exact
    }
}
// the kernels set MODE.FP_ROUND[3:2] (double precision) to round-toward-zero once, at their start; every FMA below then truncates
// (through inline asm: the backend's mode-register pass otherwise notices the change and resets the mode in front of the next
// double-precision instruction it knows about; the FMAs are asm for the same reason -- the compiler's own v_add_f64 for the exact
// subtractions run under the same mode and do not care)
__device__ __forceinline__ void set_double_rtz() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3"); }
__device__ __forceinline__ double fma_rz(double a, double b, double c) {
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ void mul_fma52(const double x[5], const double y[5], uint64_t col[10]) {
    const double C1 = 20282409603651670423947251286016.0;                 // 2^104
    const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;   // 2^104 + 2^52 (exact)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double hi = fma_rz(x[i], y[j], C1);
            const double lo = fma_rz(x[i], y[j], C2 - hi);
            col[i + j] += (uint64_t)__double_as_longlong(lo);       // 2^52 + (ab mod 2^52) as an integer: exponent bits 0x433
            col[i + j + 1] += (uint64_t)__double_as_longlong(hi);   // mantissa = floor(ab / 2^52): exponent bits 0x467
        }
}
// columns (with their exponent-bit constants removed) -> 512-bit integer
__device__ void fma52_finish(const uint64_t col[10], uint64_t out[8]) {
    // column k received n_lo(k) lo parts and n_hi(k) hi parts: products (i, j) with i + j == k resp. i + j + 1 == k
    for (int i = 0; i < 8; ++i) out[i] = 0;
    u128 carry = 0;
    uint64_t limbs[10];
    for (int k = 0; k < 10; ++k) {
        int n_lo = 0, n_hi = 0;
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                n_lo += (i + j == k);
                n_hi += (i + j + 1 == k);
            }
        const uint64_t v = col[k] - (uint64_t)n_lo * 0x4330000000000000ull - (uint64_t)n_hi * 0x4670000000000000ull;
        const u128 t = (u128)v + carry;
        limbs[k] = (uint64_t)(t & ((((u128)1) << 52) - 1));
        carry = t >> 52;
    }
    for (int k = 0; k < 10; ++k) {   // limb k at bit 52 k
        const int bit = 52 * k, w = bit >> 6, sh = bit & 63;
        if (w < 8) out[w] |= limbs[k] << sh;
        if (sh > 12 && w + 1 < 8) out[w + 1] |= limbs[k] >> (64 - sh);
    }
}
__device__ void mad29_finish(const uint64_t col[17], uint64_t out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    u128 carry = 0;
    for (int k = 0; k < 18; ++k) {
        const u128 t = (k < 17 ? (u128)col[k] : 0) + carry;
        const uint64_t limb = (uint64_t)(t & ((1u << 29) - 1));
        carry = t >> 29;
        const int bit = 29 * k, w = bit >> 6, sh = bit & 63;
        if (w < 8) out[w] |= limb << sh;
        if (sh > 35 && w + 1 < 8) out[w + 1] |= limb >> (64 - sh);
    }
}

__global__ void k_check(uint64_t seed, int n, unsigned *bad29, unsigned *bad52) {
    set_double_rtz();
    uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x1000003;
    for (int it = 0; it < n; ++it) {
        uint64_t a[4], b[4], want[8], got[8];
        for (int i = 0; i < 4; ++i) a[i] = splitmix(s), b[i] = splitmix(s);
        if (it == 0) for (int i = 0; i < 4; ++i) a[i] = b[i] = ~0ull;          // all ones
        if (it == 1) for (int i = 0; i < 4; ++i) a[i] = 0, b[i] = ~0ull;
        if (it == 2) { a[0] = 1; a[1] = a[2] = a[3] = 0; }
        if (it == 3) for (int i = 0; i < 4; ++i) a[i] = 0x000FFFFFFFFFFFFFull << (i * 3), b[i] = 0xFFFFFFFFFFFFF000ull >> i;
        ref_mul(a, b, want);
        uint32_t x[9], y[9];
        uint64_t c29[17] = {};
        split29(a, x), split29(b, y);
        mul_mad29(x, y, c29);
        mad29_finish(c29, got);
        bool ok = true;
        for (int i = 0; i < 8; ++i) ok &= got[i] == want[i];
        if (!ok) atomicAdd(bad29, 1u);
        double xd[5], yd[5];
        uint64_t c52[10] = {};
        split52(a, xd), split52(b, yd);
        mul_fma52(xd, yd, c52);
        fma52_finish(c52, got);
        ok = true;
        for (int i = 0; i < 8; ++i) ok &= got[i] == want[i];
        if (!ok) atomicAdd(bad52, 1u);
    }
}
// register-resident rate of each core: the product of two operand sets that change a little every iteration (so nothing is
// hoisted), column sums kept live across the loop
__global__ __launch_bounds__(256) void k_rate29(uint64_t *out, int iters) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; ++i) x[i] = threadIdx.x * 7 + i, y[i] = threadIdx.x * 13 + 3 * i + 1;
    uint64_t col[17] = {};
    for (int it = 0; it < iters; ++it) {
        mul_mad29(x, y, col);
        x[it % 9] ^= (uint32_t)col[3] & 0xFFFFF;
        if ((it & 7) == 7)
            for (int k = 0; k < 16; ++k) col[k + 1] += col[k] >> 29, col[k] &= (1u << 29) - 1;   // the carry normalisation it needs
    }
    uint64_t acc = 0;
    for (int k = 0; k < 17; ++k) acc += col[k];
    if (acc == 0x1234567) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_rate52(uint64_t *out, int iters) {
    set_double_rtz();
    double x[5], y[5];
    for (int i = 0; i < 5; ++i) x[i] = (double)(threadIdx.x * 7 + i + 1), y[i] = (double)(threadIdx.x * 13 + 3 * i + 1);
    uint64_t col[10] = {};
    for (int it = 0; it < iters; ++it) {
        mul_fma52(x, y, col);
        x[it % 5] = (double)((col[3] & 0xFFFFF) + 1);
    }
    uint64_t acc = 0;
    for (int k = 0; k < 10; ++k) acc += col[k];
    if (acc == 0x1234567) out[0] = acc;
}

int main() {
    unsigned *bad;
    CK(hipMalloc(&bad, 8));
    CK(hipMemset(bad, 0, 8));
    k_check<<<256, 256>>>(0xF00D, 10, bad, bad + 1);   // 655,360 products, edge operands in the first four of every thread
    CK(hipDeviceSynchronize());
    unsigned h[2];
    CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    printf("bit-exact vs 64-bit schoolbook on 655360 random + edge 256 x 256 -> 512-bit products: mad29 mismatches %u, fma52 mismatches %u\n", h[0], h[1]);
    uint64_t *out;
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        float ms29, ms52;
        k_rate29<<<blocks, 256>>>(out, 10);
        CK(hipEventRecord(e0));
        k_rate29<<<blocks, 256>>>(out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms29, e0, e1));
        k_rate52<<<blocks, 256>>>(out, 10);
        CK(hipEventRecord(e0));
        k_rate52<<<blocks, 256>>>(out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms52, e0, e1));
        const double n = (double)blocks * 256 * iters;
        printf("wide products per second, register resident: mad29 (81 v_mad_u64_u32) %.3e   fma52 (50 v_fma_f64 + 25 v_add_f64 + 50 u64 adds) %.3e   ratio fma52 / mad29 %.2f\n",
               n / (ms29 * 1e-3), n / (ms52 * 1e-3), ms29 / ms52);
    }
    return 0;
}
