// mb_skew.hip -- does the distance between the fold's two read streams matter?  (tuning harness, not product code)
// The sumcheck fold reads lo = T[j] and hi = T[j + half] -- two streams exactly half a table (a power of two) apart -- and writes
// one.  This harness runs that access pattern (dwordx4 nontemporal, a xor instead of the multiply) with an extra GAP between
// the halves.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_skew.hip -o tools/mb/bin/mb_skew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ lo, const u32x4 *__restrict__ hi, u32x4 *__restrict__ out, uint64_t n16) {
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * 256) >> 6, lane = threadIdx.x & 63;
    for (uint64_t c0 = wave * 128 + lane; c0 < n16; c0 += nw * 128) {   // a wave moves 2 KiB of each stream per iteration
        const u32x4 a = __builtin_nontemporal_load(lo + c0), b = __builtin_nontemporal_load(lo + c0 + 64);
        const u32x4 c = __builtin_nontemporal_load(hi + c0), d = __builtin_nontemporal_load(hi + c0 + 64);
        __builtin_nontemporal_store(a ^ c, out + c0);
        __builtin_nontemporal_store(b ^ d, out + c0 + 64);
    }
}
int main() {
    const uint64_t half_bytes = 256ull << 20, n16 = half_bytes / 16;
    char *buf, *out;
    CK(hipMalloc(&buf, 2 * half_bytes + (64ull << 20)));
    CK(hipMalloc(&out, half_bytes));
    CK(hipMemset(buf, 1, 2 * half_bytes + (64ull << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint64_t gaps[] = {0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, (2 << 20) + 8192, (32 << 20) + 12288};
    for (int rep = 0; rep < 2; ++rep)
        for (uint64_t gap : gaps) {
            const u32x4 *lo = (const u32x4 *)buf, *hi = (const u32x4 *)(buf + half_bytes + gap);
            for (int i = 0; i < 5; ++i) k<<<16384, 256>>>(lo, hi, (u32x4 *)out, n16);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 40; ++i) k<<<16384, 256>>>(lo, hi, (u32x4 *)out, n16);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("gap %10llu B: %7.2f us  %6.3f TB/s\n", (unsigned long long)gap, ms / 40 * 1e3, 3.0 * half_bytes / (ms / 40 * 1e-3) / 1e12);
        }
    return 0;
}
