// mb_launch.hip -- what makes a dependent launch cost 3-4 us in the prover when bare kernels chain at 1.2-1.9 us?
// (tuning harness, not product code).  Chain of 64 dependent kernels that spin 5 us; variants: kernel-argument size (64 B / 768 B by
// value), static LDS (0 / 40 KiB), grid (1 / 64 / 256 workgroups of 256 threads).  Prints the cadence minus the spin.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_launch.hip -o tools/mb/bin/mb_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
template <int N> struct Args { uint32_t w[N]; };
template <int NARG, int LDS>
__global__ __launch_bounds__(256) void k(uint64_t *chain, int i, uint64_t ticks, Args<NARG> a) {
    __shared__ uint32_t lds[LDS ? LDS / 4 : 1];
    if (LDS) lds[threadIdx.x] = a.w[threadIdx.x % NARG];
    uint64_t v = chain[i];
    uint64_t now = __builtin_amdgcn_s_memrealtime();
    const uint64_t start = now;
    while (now - start < ticks) now = __builtin_amdgcn_s_memrealtime();
    if (LDS) v += lds[(threadIdx.x * 7) % 256];
    if (threadIdx.x == 0 && blockIdx.x == 0) chain[i + 1] = v + a.w[i % NARG];
}
template <int NARG, int LDS>
static void run(const char *name, hipStream_t s, uint64_t *chain, hipEvent_t e0, hipEvent_t e1) {
    Args<NARG> a = {};
    const int N = 64;
    for (int grid : {1, 64, 256}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) k<NARG, LDS><<<grid, 256, 0, s>>>(chain, i, 500, a);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%-34s grid %3d: %6.2f us per launch, %5.2f beyond the 5 us spin\n", name, grid, ms / N * 1e3, ms / N * 1e3 - 5.0);
    }
}
int main() {
    uint64_t *chain;
    CK(hipMalloc(&chain, 4096));
    CK(hipMemset(chain, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    run<16, 0>("64 B of arguments, no LDS", s, chain, e0, e1);
    run<192, 0>("768 B of arguments, no LDS", s, chain, e0, e1);
    run<16, 40960>("64 B of arguments, 40 KiB LDS", s, chain, e0, e1);
    run<192, 40960>("768 B of arguments, 40 KiB LDS", s, chain, e0, e1);
    return 0;
}
