// mb_graph.hip -- does a hipGraph shorten the gap between DEPENDENT small launches?  (tuning harness, not product code)
// The latency-bound rounds of the prover are chains of 5-15 us kernels, each needing the previous one's challenge.  This harness
// runs a chain of 64 one-workgroup kernels that spin for a fixed number of clock ticks (each reads the word the previous one
// wrote) (a) as plain launches on a stream, (b) as a captured graph, and prints the per-launch cadence minus the spin.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_graph.hip -o tools/mb/bin/mb_graph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_spin(uint64_t *chain, int i, uint64_t ticks) {   // 100 MHz ticks
    const uint64_t t0 = __builtin_readcyclecounter();
    uint64_t v = chain[i];
    uint64_t now = __builtin_amdgcn_s_memrealtime();
    const uint64_t start = now;
    while (now - start < ticks) now = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) chain[i + 1] = v + 1 + (t0 & 0);
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
    const int N = 64;
    for (uint64_t ticks : {100ull, 500ull, 1000ull}) {   // 1, 5, 10 us
        for (int grid : {1, 64}) {
            float ms_stream = 0, ms_graph = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < N; ++i) k_spin<<<grid, 256, 0, s>>>(chain, i, ticks);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_stream, e0, e1));
            }
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < N; ++i) k_spin<<<grid, 256, 0, s>>>(chain, i, ticks);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_graph, e0, e1));
            }
            printf("spin %5.1f us, grid %3d: stream %6.2f us/launch (gap %5.2f)   graph %6.2f us/launch (gap %5.2f)\n", ticks / 100.0, grid,
                   ms_stream / N * 1e3, ms_stream / N * 1e3 - ticks / 100.0, ms_graph / N * 1e3, ms_graph / N * 1e3 - ticks / 100.0);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
