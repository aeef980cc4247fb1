// mb_read.hip -- read-bandwidth ceiling of the round kernels' access pattern (tuning harness): 2 tables x (lo, hi) halves,
// 32 bytes per lane, up to PAIRS pair indices per thread with one or two indices of prefetch, XOR-reduced (no field math).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "host_field.hpp"
#include "kernels.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int WORK>   // WORK: dummy dependent VALU instructions per pair index (0 = pure read)
__global__ __launch_bounds__(256) void read4(const uint64_t* a, const uint64_t* b, uint64_t q, uint64_t* out) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < q; j += stride) {
        const Fe x0 = fe_load(a, j), x1 = fe_load(a, j + q), y0 = fe_load(b, j), y1 = fe_load(b, j + q);
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v ^= x0.v[i] ^ x1.v[i] ^ y0.v[i] ^ y1.v[i];
#pragma unroll 8
        for (int w = 0; w < WORK; ++w) v = v * 0x9E3779B1u + (uint32_t)w;
        acc ^= v;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;   // keep the loads alive
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 24;
    const uint64_t N = 1ull << n, q = N / 2;
    const FieldInfo* fi = field_info(0);
    uint64_t *a, *b, *out;
    CK(hipMalloc(&a, N * 32)); CK(hipMalloc(&b, N * 32)); CK(hipMalloc(&out, 4096));
    k_fill_random<<<2048, 256>>>(a, N, 1, 0, fi->P);
    k_fill_random<<<2048, 256>>>(b, N, 2, 0, fi->P);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * N * 32;
    auto run = [&](const char* name, auto f) {
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s %.1f us  %.0f GB/s\n", name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
    };
    for (int grid : {2048, 4096, 8192, 16384, 32768}) {
        char nm[64];
        snprintf(nm, sizeof nm, "read4 pure, grid %d", grid); run(nm, [&] { read4<0><<<grid, 256>>>(a, b, q, out); });
    }
    run("read4 + 600 dependent VALU, grid 2048", [&] { read4<600><<<2048, 256>>>(a, b, q, out); });
    run("read4 + 600 dependent VALU, grid 8192", [&] { read4<600><<<8192, 256>>>(a, b, q, out); });
    return 0;
}
