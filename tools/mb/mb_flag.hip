// mb_flag.hip -- how a host learns that a short kernel has finished: (a) spinning on hipStreamQuery (what stream_wait does),
// (b) hipStreamSynchronize, (c) spinning on a word in pinned host memory that the kernel writes after its result
// (__threadfence_system between the two stores).  Prints the median wall clock of launch + wait over 2000 trials each.
// hipcc --offload-arch=gfx950 -O3 -o bin/mb_flag mb_flag.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_work(uint64_t *result, volatile uint32_t *flag, uint32_t seq, int spin) {
    uint64_t t0 = wall_clock64(), x = 0;
    while (wall_clock64() - t0 < (uint64_t)spin) x += 1;   // 100 MHz ticks
    result[0] = x + seq;
    if (flag) {
        __threadfence_system();
        *flag = seq;
    }
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint64_t *h_res;
    uint32_t *h_flag;
    hipHostMalloc((void **)&h_res, 64, hipHostMallocDefault);
    hipHostMalloc((void **)&h_flag, 64, hipHostMallocDefault);
    *h_flag = 0;
    for (int spin : {100, 500, 2000, 30000, 120000}) {   // 1, 5, 20, 300, 1200 us of kernel
        std::vector<double> a, b, c;
        uint32_t seq = 0;
        for (int i = 0; i < (spin > 5000 ? 400 : 2100); ++i) {
            auto t0 = std::chrono::steady_clock::now();
            k_work<<<1, 64, 0, s>>>(h_res, nullptr, ++seq, spin);
            while (hipStreamQuery(s) == hipErrorNotReady) {}
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (i >= 100) a.push_back(us);
        }
        for (int i = 0; i < (spin > 5000 ? 400 : 2100); ++i) {
            auto t0 = std::chrono::steady_clock::now();
            k_work<<<1, 64, 0, s>>>(h_res, nullptr, ++seq, spin);
            hipStreamSynchronize(s);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (i >= 100) b.push_back(us);
        }
        for (int i = 0; i < (spin > 5000 ? 400 : 2100); ++i) {
            auto t0 = std::chrono::steady_clock::now();
            k_work<<<1, 64, 0, s>>>(h_res, h_flag, ++seq, spin);
            while (*(volatile uint32_t *)h_flag != seq) {}
            std::atomic_thread_fence(std::memory_order_acquire);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (i >= 100) c.push_back(us);
            hipStreamSynchronize(s);   // (untimed) keep the trials apart
        }
        printf("kernel %4.0f us: launch + hipStreamQuery spin %6.2f us | hipStreamSynchronize %6.2f us | pinned flag %6.2f us\n", spin / 100.0, med(a), med(b), med(c));
    }
    return 0;
}
