// mb_resident.hip -- does a RESIDENT transcript workgroup (second stream, fed and read through agent-scope flags) shorten the
// prover's per-round chain, compared with a tail kernel between two round kernels?  (tuning harness, not product code)
//
// Shape of one sumcheck round in the classic rounds of prove_core:
//   classic : round kernel (G workgroups, W us of work, one partial per workgroup) -> boundary -> tail kernel (one workgroup:
//             reduce the partials ~1.5 us, transcript step ~5 us, challenge to memory) -> boundary -> next round kernel.
//   resident: the round kernels stay back-to-back launches on stream A (tables are handed over by the kernel boundary, as
//             today); each workgroup publishes its partial write-through (sc1) and adds to a counter, the workgroup whose add
//             came last reduces (~1.5 us) and publishes the total + a flag; ONE resident workgroup on stream B polls the flag,
//             runs the transcript step (~5 us), publishes the challenge + a flag; the NEXT round kernel, whose launch boundary
//             overlaps all of that, polls the challenge flag in its prologue.
// Every spin is bounded (give-up word), every polled word zeroed before each run, epochs = round + 1.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_resident.hip -o tools/mb/bin/mb_resident
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Shared {              // one 4-KiB block, zeroed before every run
    unsigned eflag[64];      // round s: total published (epoch s + 1)
    unsigned rflag[64];      // round s: challenge published
    unsigned counter[64];    // arrivals of round s
    unsigned timeout;        // give-up word
    unsigned pad[63];
    unsigned long long total[64];
    unsigned long long chal[64];
};

__device__ __forceinline__ void spin_ticks(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
// one lane polls one word, relaxed, bounded (~3 ms); returns false after giving up
__device__ __forceinline__ bool wait_eq(unsigned *flag, unsigned epoch, unsigned *tmo, unsigned code) {
    const uint64_t t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load((gu32 *)flag, RLX_AGENT) == epoch) return true;
        if (__hip_atomic_load((gu32 *)tmo, RLX_AGENT) != 0) return false;
        if (wall_clock64() - t0 > 300000) { __hip_atomic_store((gu32 *)tmo, code, RLX_AGENT); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// round kernel.  MODE 0: classic (challenge of the previous round arrives through the kernel boundary); 1: polls rflag[s-1]
template <int MODE>
__global__ __launch_bounds__(256) void k_round(Shared *sh, unsigned long long *partials, int s, uint64_t work_ticks, uint64_t reduce_ticks) {
    __shared__ unsigned ok_s, last_s;
    unsigned long long r = 0;
    if (MODE == 1 && s > 0) {
        if (threadIdx.x == 0) ok_s = wait_eq(&sh->rflag[s - 1], (unsigned)s, &sh->timeout, 100 + s) ? 1u : 0u;
        __syncthreads();
        if (!ok_s) return;
        r = __hip_atomic_load((gu64 *)&sh->chal[s - 1], RLX_AGENT);
    } else if (s > 0) {
        r = sh->chal[s - 1];
    }
    spin_ticks(work_ticks);
    if (MODE == 0) {
        if (threadIdx.x == 0) partials[(size_t)s * gridDim.x + blockIdx.x] = r + blockIdx.x + 1;
        return;
    }
    if (threadIdx.x == 0) __hip_atomic_store((gu64 *)&partials[(size_t)s * gridDim.x + blockIdx.x], r + blockIdx.x + 1, RLX_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_s = __hip_atomic_fetch_add((gu32 *)&sh->counter[s], 1u, RLX_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_s) return;
    // last arriver: reduce the partials (sc1 loads), publish total + flag
    unsigned long long acc = 0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) acc += __hip_atomic_load((gu64 *)&partials[(size_t)s * gridDim.x + b], RLX_AGENT);
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ unsigned long long wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    spin_ticks(reduce_ticks);
    if (threadIdx.x == 0) {
        __hip_atomic_store((gu64 *)&sh->total[s], wsum[0] + wsum[1] + wsum[2] + wsum[3], RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((gu32 *)&sh->eflag[s], (unsigned)s + 1, RLX_AGENT);
    }
}
// classic tail: one workgroup reduces and runs the transcript step
__global__ __launch_bounds__(256) void k_tail(Shared *sh, const unsigned long long *partials, int s, unsigned grid, uint64_t reduce_ticks, uint64_t transcript_ticks) {
    unsigned long long acc = 0;
    for (unsigned b = threadIdx.x; b < grid; b += 256) acc += partials[(size_t)s * grid + b];
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ unsigned long long wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    spin_ticks(reduce_ticks + transcript_ticks);
    if (threadIdx.x == 0) { sh->total[s] = wsum[0] + wsum[1] + wsum[2] + wsum[3]; sh->chal[s] = sh->total[s] * 3 + 1; }
}
// ---- granule form: no counter, no last-arriver reduction, no separate flags --------------------------------------------------
// Every round-kernel workgroup publishes its partial as GR tagged 8-byte granules {epoch, value} (one sc1 store each: the data IS
// the flag); the resident workgroup (256 threads) sweeps all G * GR granules until every tag carries the round's epoch, reduces,
// runs the transcript step and publishes the challenge as ONE granule that the next round kernel's workgroups poll directly.
constexpr int GR = 24;   // a (D + 1) = 3 element partial is 96 bytes = 24 granules of 4 payload bytes
template <int DUMMY>
__global__ __launch_bounds__(256) void k_round_granule(Shared *sh, unsigned long long *gran, int s, uint64_t work_ticks) {
    __shared__ unsigned long long r_s;
    __shared__ unsigned ok_s;
    if (s > 0) {
        if (threadIdx.x == 0) {
            const uint64_t t0 = wall_clock64();
            ok_s = 0;
            for (;;) {
                const unsigned long long g = __hip_atomic_load((gu64 *)&sh->chal[s - 1], RLX_AGENT);
                if ((unsigned)(g >> 32) == (unsigned)s) { r_s = g & 0xffffffffu; ok_s = 1; break; }
                if (__hip_atomic_load((gu32 *)&sh->timeout, RLX_AGENT) != 0) break;
                if (wall_clock64() - t0 > 300000) { __hip_atomic_store((gu32 *)&sh->timeout, 300u + s, RLX_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (!ok_s) return;
    } else if (threadIdx.x == 0) {
        r_s = 0;
    }
    __syncthreads();
    spin_ticks(work_ticks);
    if (threadIdx.x < GR)
        __hip_atomic_store((gu64 *)&gran[((size_t)(s & 1) * gridDim.x + blockIdx.x) * GR + threadIdx.x],
                           ((unsigned long long)(s + 1) << 32) | ((r_s + blockIdx.x + 1) & 0xffffffffu), RLX_AGENT);
}
__global__ __launch_bounds__(256) void k_resident_granule(Shared *sh, const unsigned long long *gran, int rounds, unsigned grid, uint64_t reduce_ticks,
                                                           uint64_t transcript_ticks) {
    __shared__ unsigned long long wsum[4];
    __shared__ unsigned all_ok;
    for (int s = 0; s < rounds; ++s) {
        const unsigned long long *g = gran + (size_t)(s & 1) * grid * GR;
        const unsigned n = grid * GR;
        unsigned long long acc = 0;
        const uint64_t t0 = wall_clock64();
        for (;;) {   // sweep until every granule of this round has arrived
            bool ok = true;
            acc = 0;
            for (unsigned i = threadIdx.x; i < n; i += 256) {
                const unsigned long long x = __hip_atomic_load((gu64 *)&g[i], RLX_AGENT);
                ok &= (unsigned)(x >> 32) == (unsigned)(s + 1);
                if (i % GR == 0) acc += x & 0xffffffffu;
            }
            if (threadIdx.x == 0) all_ok = 1;
            __syncthreads();
            if (!ok) all_ok = 0;
            __syncthreads();
            if (all_ok) break;
            if (__hip_atomic_load((gu32 *)&sh->timeout, RLX_AGENT) != 0) return;
            if (wall_clock64() - t0 > 300000) { if (threadIdx.x == 0) __hip_atomic_store((gu32 *)&sh->timeout, 400u + s, RLX_AGENT); return; }
            __syncthreads();
        }
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x < 64) {
            spin_ticks(reduce_ticks + transcript_ticks);
            const unsigned long long t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (threadIdx.x == 0) {
                sh->total[s] = t;
                __hip_atomic_store((gu64 *)&sh->chal[s], ((unsigned long long)(s + 1) << 32) | ((t * 3 + 1) & 0xffffffffu), RLX_AGENT);
            }
        }
        __syncthreads();
    }
}
// resident transcript workgroup (one wave)
__global__ __launch_bounds__(64) void k_resident(Shared *sh, int rounds, uint64_t transcript_ticks) {
    for (int s = 0; s < rounds; ++s) {
        if (!wait_eq(&sh->eflag[s], (unsigned)s + 1, &sh->timeout, 200 + s)) return;
        const unsigned long long t = __hip_atomic_load((gu64 *)&sh->total[s], RLX_AGENT);
        spin_ticks(transcript_ticks);
        if (threadIdx.x == 0) {
            __hip_atomic_store((gu64 *)&sh->chal[s], t * 3 + 1, RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store((gu32 *)&sh->rflag[s], (unsigned)s + 1, RLX_AGENT);
        }
    }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    Shared *sh;
    unsigned long long *partials;
    CK(hipMalloc(&sh, sizeof(Shared)));
    CK(hipMalloc(&partials, 64 * 4096 * 8));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int R = 16;
    Shared *h = (Shared *)malloc(sizeof(Shared));
    for (int grid : {64, 512, 2048}) {
        for (uint64_t work : {200, 500, 2000}) {   // 2, 5, 20 us
            std::vector<double> tc, tr;
            unsigned long long want = 0;
            for (int rep = 0; rep < 24; ++rep) {
                CK(hipMemsetAsync(sh, 0, sizeof(Shared), sa));
                CK(hipStreamSynchronize(sa));
                auto t0 = std::chrono::steady_clock::now();
                for (int s = 0; s < R; ++s) {
                    k_round<0><<<grid, 256, 0, sa>>>(sh, partials, s, work, 150);
                    k_tail<<<1, 256, 0, sa>>>(sh, partials, s, grid, 150, 500);
                }
                CK(hipStreamSynchronize(sa));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep >= 4) tc.push_back(us / R);
                CK(hipMemcpy(h, sh, sizeof(Shared), hipMemcpyDeviceToHost));
                want = h->chal[R - 1];
            }
            int bad = 0, tmo = 0;
            for (int rep = 0; rep < 24; ++rep) {
                CK(hipMemsetAsync(sh, 0, sizeof(Shared), sa));
                CK(hipStreamSynchronize(sa));
                auto t0 = std::chrono::steady_clock::now();
                k_resident<<<1, 64, 0, sb>>>(sh, R, 500);
                for (int s = 0; s < R; ++s) k_round<1><<<grid, 256, 0, sa>>>(sh, partials, s, work, 150);
                CK(hipStreamSynchronize(sb));
                CK(hipStreamSynchronize(sa));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep >= 4) tr.push_back(us / R);
                CK(hipMemcpy(h, sh, sizeof(Shared), hipMemcpyDeviceToHost));
                if (h->timeout) ++tmo;
                else if (h->chal[R - 1] != want) ++bad;
            }
            // granule form (values are 32-bit here: compare the last challenge's payload with the classic run's low 32 bits of the same recurrence)
            std::vector<double> tg;
            int tmo_g = 0;
            unsigned long long *gran;
            CK(hipMalloc(&gran, (size_t)2 * grid * GR * 8));
            for (int rep = 0; rep < 24; ++rep) {
                CK(hipMemsetAsync(sh, 0, sizeof(Shared), sa));
                CK(hipMemsetAsync(gran, 0, (size_t)2 * grid * GR * 8, sa));
                CK(hipStreamSynchronize(sa));
                auto t0 = std::chrono::steady_clock::now();
                k_resident_granule<<<1, 256, 0, sb>>>(sh, gran, R, grid, 150, 500);
                for (int s = 0; s < R; ++s) k_round_granule<0><<<grid, 256, 0, sa>>>(sh, gran, s, work);
                CK(hipStreamSynchronize(sb));
                CK(hipStreamSynchronize(sa));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep >= 4) tg.push_back(us / R);
                CK(hipMemcpy(h, sh, sizeof(Shared), hipMemcpyDeviceToHost));
                if (h->timeout) ++tmo_g;
            }
            CK(hipFree(gran));
            printf("grid %4d work %4.0f us: granules %6.2f us/round (beyond work+reduce+transcript: %5.2f) timeouts %d\n", grid, work / 100.0, med(tg),
                   med(tg) - work / 100.0 - 6.5, tmo_g);
            printf("grid %4d work %4.0f us: classic %6.2f us/round | resident %6.2f us/round  (minus work+reduce+transcript %.1f: %5.2f vs %5.2f)  wrong %d timeouts %d\n",
                   grid, work / 100.0, med(tc), med(tr), work / 100.0 + 6.5, med(tc) - work / 100.0 - 6.5, med(tr) - work / 100.0 - 6.5, bad, tmo);
        }
    }
    return 0;
}
