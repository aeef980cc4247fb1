// mb_stream.hip -- microbenchmarks for the HBM-streaming side of the fold (tuning harness, not product code).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_stream.hip -o tools/mb/bin/mb_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "host_field.hpp"
#include "kernels.cuh"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntl(const uint4* p) { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nts(uint4 v, uint4* p) { u32x4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p)); }
// ---------------- copy variants ----------------
template <bool NT>
__global__ __launch_bounds__(256) void copy_gs(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        uint4 v = NT ? ntl(in + j) : in[j];
        if (NT) nts(v, out + j); else out[j] = v;
    }
}
// each thread moves U uint4 per iteration, all loads issued before the stores
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_unroll(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < n; base += stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { uint64_t j = base + (uint64_t)u * blockDim.x; if (j < n) v[u] = NT ? ntl(in + j) : in[j]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { uint64_t j = base + (uint64_t)u * blockDim.x; if (j < n) { if (NT) nts(v[u], out + j); else out[j] = v[u]; } }
    }
}

// ---------------- fold variants ----------------
// A: product kernel (k_fold from kernels.cuh)
// B: nontemporal loads/stores
ZK_D Fe fe_load_nt(const uint64_t* base, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(base + 4 * idx);
    uint4 a = ntl(q), b = ntl(q + 1);
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
ZK_D void fe_store_nt(uint64_t* base, uint64_t idx, const Fe& r) {
    uint4* q = reinterpret_cast<uint4*>(base + 4 * idx);
    nts(make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]), q);
    nts(make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]), q + 1);
}
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void fold_v(const uint64_t* in, uint64_t* out, uint64_t half, FieldParams P, Fe r) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < half; base += stride) {
        Fe lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t j = base + (uint64_t)u * blockDim.x;
            if (j < half) { lo[u] = NTL ? fe_load_nt(in, j) : fe_load(in, j); hi[u] = NTL ? fe_load_nt(in, j + half) : fe_load(in, j + half); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t j = base + (uint64_t)u * blockDim.x;
            if (j < half) {
                Fe o = fe_sub(lo[u], fe_mul(r, fe_sub(lo[u], hi[u], P), P), P);
                if (NTS) fe_store_nt(out, j, o); else fe_store(out, j, o);
            }
        }
    }
}
// fold with the modular multiply stubbed out (same loads/stores, same subs): what the access pattern alone sustains
template <int MODE>
__global__ __launch_bounds__(256) void fold_nomul(const uint64_t* in, uint64_t* out, uint64_t half, FieldParams P, Fe r) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < half; j += stride) {
        const Fe lo = fe_load(in, j), hi = fe_load(in, j + half);
        Fe d = fe_sub(lo, hi, P);
        if (MODE == 1) { Fe m = d; for (int i = 0; i < 8; ++i) m.v[i] ^= r.v[i]; d = m; }   // trivial "multiply"
        fe_store(out, j, MODE == 0 ? d : fe_sub(lo, d, P));
    }
}
// C: lane-pair coalesced access: a wave reads 128 consecutive elements as 4 fully coalesced 1-KiB loads per stream;
// lanes 2i / 2i+1 exchange halves so that each lane ends up with 2 whole elements.
template <bool NT>
__global__ __launch_bounds__(256) void fold_pair(const uint64_t* in, uint64_t* out, uint64_t half, FieldParams P, Fe r) {
    // each wave handles blocks of 64 elements: 2 loads of 1 KiB per stream -> each lane 1 element... (2 chunks -> 1 elem per lane)
    const uint32_t lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const uint64_t wave_global = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    uint4* out4 = reinterpret_cast<uint4*>(out);
    for (uint64_t e0 = wave_global * 64; e0 < half; e0 += nwaves * 64) {   // half is a multiple of 64 here
        // element e0+i occupies uint4 index 2*(e0+i), 2*(e0+i)+1 ; load k covers uint4 [2*e0 + 64k, +64)
        const uint64_t c0 = 2 * e0 + lane;
        uint4 la = NT ? ntl(in4 + c0) : in4[c0];
        uint4 lb = NT ? ntl(in4 + c0 + 64) : in4[c0 + 64];
        uint4 ha = NT ? ntl(in4 + c0 + 2 * half) : in4[c0 + 2 * half];
        uint4 hb = NT ? ntl(in4 + c0 + 2 * half + 64) : in4[c0 + 2 * half + 64];
        Fe lo = pair_gather(la, lb, odd), hi = pair_gather(ha, hb, odd);
        Fe o = fe_sub(lo, fe_mul(r, fe_sub(lo, hi, P), P), P);
        uint4 oa, ob;
        pair_scatter(o, odd, oa, ob);
        if (NT) { nts(oa, out4 + c0); nts(ob, out4 + c0 + 64); }
        else { out4[c0] = oa; out4[c0 + 64] = ob; }
    }
}

// product kernel structure (k_fold_msb) with the carry-free multiply, U runs of 64 elements in flight per wave
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fold_pair29(const uint64_t* in, uint64_t* out, uint64_t half, FieldParams P, Mul29 r) {
    const uint32_t lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const uint64_t wave = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * BLOCK) >> 6;
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    uint4* out4 = reinterpret_cast<uint4*>(out);
    for (uint64_t e0 = wave * 64 * U; e0 < half; e0 += nwaves * 64 * U) {
        uint4 la[U], lb[U], ha[U], hb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t c0 = 2 * (e0 + 64 * u) + lane;
            la[u] = ntl(in4 + c0); lb[u] = ntl(in4 + c0 + 64); ha[u] = ntl(in4 + c0 + 2 * half); hb[u] = ntl(in4 + c0 + 2 * half + 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t c0 = 2 * (e0 + 64 * u) + lane;
            Fe lo = pair_gather(la[u], lb[u], odd), hi = pair_gather(ha[u], hb[u], odd);
            Fe o = fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P);
            uint4 oa, ob;
            pair_scatter(o, odd, oa, ob);
            nts(oa, out4 + c0); nts(ob, out4 + c0 + 64);
        }
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> double ms(F f, int reps) {
        f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        return t / reps;
    }
};

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 24;
    const uint64_t N = 1ull << n, half = N / 2;
    const FieldInfo* fi = field_info(0);
    const FieldParams P = fi->P;
    uint64_t *in, *out, *ref;
    CK(hipMalloc(&in, N * 32)); CK(hipMalloc(&out, half * 32)); CK(hipMalloc(&ref, half * 32));
    k_fill_random<<<2048, 256>>>(in, N, 42, 0, P);
    Fe r = fi->two_adic_root;
    Timer T;
    const double fold_bytes = 48.0 * N, copy_bytes = 2.0 * half * 32;
    printf("# n=%d  fold algorithmic bytes %.0f\n", n, fold_bytes);
    // ---- copy (read half-table bytes -> write same): 256 MiB in, 256 MiB out at n=24
    {
        const uint4* ci = reinterpret_cast<const uint4*>(in); uint4* co = reinterpret_cast<uint4*>(out); uint64_t n16 = half * 2;
        for (int grid : {1024, 2048, 4096, 8192, 16384}) {
            double t0 = T.ms([&] { copy_gs<false><<<grid, 256>>>(ci, co, n16); }, 20);
            double t1 = T.ms([&] { copy_gs<true><<<grid, 256>>>(ci, co, n16); }, 20);
            double t2 = T.ms([&] { copy_unroll<4, false><<<grid, 256>>>(ci, co, n16); }, 20);
            double t3 = T.ms([&] { copy_unroll<4, true><<<grid, 256>>>(ci, co, n16); }, 20);
            double t4 = T.ms([&] { copy_unroll<8, true><<<grid, 256>>>(ci, co, n16); }, 20);
            printf("copy grid %5d: gs %.0f  gs_nt %.0f  u4 %.0f  u4_nt %.0f  u8_nt %.0f GB/s\n", grid, copy_bytes / t0 / 1e6, copy_bytes / t1 / 1e6,
                   copy_bytes / t2 / 1e6, copy_bytes / t3 / 1e6, copy_bytes / t4 / 1e6);
        }
        double tm = T.ms([&] { CK(hipMemcpyAsync(out, in, half * 32, hipMemcpyDeviceToDevice, 0)); }, 20);
        printf("copy hipMemcpyDtoD: %.0f GB/s\n", copy_bytes / tm / 1e6);
    }
    // ---- fold
    k_fold<<<2048, 256>>>(in, ref, half, n - 1, P, mul29_prepare(r, P));
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> href(half * 4), hout(half * 4);
    CK(hipMemcpy(href.data(), ref, half * 32, hipMemcpyDeviceToHost));
    auto check = [&](const char* name) {
        CK(hipMemcpy(hout.data(), out, half * 32, hipMemcpyDeviceToHost));
        if (memcmp(href.data(), hout.data(), half * 32)) printf("  !! %s MISMATCH\n", name);
        CK(hipMemset(out, 0, half * 32));
    };
    {
        const Mul29 r29 = mul29_prepare(r, P);
        for (int rep = 0; rep < 2; ++rep) {
            for (int grid : {4096, 8192, 16384, 32768}) {
                double a1 = T.ms([&] { fold_pair29<1, 256><<<grid, 256>>>(in, out, half, P, r29); }, 30); check("p29_u1_256");
                double a2 = T.ms([&] { fold_pair29<2, 256><<<grid, 256>>>(in, out, half, P, r29); }, 30); check("p29_u2_256");
                double a3 = T.ms([&] { fold_pair29<1, 512><<<grid / 2, 512>>>(in, out, half, P, r29); }, 30); check("p29_u1_512");
                double a4 = T.ms([&] { fold_pair29<2, 512><<<grid / 2, 512>>>(in, out, half, P, r29); }, 30); check("p29_u2_512");
                double a5 = T.ms([&] { k_fold_msb<<<grid, 256>>>(in, out, half, P, r29); }, 30); check("k_fold_msb");
                printf("pair29 grid %5d: u1_256 %.0f  u2_256 %.0f  u1_512 %.0f  u2_512 %.0f  product k_fold_msb %.0f GB/s\n", grid, fold_bytes / a1 / 1e6,
                       fold_bytes / a2 / 1e6, fold_bytes / a3 / 1e6, fold_bytes / a4 / 1e6, fold_bytes / a5 / 1e6);
            }
        }
    }
    for (int grid : {1024, 2048, 4096, 8192, 16384, 32768}) {
        double a = T.ms([&] { k_fold<<<grid, 256>>>(in, out, half, n - 1, P, mul29_prepare(r, P)); }, 20); check("k_fold");
        double b = T.ms([&] { fold_v<true, true, 1><<<grid, 256>>>(in, out, half, P, r); }, 20); check("nt");
        double c = T.ms([&] { fold_v<false, false, 2><<<grid, 256>>>(in, out, half, P, r); }, 20); check("u2");
        double d = T.ms([&] { fold_v<true, true, 2><<<grid, 256>>>(in, out, half, P, r); }, 20); check("u2nt");
        double e = T.ms([&] { fold_v<true, false, 1><<<grid, 256>>>(in, out, half, P, r); }, 20); check("ntl");
        double f = T.ms([&] { fold_pair<false><<<grid, 256>>>(in, out, half, P, r); }, 20); check("pair");
        double g = T.ms([&] { fold_pair<true><<<grid, 256>>>(in, out, half, P, r); }, 20); check("pair_nt");
        double h0 = T.ms([&] { fold_nomul<0><<<grid, 256>>>(in, out, half, P, r); }, 20);
        double h1 = T.ms([&] { fold_nomul<1><<<grid, 256>>>(in, out, half, P, r); }, 20);
        printf("fold grid %5d: nomul(sub only) %.0f  nomul(2 subs+xor) %.0f GB/s\n", grid, fold_bytes / h0 / 1e6, fold_bytes / h1 / 1e6);
        printf("fold grid %5d: base %.0f  nt %.0f  u2 %.0f  u2nt %.0f  ntl %.0f  pair %.0f  pair_nt %.0f GB/s\n", grid, fold_bytes / a / 1e6, fold_bytes / b / 1e6,
               fold_bytes / c / 1e6, fold_bytes / d / 1e6, fold_bytes / e / 1e6, fold_bytes / f / 1e6, fold_bytes / g / 1e6);
    }
    return 0;
}
