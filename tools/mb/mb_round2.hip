// mb_round2.hip -- "two rounds per pass", priced with a purpose-built throughput kernel.
//
// The prover's big rounds each read a table and write its half: round s folds table_{s-1} at r_{s-1} and sums over table_s (k_round_kd,
// SKIP1 + LEAD: 3 modular multiplications per input element pair and factor).  Two consecutive rounds move 2.25 N x 32 B per factor.  A pass
// that starts with BOTH r_{s-2} and r_{s-1} known can fold twice (table_{s-2} -> table_s, N read, N/4 written: 1.25 N), form S_s directly and
// prepare round s+1 in the pending-challenge form E_{s+1}(t; rho) at the nodes rho in {0, inf, 1} x t in {0, 1, 2}; S_s then comes out of the
// same nine sums for free (S_s(0) = E(0;0) + E(1;0), S_s(1) = E(0;1) + E(1;1), leading coefficient = E(0;inf) + E(1;inf)), and a tail closes
// rounds s and s+1 back to back.  Per 16 input elements of each of the two factors: 24 fold multiplications + 9 products = 33 modular
// multiplications for 1280 B moved, against 36 for 2304 B in two classic rounds -- the same arithmetic on 44 % fewer bytes.
//
// This harness builds exactly that pass (K = 2 factors, no transcript plumbing) in the layout that fits the registers -- one 16-element group
// per QUAD: lane c folds elements c of both factors (6 multiplications by the wave-uniform challenges), DPP quad broadcasts hand the four
// folded values of a factor to every lane, lane = node forms its three products (t = 0, 1, 2) into unreduced accumulators -- and times it
// against two launches of the shipped k_round_kd<2,2,FUSED,SKIP1,LEAD> on the same tables, after checking that (a) the quarter tables are
// bit-identical and (b) S(0) and the leading coefficient derived from the nine sums equal the classic second round's.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk_amd/csrc tools/mb/mb_round2.hip -o tools/mb/bin/mb_round2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_field.hpp"
#include "round_kernels.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int kThreads = 256, kQuads = kThreads / 4;
struct NodeMasks {
    uint32_t u, v, w;   // node 0 (rho = 0), inf, 1
};
__device__ __forceinline__ Fe node_value(const Fe &p, const Fe &q, const NodeMasks &m, const FieldParams &P) {   // p: rho = 0, q: rho = 1
    const Fe d = fe_sub(q, p, P);
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = (p.v[i] & m.u) | (d.v[i] & m.v) | (q.v[i] & m.w);
    return o;
}
__device__ __forceinline__ void store_wt(uint64_t *base, uint64_t idx, const Fe &r) {
    u32x4_t *q = reinterpret_cast<u32x4_t *>(base + 4 * idx);
    const u32x4_t lo = {r.v[0], r.v[1], r.v[2], r.v[3]}, hi = {r.v[4], r.v[5], r.v[6], r.v[7]};
    __builtin_nontemporal_store(lo, q);
    __builtin_nontemporal_store(hi, q + 1);
}
__device__ __forceinline__ Fe load_nt(const uint64_t *base, uint64_t idx) {
    const u32x4_t *q = reinterpret_cast<const u32x4_t *>(base + 4 * idx);
    const u32x4_t a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}

// in: two tables of N elements; out: two tables of N / 4; partials: [block][9] = E(t; node) at [t * 3 + node]
template <bool PREFETCH>
__global__ __launch_bounds__(kThreads, 2) void k_round2(const uint64_t *__restrict__ t0, const uint64_t *__restrict__ t1, uint64_t *__restrict__ o0,
                                                         uint64_t *__restrict__ o1, uint64_t n, FieldParams P, Mul29 ra, Mul29 rb,
                                                         uint64_t *__restrict__ partials) {
    __shared__ Fe redw[kThreads / 64][12];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 3;
    NodeMasks m;
    m.u = c == 0 ? 0xffffffffu : 0u;
    m.v = c == 1 ? 0xffffffffu : 0u;
    m.w = c == 2 ? 0xffffffffu : 0u;
    const uint64_t groups = n >> 4, q16 = n >> 4, q4 = n >> 2;
    WideAcc acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) wide_zero(acc[t]);
    const uint64_t stride = (uint64_t)gridDim.x * kQuads;
    uint64_t g = (uint64_t)blockIdx.x * kQuads + (threadIdx.x >> 2);
    Fe cur[2][4], nxt[2][4];
    auto load = [&](Fe (&dst)[2][4], uint64_t gg) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            dst[0][d] = load_nt(t0, gg + (uint64_t)c * q16 + (uint64_t)d * q4);
            dst[1][d] = load_nt(t1, gg + (uint64_t)c * q16 + (uint64_t)d * q4);
        }
    };
    if (g < groups) load(cur, g);
    while (g < groups) {
        const uint64_t gn = g + stride;
        if (PREFETCH && gn < groups) load(nxt, gn);
        Fe x[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            // fold at r_a: (d, d + 2) are N/2 apart; then at r_b: the two results are N/4 apart in the half table
            const Fe a0 = fe_sub(cur[f][0], fe_mul29(fe_sub(cur[f][0], cur[f][2], P), ra, P), P);
            const Fe a1 = fe_sub(cur[f][1], fe_mul29(fe_sub(cur[f][1], cur[f][3], P), ra, P), P);
            x[f] = fe_sub(a0, fe_mul29(fe_sub(a0, a1, P), rb, P), P);
            store_wt(f == 0 ? o0 : o1, g + (uint64_t)c * q16, x[f]);
        }
        // this lane's node values of both factors for t = 0, 1, 2 -> three products
        Fe v0[2], v1[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const Fe b0 = quad_bcast<0>(x[f]), b1 = quad_bcast<1>(x[f]), b2 = quad_bcast<2>(x[f]), b3 = quad_bcast<3>(x[f]);
            v0[f] = node_value(b0, b2, m, P);   // U_f[0] at this lane's node
            v1[f] = node_value(b1, b3, m, P);   // U_f[1]
        }
        const Fe d0 = fe_sub(v1[0], v0[0], P), d1 = fe_sub(v1[1], v0[1], P);
        wide_mac(acc[0], v0[0].v, v0[1].v);
        wide_mac(acc[1], v1[0].v, v1[1].v);
        const Fe w0 = fe_add(v1[0], d0, P), w1 = fe_add(v1[1], d1, P);
        wide_mac(acc[2], w0.v, w1.v);
        if (PREFETCH) {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int d = 0; d < 4; ++d) cur[f][d] = nxt[f][d];
        } else if (gn < groups) {
            load(cur, gn);
        }
        g = gn;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        Fe e = redc_wide(acc[t], P);
        Fe a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto xx = __builtin_amdgcn_permlane32_swap(e.v[i], e.v[i], false, false);
            a.v[i] = xx[0];
            b.v[i] = xx[1];
        }
        e = fe_add(a, b, P);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto xx = __builtin_amdgcn_permlane16_swap(e.v[i], e.v[i], false, false);
            a.v[i] = xx[0];
            b.v[i] = xx[1];
        }
        e = fe_add(a, b, P);
        e = fe_add(e, fe_dpp<0x128>(e), P);
        e = fe_add(e, fe_dpp<0x12C>(e), P);
        if (lane < 3) redw[wave][t * 3 + lane] = e;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        Fe tot = redw[0][threadIdx.x];
        for (int w = 1; w < kThreads / 64; ++w) tot = fe_add(tot, redw[w][threadIdx.x], P);
        fe_store(partials, (uint64_t)blockIdx.x * 9 + threadIdx.x, tot);
    }
}

// the ENTRY such a schedule needs: round 0 computing E_1(t; rho) (from which S_0 follows) instead of the plain S_0 -- nine products per group
// of four elements per factor (pairs (j, j + N/2) of round 0, then (j, j + N/4)) where k_round0_dot29 forms six.  Same quad layout, no fold.
__global__ __launch_bounds__(kThreads, 2) void k_round0_e(const uint64_t *__restrict__ t0, const uint64_t *__restrict__ t1, uint64_t n, FieldParams P,
                                                          uint64_t *__restrict__ partials) {
    __shared__ Fe redw[kThreads / 64][12];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 3;
    NodeMasks m;
    m.u = c == 0 ? 0xffffffffu : 0u;
    m.v = c == 1 ? 0xffffffffu : 0u;
    m.w = c == 2 ? 0xffffffffu : 0u;
    const uint64_t groups = n >> 2, q4 = n >> 2;
    WideAcc acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) wide_zero(acc[t]);
    const uint64_t stride = (uint64_t)gridDim.x * kQuads;
    uint64_t g = (uint64_t)blockIdx.x * kQuads + (threadIdx.x >> 2);
    Fe cur[2], nxt[2];
    if (g < groups) cur[0] = load_nt(t0, g + (uint64_t)c * q4), cur[1] = load_nt(t1, g + (uint64_t)c * q4);
    while (g < groups) {
        const uint64_t gn = g + stride;
        if (gn < groups) nxt[0] = load_nt(t0, gn + (uint64_t)c * q4), nxt[1] = load_nt(t1, gn + (uint64_t)c * q4);
        Fe v0[2], v1[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            // elements c = 0..3 at j + c N/4: round 0 pairs (0, 2) and (1, 3); the pending fold gives U[0] from (0, 2), U[1] from (1, 3)
            const Fe b0 = quad_bcast<0>(cur[f]), b1 = quad_bcast<1>(cur[f]), b2 = quad_bcast<2>(cur[f]), b3 = quad_bcast<3>(cur[f]);
            v0[f] = node_value(b0, b2, m, P);
            v1[f] = node_value(b1, b3, m, P);
        }
        const Fe d0 = fe_sub(v1[0], v0[0], P), d1 = fe_sub(v1[1], v0[1], P);
        wide_mac(acc[0], v0[0].v, v0[1].v);
        wide_mac(acc[1], v1[0].v, v1[1].v);
        const Fe w0 = fe_add(v1[0], d0, P), w1 = fe_add(v1[1], d1, P);
        wide_mac(acc[2], w0.v, w1.v);
        cur[0] = nxt[0], cur[1] = nxt[1];
        g = gn;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        Fe e = redc_wide(acc[t], P);
        Fe a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto xx = __builtin_amdgcn_permlane32_swap(e.v[i], e.v[i], false, false);
            a.v[i] = xx[0];
            b.v[i] = xx[1];
        }
        e = fe_add(a, b, P);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto xx = __builtin_amdgcn_permlane16_swap(e.v[i], e.v[i], false, false);
            a.v[i] = xx[0];
            b.v[i] = xx[1];
        }
        e = fe_add(a, b, P);
        e = fe_add(e, fe_dpp<0x128>(e), P);
        e = fe_add(e, fe_dpp<0x12C>(e), P);
        if (lane < 3) redw[wave][t * 3 + lane] = e;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        Fe tot = redw[0][threadIdx.x];
        for (int w = 1; w < kThreads / 64; ++w) tot = fe_add(tot, redw[w][threadIdx.x], P);
        fe_store(partials, (uint64_t)blockIdx.x * 9 + threadIdx.x, tot);
    }
}

__global__ void k_fill(uint64_t *t, uint64_t n, uint64_t seed, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t s = seed + i * 0x9E3779B97F4A7C15ull;
        Fe x;
        for (int w = 0; w < 4; ++w) {
            s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
            x.v[2 * w] = (uint32_t)s;
            x.v[2 * w + 1] = (uint32_t)(s >> 32);
        }
        x.v[7] &= 0x0fffffffu;   // < 2^252 < p: a valid (Montgomery-form) element
        fe_store(t, i, x);
    }
}

static Fe host_sum(const std::vector<uint64_t> &part, size_t blocks, size_t per, size_t slot, const FieldParams &P) {
    Fe s = fe_zero();
    for (size_t b = 0; b < blocks; ++b) s = fe_add(s, fe_from_u64limbs(part.data() + (b * per + slot) * 4), P);
    return s;
}

int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 20;
    const uint64_t n = 1ull << log_n;
    const FieldInfo *fi = field_info(0);
    const FieldParams &P = fi->P;
    uint64_t *T[2], *H[2], *Q[2], *O[2], *part, *chal;
    for (int f = 0; f < 2; ++f) {
        CK(hipMalloc(&T[f], n * 32));
        CK(hipMalloc(&H[f], n * 16));
        CK(hipMalloc(&Q[f], n * 8));
        CK(hipMalloc(&O[f], n * 8));
        k_fill<<<2048, 256>>>(T[f], n, 0x5EED + 77 * f, P);
    }
    CK(hipMalloc(&part, 2048 * 9 * 32));
    CK(hipMalloc(&chal, 2 * 128));
    // two challenges: powers of the two-adic root (any field elements do)
    const Fe ra_fe = fe_pow_u64(fi->two_adic_root, 12345, P), rb_fe = fe_pow_u64(fi->two_adic_root, 54321, P);
    const Mul29 ra = mul29_prepare(ra_fe, P), rb = mul29_prepare(rb_fe, P);
    uint32_t rec[2][32] = {};
    for (int i = 0; i < 9; ++i) rec[0][8 + i] = ra.l[i], rec[1][8 + i] = rb.l[i];
    CK(hipMemcpy(chal, rec, sizeof rec, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());

    auto classic = [&]() {
        FactorPtrs fp = {};
        fp.in[0] = T[0], fp.in[1] = T[1], fp.out[0] = H[0], fp.out[1] = H[1];
        uint64_t q = n >> 2;
        uint32_t g = (uint32_t)((q + 256ull * kMaxLazy - 1) / (256ull * kMaxLazy));
        if (g < 512) g = 512;
        k_round_kd<2, 2, true, 0, true, true><<<g, kBlock>>>(fp, q, P, chal, part, ClaimJob{});
        fp.in[0] = H[0], fp.in[1] = H[1], fp.out[0] = Q[0], fp.out[1] = Q[1];
        q = n >> 3;
        g = (uint32_t)((q + 256ull * kMaxLazy - 1) / (256ull * kMaxLazy));
        if (g < 512) g = 512;
        k_round_kd<2, 2, true, 0, true, true><<<g, kBlock>>>(fp, q, P, chal + 16, part, ClaimJob{});
        return g;
    };
    const uint64_t groups = n >> 4;
    uint32_t g2 = (uint32_t)((groups + (uint64_t)kQuads * kMaxLazy - 1) / ((uint64_t)kQuads * kMaxLazy));
    if (g2 < 512) g2 = 512;
    if (g2 > 2048) g2 = 2048;
    if ((uint64_t)g2 * kQuads * kMaxLazy < groups) {
        printf("grid too small for the lazy accumulators at this size\n");
        return 1;
    }
    // ---- correctness
    const uint32_t gc = classic();
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> pc((size_t)gc * 3 * 4), p2((size_t)g2 * 9 * 4), qa((size_t)(n >> 2) * 4), qb((size_t)(n >> 2) * 4);
    CK(hipMemcpy(pc.data(), part, pc.size() * 8, hipMemcpyDeviceToHost));
    k_round2<true><<<g2, kThreads>>>(T[0], T[1], O[0], O[1], n, P, ra, rb, part);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(p2.data(), part, p2.size() * 8, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int f = 0; f < 2; ++f) {
        CK(hipMemcpy(qa.data(), Q[f], qa.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(qb.data(), O[f], qb.size() * 8, hipMemcpyDeviceToHost));
        ok = ok && memcmp(qa.data(), qb.data(), qa.size() * 8) == 0;
    }
    const Fe s0_c = host_sum(pc, gc, 3, 0, P), l_c = host_sum(pc, gc, 3, 2, P);
    const Fe s0_2 = fe_add(host_sum(p2, g2, 9, 0 * 3 + 0, P), host_sum(p2, g2, 9, 1 * 3 + 0, P), P);
    const Fe l_2 = fe_add(host_sum(p2, g2, 9, 0 * 3 + 1, P), host_sum(p2, g2, 9, 1 * 3 + 1, P), P);
    ok = ok && fe_eq(s0_c, s0_2) && fe_eq(l_c, l_2);
    printf("n = 2^%d per factor: quarter tables %s, S(0) and leading coefficient from the nine sums %s the classic second round's\n", log_n,
           ok ? "bit-identical" : "DIFFER", (fe_eq(s0_c, s0_2) && fe_eq(l_c, l_2)) ? "equal" : "DIFFER from");
    if (!ok) return 1;
    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round) {
        float ms_c = 0, ms_p = 0, ms_n = 0;
        classic();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) classic();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_c, e0, e1));
        k_round2<true><<<g2, kThreads>>>(T[0], T[1], O[0], O[1], n, P, ra, rb, part);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) k_round2<true><<<g2, kThreads>>>(T[0], T[1], O[0], O[1], n, P, ra, rb, part);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_p, e0, e1));
        k_round2<false><<<g2, kThreads>>>(T[0], T[1], O[0], O[1], n, P, ra, rb, part);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) k_round2<false><<<g2, kThreads>>>(T[0], T[1], O[0], O[1], n, P, ra, rb, part);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_n, e0, e1));
        const double bytes_c = 2.25 * n * 64, bytes_2 = 1.25 * n * 64;
        printf("two classic rounds %8.1f us (%5.2f TB/s on %.2f GB)   two rounds in one pass: prefetch %8.1f us (%5.2f TB/s on %.2f GB, x%.3f)   no prefetch %8.1f us (x%.3f)\n",
               ms_c / reps * 1e3, bytes_c / (ms_c / reps * 1e-3) / 1e12, bytes_c / 1e9, ms_p / reps * 1e3, bytes_2 / (ms_p / reps * 1e-3) / 1e12, bytes_2 / 1e9,
               ms_c / ms_p, ms_n / reps * 1e3, ms_c / ms_n);
    }
    // ---- what the schedule's ENTRY costs: round 0 in the pending-challenge form against the shipped round-0 kernel (sums S_0 only)
    {
        FactorPtrs fp = {};
        fp.in[0] = T[0], fp.in[1] = T[1];
        const uint64_t q = n >> 1;
        uint32_t g0 = (uint32_t)((q + 256ull * kMaxLazy - 1) / (256ull * kMaxLazy));
        if (g0 < 512) g0 = 512;
        uint32_t ge = 2048;   // groups = n / 4 = 2^22 at n = 2^24: 2048 blocks x 64 quads x 32 groups per lane
        if ((uint64_t)ge * kQuads * kMaxLazy < (n >> 2)) {
            printf("entry kernel: size too large for one launch of lazy accumulators\n");
            return 0;
        }
        // S_0(0) from the entry kernel's sums must equal the shipped round-0 kernel's
        k_round0_dot29<0><<<g0, kBlock>>>(fp, q, P, part);
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> p0((size_t)g0 * 3 * 4), pe((size_t)ge * 9 * 4);
        CK(hipMemcpy(p0.data(), part, p0.size() * 8, hipMemcpyDeviceToHost));
        k_round0_e<<<ge, kThreads>>>(T[0], T[1], n, P, part);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(pe.data(), part, pe.size() * 8, hipMemcpyDeviceToHost));
        const bool same = fe_eq(host_sum(p0, g0, 3, 0, P), fe_add(host_sum(pe, ge, 9, 0, P), host_sum(pe, ge, 9, 3, P), P)) &&
                          fe_eq(host_sum(p0, g0, 3, 1, P), fe_add(host_sum(pe, ge, 9, 2, P), host_sum(pe, ge, 9, 5, P), P));
        printf("entry: S_0(0), S_0(1) from E_1's sums %s the shipped round-0 kernel's\n", same ? "equal" : "DIFFER from");
        for (int round = 0; round < 3; ++round) {
            float ms_0 = 0, ms_e = 0;
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) k_round0_dot29<0><<<g0, kBlock>>>(fp, q, P, part);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_0, e0, e1));
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) k_round0_e<<<ge, kThreads>>>(T[0], T[1], n, P, part);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_e, e0, e1));
            printf("round 0: shipped k_round0_dot29 (S_0: 6 products per 4 elements per factor) %8.1f us   pending-challenge entry (E_1: 9 products) %8.1f us (+%.1f us)\n",
                   ms_0 / reps * 1e3, ms_e / reps * 1e3, (ms_e - ms_0) / reps * 1e3);
        }
    }
    return 0;
}
