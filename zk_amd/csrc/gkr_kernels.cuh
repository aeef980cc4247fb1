// gkr_kernels.cuh -- device side of the GKR-shaped driver (the reference has NO gkr crate, so there is no
// reference file to cite; the building block it would call is prove_partial, sumcheck/src/prover.rs:24-30, whose intent
// for GKR is noted at polynomial/src/multilinear/evaluation_form.rs:45-48).  Protocol and formats are ours: DESIGN.md 10.
//
// A layer is 2^s_out fan-in-2 gates over the 2^s_in values of the layer below: gate z = (op, left x, right y),
// op 0 = add, 1 = mul.  All tables use the MLE index convention of the reference (variable 0 = index MSB).
// Bookkeeping tables for the two-phase (Libra-style) layer sumcheck, E = alpha*eq(g1,.) + beta*eq(g2,.):
//   phase 1 (over x):  H [x] = sum_{add (z,x,y)} E[z] + sum_{mul (z,x,y)} E[z]*W[y],   B1[x] = sum_{add (z,x,y)} E[z]*W[y]
//                      layer polynomial  P1(x) = W(x)*H(x) + B1(x)
//   phase 2 (over y):  A2[y] = sum_{add (z,x,y)} E[z]*eq_u[x],  M2[y] = sum_{mul (z,x,y)} E[z]*eq_u[x]
//                      H2 = A2 + W(u)*M2,  C2 = W(u)*A2,   P2(y) = W(y)*H2(y) + C2(y)
// Rows are walked through a CSR index of the gate list (by left input / by right input) built once at upload, whose
// entries carry the gate's output index, other input and op: no atomics on 256-bit values, one thread per row.
#pragma once
#include "common.cuh"
#include "transcript.cuh"

namespace zk {

// eq(point, .) over nv variables by direct products (small tables only: nv muls per entry):
// out[j] = scale * prod_w (bit_{nv-1-w}(j) ? g_w : 1 - g_w)
__global__ __launch_bounds__(kBlock) void k_eq_direct(const uint64_t *__restrict__ point, uint32_t nv, Fe scale,
                                                      uint64_t *__restrict__ out, FieldParams P, const uint64_t *__restrict__ d_scale = nullptr) {
    if (d_scale) scale = fe_load(d_scale, 0);   // the scale lives on the device (the GKR driver's alpha / beta)
    const uint64_t n = 1ull << nv, stride = (uint64_t)gridDim.x * kBlock;
    Fe one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        Fe acc = scale;
        for (uint32_t w = 0; w < nv; ++w) {
            const Fe g = fe_load(point, w);
            const bool bit = (j >> (nv - 1 - w)) & 1;
            acc = fe_mul(acc, bit ? g : fe_sub(one, g, P), P);
        }
        fe_store(out, j, acc);
    }
}
// out[idx] (+)= hi[idx >> lo_bits] * lo[idx & mask]: the full table as the outer product of its two halves
template <bool ACCUMULATE>
__global__ __launch_bounds__(kBlock) void k_eq_outer(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo,
                                                     uint32_t lo_bits, uint64_t n, uint64_t *__restrict__ out, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock, mask = (1ull << lo_bits) - 1;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        Fe v = fe_mul(fe_load(hi, j >> lo_bits), fe_load(lo, j & mask), P);
        if (ACCUMULATE) v = fe_add(v, fe_load(out, j), P);
        fe_store(out, j, v);
    }
}

// The two half tables of eq(point, .) for m <= 30 in one launch: workgroup 0 builds hi (the first m/2 variables, scale
// folded in), workgroup 1 builds lo.  A half of nv <= 15 variables is itself the outer product of two quarter tables that
// are built by direct products in LDS (<= 8 dependent multiplies), so the whole chain is <= 9 multiplies deep; the point
// is read from DEVICE memory once.  k_eq_outer then needs ONE multiply per element of the full table.
// With a second point (point2 != null, grid = 4) workgroups 2, 3 build its halves into d_hi2 / d_lo2 with *d_scale2.

// ---- eq tables that are never written out (the GKR kernels' operands) ----------------------------------------------------------
constexpr uint32_t kEqLoWords = 12;   // a prepared multiplier: nine 29-bit limbs, padded to three 16-byte loads
ZK_D void store_eq_lo(uint32_t *q, const Mul29 &m) {
    uint4 *o = reinterpret_cast<uint4 *>(q);
    o[0] = make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]);
    o[1] = make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]);
    o[2] = make_uint4(m.l[8], 0, 0, 0);
}
ZK_D Mul29 load_eq_lo(const uint32_t *q) {
    const uint4 *o = reinterpret_cast<const uint4 *>(q);
    const uint4 a = o[0], b = o[1], c = o[2];
    Mul29 m = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x}};
    return m;
}
__global__ __launch_bounds__(kBlock) void k_eq_halves(const uint64_t *__restrict__ point, uint32_t m, Fe scale,
                                                      uint64_t *__restrict__ d_hi, uint64_t *__restrict__ d_lo, FieldParams P,
                                                      const uint64_t *__restrict__ d_scale = nullptr, const uint64_t *__restrict__ point2 = nullptr,
                                                      const uint64_t *__restrict__ d_scale2 = nullptr, uint64_t *__restrict__ d_hi2 = nullptr,
                                                      uint64_t *__restrict__ d_lo2 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char eq_smem[];
    if (blockIdx.x >= 2) {   // (block-uniform)
        point = point2;
        d_scale = d_scale2;
        d_hi = d_hi2;
        d_lo = d_lo2;
    }
    if (d_scale) scale = fe_load(d_scale, 0);
    const uint32_t hi_bits = m / 2, lo_bits = m - hi_bits;
    const bool is_hi = (blockIdx.x & 1) == 0;
    const uint32_t nv = is_hi ? hi_bits : lo_bits, first = is_hi ? 0 : hi_bits;
    const uint32_t a = nv / 2, b = nv - a, na = 1u << a, nb = 1u << b;
    uint64_t *qa = reinterpret_cast<uint64_t *>(eq_smem), *qb = qa + (size_t)na * 4, *pt = qb + (size_t)nb * 4;
    Fe one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
    if (threadIdx.x < nv) {   // (1 - g, g) pairs of this half's variables
        const Fe g = fe_load(point, first + threadIdx.x);
        fe_store(pt, 2 * threadIdx.x, fe_sub(one, g, P));
        fe_store(pt, 2 * threadIdx.x + 1, g);
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < na + nb; e += kBlock) {
        const bool in_a = e < na;
        const uint32_t j = in_a ? e : e - na, cnt = in_a ? a : b, off = in_a ? 0 : a;
        Fe acc = (in_a && is_hi) ? scale : one;
        for (uint32_t w = 0; w < cnt; ++w) acc = fe_mul(acc, fe_load(pt, 2 * (off + w) + ((j >> (cnt - 1 - w)) & 1)), P);
        fe_store(in_a ? qa : qb, j, acc);
    }
    __syncthreads();
    uint64_t *dst = is_hi ? d_hi : d_lo;
    for (uint32_t i = threadIdx.x; i < (1u << nv); i += kBlock) fe_store(dst, i, fe_mul(fe_load(qa, i >> b), fe_load(qb, i & (nb - 1)), P));
}

// An eq table that is never written out: v[i] = hi[i >> lo_bits] * lo[i & mask] (+ hi2[..] * lo2[..] for alpha*eq(g1,.) +
// beta*eq(g2,.); the scales sit in the hi halves), read where it is needed from the two halves k_eq_halves leaves -- 2^(m/2)
// entries each, a few dozen KB that stay in L2 -- at the price of one multiplication (one and a half for two points: the lo
// halves are stored as prepared 29-bit multipliers, so both products share one reduction, fe_dot2_29).  The GKR kernels gather
// E[z], eq_u[x], eq_v[y] at random indices: out of a 32-MiB table that is a 32-byte read from the memory side per gate (what
// bounded those kernels); out of the halves it is two cache hits.  Exact field arithmetic: the same values as the table's.
struct EqFactor {
    const uint64_t *hi, *hi2;    // hi2 == nullptr: one point
    const uint32_t *lo, *lo2;    // prepared multipliers (k_eq_halves lo_prepared), kEqLoWords words per entry
    uint32_t lo_bits;
};
ZK_D Fe eq_factor_at(const EqFactor &f, uint32_t i, const FieldParams &P) {
    const uint32_t h = i >> f.lo_bits, l = i & ((1u << f.lo_bits) - 1);
    const Fe a = fe_load(f.hi, h);
    const Mul29 c = load_eq_lo(f.lo + (size_t)l * kEqLoWords);
    if (!f.hi2) return fe_mul29(a, c, P);   // kernel-uniform branch
    return fe_dot2_29(a, c, fe_load(f.hi2, h), load_eq_lo(f.lo2 + (size_t)l * kEqLoWords), P);   // both products, ONE reduction
}

// Phase 2's eq_u: its EqFactor carries the SAME point twice, the second hi half scaled by W(u) (free when the halves are built).
// A mul gate's term is wanted times W(u) (H2 = a + W(u) m), an add gate's plain, so the gate type picks the hi half and the row
// is closed with one multiplication (C2 = W(u) a) instead of two.
ZK_D Fe eq_u_at(const EqFactor &f, uint32_t i, bool scaled, const FieldParams &P) {
    const uint32_t h = i >> f.lo_bits, l = i & ((1u << f.lo_bits) - 1);
    return fe_mul29(fe_load(scaled ? f.hi2 : f.hi, h), load_eq_lo(f.lo + (size_t)l * kEqLoWords), P);
}
// The two halves EqFactor reads, for one or two points in one launch: hi = eq over the first m - lo_bits variables with *scale
// folded in (plain elements), lo = eq over the last lo_bits variables (prepared multipliers).  The split is NOT down the middle:
// both halves are indexed at random, so the lo half -- the wider records, three loads each -- is kept small enough to sit in a
// CU's L1 (2^7 entries x 48 B = 6 KiB per table; the hi half of a 2^20 table is then 2^13 elements = 256 KiB, L2-resident).
// Against a 10 + 10 split: k_gkr_phase<1> 56.1 -> 52.7 us, <2> 79.5 -> 77.8 us (the kernels are VALU-bound by then, see the PMC
// counters in profiles/r03_gkr_pmc.log).  grid = points * (bh + bl) workgroups, 1024 entries each; every workgroup rebuilds the
// two quarter tables of its half in LDS (direct products, <= 8 multiplications deep; <= 16 variables per half, so <= 2^8
// entries per quarter).
struct EqSplitJob {
    const uint64_t *point, *scale;   // scale: device scalar or null (= 1)
    uint64_t *hi;
    uint32_t *lo;
};
static inline uint32_t eq_split_lo_bits(uint64_t m) { return (uint32_t)(m <= 7 ? m : (m > 23 ? m - 16 : 7)); }   // hi <= 16 variables
__global__ __launch_bounds__(kBlock) void k_eq_split(EqSplitJob j0, EqSplitJob j1, uint32_t m, uint32_t lo_bits, uint32_t bh, uint32_t bl,
                                                     FieldParams P) {
    __shared__ uint64_t qa[256 * 4], qb[256 * 4], pt[2 * 16 * 4];
    const uint32_t per = bh + bl, slot = blockIdx.x % per;
    const EqSplitJob job = blockIdx.x / per ? j1 : j0;   // (block-uniform)
    const bool is_hi = slot < bh;
    const uint32_t slice = is_hi ? slot : slot - bh;
    const uint32_t nv = is_hi ? m - lo_bits : lo_bits, first = is_hi ? 0 : m - lo_bits;
    const uint32_t a = nv / 2, b = nv - a, na = 1u << a, nb = 1u << b;
    Fe one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
    if (threadIdx.x < nv) {   // (1 - g, g) pairs of this half's variables
        const Fe g = fe_load(job.point, first + threadIdx.x);
        fe_store(pt, 2 * threadIdx.x, fe_sub(one, g, P));
        fe_store(pt, 2 * threadIdx.x + 1, g);
    }
    __syncthreads();
    const Fe scale = (is_hi && job.scale) ? fe_load(job.scale, 0) : one;
    for (uint32_t e = threadIdx.x; e < na + nb; e += kBlock) {
        const bool in_a = e < na;
        const uint32_t j = in_a ? e : e - na, cnt = in_a ? a : b, off = in_a ? 0 : a;
        Fe acc = in_a ? scale : one;
        for (uint32_t w = 0; w < cnt; ++w) acc = fe_mul(acc, fe_load(pt, 2 * (off + w) + ((j >> (cnt - 1 - w)) & 1)), P);
        fe_store(in_a ? qa : qb, j, acc);
    }
    __syncthreads();
    const uint32_t end = (slice + 1) * 1024 < (1u << nv) ? (slice + 1) * 1024 : (1u << nv);
    for (uint32_t i = slice * 1024 + threadIdx.x; i < end; i += kBlock) {
        const Fe v = fe_mul(fe_load(qa, i >> b), fe_load(qb, i & (nb - 1)), P);
        if (is_hi) fe_store(job.hi, i, v);
        else store_eq_lo(job.lo + (size_t)i * kEqLoWords, mul29_prepare(v, P));
    }
}

// layer evaluation: out[z] = W[left[z]] (+|*) W[right[z]]
__global__ __launch_bounds__(kBlock) void k_gkr_forward(const uint8_t *__restrict__ op, const uint32_t *__restrict__ left,
                                                        const uint32_t *__restrict__ right, const uint64_t *__restrict__ W,
                                                        uint64_t n_gates, uint64_t *__restrict__ out, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t z = (uint64_t)blockIdx.x * kBlock + threadIdx.x; z < n_gates; z += stride) {
        const Fe a = fe_load(W, left[z]), b = fe_load(W, right[z]);
        fe_store(out, z, op[z] ? fe_mul(a, b, P) : fe_add(a, b, P));
    }
}

// Rows longer than this (an input wire that feeds very many gates) are left to a whole workgroup each: a single thread
// walking 2^20 entries would take a second.
constexpr uint32_t kGkrHeavyRow = 256;
// two field sums per thread -> workgroup totals in sum[] of thread 0
ZK_D void gkr_block_sum2(Fe (&sum)[2], const FieldParams &P) {
    __shared__ uint32_t red2[kBlock / 64][2][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        sum[t] = fe_wave_sum(sum[t], P);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) red2[wave][t][i] = sum[t].v[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Fe acc = fe_zero();
            for (int w = 0; w < kBlock / 64; ++w) {
                Fe o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o.v[i] = red2[w][t][i];
                acc = fe_add(acc, o, P);
            }
            sum[t] = acc;
        }
    }
}
// heavy rows of phase 1 / phase 2: workgroup b takes row heavy[b]
__global__ __launch_bounds__(kBlock) void k_gkr_phase1_heavy(const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ lptr,
                                                             const uint2 *__restrict__ lent, EqFactor E,
                                                             const uint64_t *__restrict__ W, uint64_t *__restrict__ H,
                                                             uint64_t *__restrict__ B1, FieldParams P) {
    const uint32_t x = heavy[blockIdx.x];
    Fe s[2] = {fe_zero(), fe_zero()};   // h, b
    for (uint32_t e = lptr[x] + threadIdx.x; e < lptr[x + 1]; e += kBlock) {
        const uint2 ent = lent[e];
        const Fe ez = eq_factor_at(E, ent.x, P), t = fe_mul(ez, fe_load(W, ent.y & 0x7FFFFFFFu), P);
        if (ent.y >> 31) {
            s[0] = fe_add(s[0], t, P);
        } else {
            s[0] = fe_add(s[0], ez, P);
            s[1] = fe_add(s[1], t, P);
        }
    }
    gkr_block_sum2(s, P);
    if (threadIdx.x == 0) {
        fe_store(H, x, s[0]);
        fe_store(B1, x, s[1]);
    }
}
__global__ __launch_bounds__(kBlock) void k_gkr_phase2_heavy(const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ rptr,
                                                             const uint2 *__restrict__ rent, EqFactor E,
                                                             EqFactor eq_u, const uint64_t *__restrict__ wu,
                                                             uint64_t *__restrict__ H2, uint64_t *__restrict__ C2, FieldParams P) {
    const uint32_t y = heavy[blockIdx.x];
    Fe s[2] = {fe_zero(), fe_zero()};   // a, m
    for (uint32_t e = rptr[y] + threadIdx.x; e < rptr[y + 1]; e += kBlock) {
        const uint2 ent = rent[e];
        const Fe t = fe_mul(eq_factor_at(E, ent.x, P), eq_u_at(eq_u, ent.y & 0x7FFFFFFFu, ent.y >> 31, P), P);   // mul gates: times W(u)
        if (ent.y >> 31) s[1] = fe_add(s[1], t, P);
        else s[0] = fe_add(s[0], t, P);
    }
    gkr_block_sum2(s, P);
    if (threadIdx.x == 0) {
        fe_store(H2, y, fe_add(s[0], s[1], P));
        fe_store(C2, y, fe_mul(fe_load(wu, 0), s[0], P));
    }
}

// Bookkeeping tables of the two sumchecks of a layer (Libra's phase 1 / phase 2) from the CSR of the wiring by left / right
// input.  An entry is {z, other | op << 31}: the gate's output index and its OTHER input, so a row needs no second indirection
// through the gate arrays -- the only random accesses are the two 32-byte elements E[z] and T[other].
//   PHASE 1 (rows x, T = W):    H[x]  = sum_mul E[z] W[y] + sum_add E[z],   B1[x] = sum_add E[z] W[y]
//   PHASE 2 (rows y, T = eq_u): a = sum_add E[z] eq_u[x], m = sum_mul E[z] W(u) eq_u[x]:  H2[y] = a + m,  C2[y] = W(u) a
// ENTRY-parallel: a workgroup owns kGkrRows consecutive rows and walks THEIR entries one per thread (coalesced entry reads, one
// pair of gathers and one multiplication per lane, all lanes busy), parks the two addends of every entry in LDS, and the row's
// thread adds up its own (LDS reads and modular additions only).  The row-parallel form ran as many gather rounds per wave as
// its longest row (4-5 with random wiring, where the row lengths are Poisson(1)); this one runs one.  kGkrRows = 224: the
// entries of 224 rows of a random wiring (224 +- 15) fit one 256-entry chunk 98 % of the time.  Rows longer than kGkrHeavyRow
// are empty in the CSR this kernel is given (k_gkr_phase*_heavy, launched after it, owns their outputs).
constexpr uint32_t kGkrRows = 224;
static inline uint32_t gkr_phase_grid(uint64_t n_rows) { return (uint32_t)((n_rows + kGkrRows - 1) / kGkrRows); }
template <int PHASE>
__global__ __launch_bounds__(kBlock) void k_gkr_phase(const uint32_t *__restrict__ ptr, const uint2 *__restrict__ ent, EqFactor E,
                                                      const uint64_t *__restrict__ W /* PHASE 1 */, EqFactor eq_u /* PHASE 2 */,
                                                      const uint64_t *__restrict__ wu, uint64_t n_rows, uint64_t *__restrict__ out0,
                                                      uint64_t *__restrict__ out1, FieldParams P) {
    __shared__ uint32_t park[2][8][kBlock];   // [addend][word][entry of the chunk]
    const uint64_t x0 = (uint64_t)blockIdx.x * kGkrRows;
    const uint64_t x_end = x0 + kGkrRows < n_rows ? x0 + kGkrRows : n_rows;
    const uint64_t x = x0 + threadIdx.x;
    const bool row = threadIdx.x < kGkrRows && x < n_rows;
    const uint32_t rs = row ? ptr[x] : 0, re = row ? ptr[x + 1] : 0;
    const uint32_t e_lo = ptr[x0], e_hi = ptr[x_end];
    Fe acc0 = fe_zero(), acc1 = fe_zero();
    for (uint32_t c0 = e_lo; c0 < e_hi; c0 += kBlock) {   // block-uniform trip count
        const uint32_t e = c0 + threadIdx.x;
        if (e < e_hi) {
            const uint2 en = ent[e];
            const Fe ez = eq_factor_at(E, en.x, P);
            const Fe t = fe_mul(ez, PHASE == 1 ? fe_load(W, en.y & 0x7FFFFFFFu) : eq_u_at(eq_u, en.y & 0x7FFFFFFFu, en.y >> 31, P), P);
            const uint32_t mul = 0u - (en.y >> 31);   // all ones for a mul gate
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PHASE == 1) {   // {h, b} += mul ? {t, 0} : {ez, t}
                    park[0][i][threadIdx.x] = (t.v[i] & mul) | (ez.v[i] & ~mul);
                    park[1][i][threadIdx.x] = t.v[i] & ~mul;
                } else {            // {a, m} += mul ? {0, t} : {t, 0}
                    park[0][i][threadIdx.x] = t.v[i] & ~mul;
                    park[1][i][threadIdx.x] = t.v[i] & mul;
                }
            }
        }
        __syncthreads();
        const uint32_t lo = rs > c0 ? rs : c0, hi = re < c0 + kBlock ? re : c0 + kBlock;
        for (uint32_t q = lo; q < hi; ++q) {
            Fe v0, v1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v0.v[i] = park[0][i][q - c0];
                v1.v[i] = park[1][i][q - c0];
            }
            acc0 = fe_add(acc0, v0, P);
            acc1 = fe_add(acc1, v1, P);
        }
        __syncthreads();
    }
    if (!row) return;
    if (PHASE == 1) {
        fe_store(out0, x, acc0);
        fe_store(out1, x, acc1);
    } else {
        fe_store(out0, x, fe_add(acc0, acc1, P));   // the mul gates' terms came multiplied by W(u) (eq_u_at)
        fe_store(out1, x, fe_mul(fe_load(wu, 0), acc0, P));
    }
}

// verifier side: the wiring predicates of one layer at (u, v),
//   add~E(u,v) = sum_{add (z,x,y)} E[z]*eq_u[x]*eq_v[y],  mul~E(u,v) = the same over mul gates
// -> per-block partials [block][2] (reduced by k_round_tail with ns = 2)
__global__ __launch_bounds__(kBlock) void k_gkr_wiring_eval(const uint8_t *__restrict__ op, const uint32_t *__restrict__ left,
                                                            const uint32_t *__restrict__ right, EqFactor E, EqFactor eq_u, EqFactor eq_v,
                                                            uint64_t n_gates, uint64_t *__restrict__ partials, FieldParams P) {
    __shared__ uint32_t red[kBlock / 64][2][8];
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    Fe sum[2] = {fe_zero(), fe_zero()};
    for (uint64_t z = (uint64_t)blockIdx.x * kBlock + threadIdx.x; z < n_gates; z += stride) {
        const Fe t = fe_mul(fe_mul(eq_factor_at(E, (uint32_t)z, P), eq_factor_at(eq_u, left[z], P), P), eq_factor_at(eq_v, right[z], P), P);
        if (op[z]) sum[1] = fe_add(sum[1], t, P);
        else sum[0] = fe_add(sum[0], t, P);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        sum[t] = fe_wave_sum(sum[t], P);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) red[wave][t][i] = sum[t].v[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        Fe acc = fe_zero();
        for (int w = 0; w < kBlock / 64; ++w) {
            Fe o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = red[w][threadIdx.x][i];
            acc = fe_add(acc, o, P);
        }
        fe_store(partials, (uint64_t)blockIdx.x * 2 + threadIdx.x, acc);
    }
}

// The factors of a sum-of-products sumcheck at the challenge point: the fold after the LAST round (prover.rs:64), which
// the reference computes and drops; a layered driver needs it (W(u), W(v)).  tables hold 2 elements each.
__global__ void k_final_evals(FactorPtrs fp, uint32_t k, const uint64_t *__restrict__ d_challenge, uint64_t *__restrict__ out,
                              FieldParams P) {
    const uint32_t f = threadIdx.x;
    if (f >= k) return;
    const Fe r = fe_load(d_challenge, 0), lo = fe_load(fp.in[f], 0), hi = fe_load(fp.in[f], 1);
    fe_store(out, f, fe_sub(lo, fe_mul(r, fe_sub(lo, hi, P), P), P));
}

// ---- pieces of the multi-GPU four-step NTT (the fft crate's transform, fft/src/lib.rs:21-46, across
// ranks).  t[j] *= scale * base^j: lane-strided geometric sequence, one exponentiation per thread then 2 multiplies per
// element (the inter-rank twiddles omega_N^(rank*j)).
__global__ __launch_bounds__(kBlock) void k_mul_powers(uint64_t *__restrict__ t, uint64_t n, Fe base, Fe scale, Fe step /* base^stride */,
                                                       FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock, first = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (first >= n) return;
    Fe cur = scale, b = base;   // cur = scale * base^first by square-and-multiply
    for (uint64_t e = first; e; e >>= 1) {
        if (e & 1) cur = fe_mul(cur, b, P);
        b = fe_mul(b, b, P);
    }
    for (uint64_t j = first; j < n; j += stride) {
        fe_store(t, j, fe_mul(fe_load(t, j), cur, P));
        cur = fe_mul(cur, step, P);
    }
}
// out[k][j] = sum_r w^(r*k) in[r][j], r, k in [0, W): the W-point transforms ACROSS the ranks' rows, one thread per column j.
// pw: W elements, pw[e] = w^e.
__global__ __launch_bounds__(kBlock) void k_dft_across(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t W,
                                                       uint64_t L, const uint64_t *__restrict__ pw, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < L; j += stride) {
        for (uint32_t k = 0; k < W; ++k) {
            Fe acc = fe_load(in, j);   // r = 0
            for (uint32_t r = 1; r < W; ++r) acc = fe_add(acc, fe_mul(fe_load(in, (uint64_t)r * L + j), fe_load(pw, (r * k) & (W - 1)), P), P);
            fe_store(out, (uint64_t)k * L + j, acc);
        }
    }
}

// ---- the driver's transcript on the device (one wave per step) ------------------------------------------------------------------
// The GKR driver keeps ONE sponge for the whole proof: every sumcheck continues it, and the few
// scalar steps between two sumchecks run here, so that a proof needs no host synchronisation between its first and its last
// kernel.  sc = the proof's scalar block in device memory: [0] current claim, [1] alpha, [2] beta, [3] claim of sumcheck #2.
constexpr int kGkrScClaim = 0, kGkrScAlpha = 1, kGkrScBeta = 2, kGkrScSum2 = 3, kGkrScalars = 4;
// absorb the claim of the first sumcheck (prover.rs:42 of that sumcheck)
__global__ __launch_bounds__(64) void k_gkr_chain_start(WordSponge *__restrict__ gsp, const uint64_t *__restrict__ sc, FieldParams P) {
    __shared__ Fe buf[2];
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = lane_sponge_load(gsp, L);
    if (L.lane == 0) buf[0] = fe_load(sc, kGkrScClaim);
    lane_absorb_elems(sp, L, buf, 1, P);
    lane_sponge_store(gsp, sp, L);
}
// after sumcheck #1 of a layer: fin = [W(u), H(u), B1(u)].  proof gets W(u); absorb W(u); the claim of #2 is P1(u) = W(u) H(u) +
// B1(u), absorbed as that sumcheck's first message
__global__ __launch_bounds__(64) void k_gkr_chain_mid(WordSponge *__restrict__ gsp, const uint64_t *__restrict__ fin, uint64_t *__restrict__ proof_wu,
                                                       uint64_t *__restrict__ sc, FieldParams P) {
    __shared__ Fe buf[2];
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = lane_sponge_load(gsp, L);
    if (L.lane == 0) {
        const Fe wu = fe_load(fin, 0), hu = fe_load(fin, 1), bu = fe_load(fin, 2);
        const Fe s2 = fe_add(fe_mul(wu, hu, P), bu, P);
        buf[0] = wu;
        buf[1] = s2;
        fe_store(proof_wu, 0, wu);
        fe_store(sc, kGkrScSum2, s2);
    }
    lane_absorb_elems(sp, L, buf, 2, P);
    lane_sponge_store(gsp, sp, L);
}
// after sumcheck #2: fin = [W(v), ..].  proof gets W(v); absorb W(v); draw alpha, beta; next claim alpha W(u) + beta W(v), absorbed
// as the next layer's first message (after the last layer nobody reads the transcript again)
__global__ __launch_bounds__(64) void k_gkr_chain_end(WordSponge *__restrict__ gsp, const uint64_t *__restrict__ fin, const uint64_t *__restrict__ proof_wu,
                                                       uint64_t *__restrict__ proof_wv, uint64_t *__restrict__ sc, FieldParams P) {
    __shared__ Fe buf[2];
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = lane_sponge_load(gsp, L);
    const Fe wu = fe_load(proof_wu, 0), wv = fe_load(fin, 0);
    if (L.lane == 0) {
        buf[0] = wv;
        fe_store(proof_wv, 0, wv);
    }
    lane_absorb_elems(sp, L, buf, 1, P);
    Mul29 t29;
    const Fe alpha = lane_squeeze(sp, L, P, t29), beta = lane_squeeze(sp, L, P, t29);
    const Fe claim = fe_add(fe_mul(alpha, wu, P), fe_mul(beta, wv, P), P);
    if (L.lane == 0) {
        buf[1] = claim;
        fe_store(sc, kGkrScAlpha, alpha);
        fe_store(sc, kGkrScBeta, beta);
        fe_store(sc, kGkrScClaim, claim);
    }
    lane_absorb_elems(sp, L, buf + 1, 1, P);
    lane_sponge_store(gsp, sp, L);
}

// ---- statement digest: Keccak-256 tree hash (the driver binds circuit, inputs and outputs before the output point is drawn) ----
// digest(data) = 128-byte leaves hashed one per thread, then 4-ary nodes Keccak256(child digests) level by level
// (oracle/gkr_ref.py: tree_digest).  One sponge over 2^20 elements would be serial and cost more than the whole proof; a
// message of <= 128 bytes is ONE permutation (rate 136), so a level is one permutation per thread.
ZK_D void tree_keccak_finish(uint64_t (&s)[25], uint32_t len_bytes, uint64_t *__restrict__ out) {
    s[len_bytes >> 3] ^= 0x01ull << (8 * (len_bytes & 7));   // pad10*1 with Keccak's 0x01 domain byte
    s[16] ^= 0x8000000000000000ull;
    keccak_f1600(s);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = s[i];
}
// leaf i = the to_bytes image (32-byte big-endian canonical integers, evaluation_form.rs:97-103) of elements [4i, 4i+4)
__global__ __launch_bounds__(kBlock) void k_tree_leaves_table(const uint64_t *__restrict__ table, uint64_t n_elems, uint64_t n_leaves,
                                                              uint64_t *__restrict__ out, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_leaves; i += stride) {
        uint64_t s[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) s[k] = 0;
        uint32_t len = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (4 * i + e < n_elems) {
                const Fe c = fe_to_canonical(fe_load(table, 4 * i + e), P);
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    s[4 * e + w] = WordSponge::bswap64((uint64_t)c.v[2 * (3 - w)] | ((uint64_t)c.v[2 * (3 - w) + 1] << 32));
                len += 32;
            }
        }
        tree_keccak_finish(s, len, out + 4 * i);
    }
}
// leaf i = bytes [128 i, min(128 i + 128, nbytes)) of a raw device array (gate lists)
__global__ __launch_bounds__(kBlock) void k_tree_leaves_bytes(const uint8_t *__restrict__ data, uint64_t nbytes, uint64_t n_leaves,
                                                              uint64_t *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_leaves; i += stride) {
        uint64_t s[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) s[k] = 0;
        const uint64_t base = 128 * i;
        const uint32_t len = (uint32_t)(nbytes - base < 128 ? nbytes - base : 128);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            uint64_t x = 0;
            for (int b = 0; b < 8; ++b)
                if ((uint32_t)(8 * w + b) < len) x |= (uint64_t)data[base + 8 * w + b] << (8 * b);
            s[w] = x;
        }
        tree_keccak_finish(s, len, out + 4 * i);
    }
}
// node j = Keccak256(digests [4j, min(4j + 4, n_in)) concatenated)
__global__ __launch_bounds__(kBlock) void k_tree_level(const uint64_t *__restrict__ in, uint64_t n_in, uint64_t n_out, uint64_t *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n_out; j += stride) {
        uint64_t s[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) s[k] = 0;
        uint32_t len = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (4 * j + c < n_in) {
#pragma unroll
                for (int w = 0; w < 4; ++w) s[4 * c + w] = in[4 * (4 * j + c) + w];
                len += 32;
            }
        }
        tree_keccak_finish(s, len, out + 4 * j);
    }
}
// The same node function with ONE WAVE per node (lane-parallel Keccak-f[1600], transcript.cuh): a permutation takes ~3 us on a
// wave against ~17 us for a single lane's 64-bit state, so the small upper levels of the tree -- pure latency -- go this way.
__global__ __launch_bounds__(64) void k_tree_level_wave(const uint64_t *__restrict__ in, uint64_t n_in, uint64_t *__restrict__ out) {
    const uint64_t j = blockIdx.x;
    const LaneKeccak L = lane_keccak_init();
    const uint64_t rem = n_in - 4 * j;
    const uint32_t nwords = 4 * (uint32_t)(rem < 4 ? rem : 4);   // message words (8 bytes each)
    uint64_t a = 0;
    if (L.index >= 0 && (uint32_t)L.index < nwords) a = in[16 * j + (uint32_t)L.index];
    if ((uint32_t)L.index == nwords) a ^= 0x01ull;                 // pad10*1, Keccak domain byte
    if (L.index == 16) a ^= 0x8000000000000000ull;
    a = lane_keccak_f1600(a, L);
    if (L.index >= 0 && L.index < 4) out[4 * j + (uint32_t)L.index] = a;
}

}  // namespace zk
