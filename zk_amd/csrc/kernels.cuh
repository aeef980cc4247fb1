// kernels.cuh -- gfx950 kernels for the MLE fold / sumcheck round / serialisation path.
//
// Data layout in HBM: a table is a dense array of 2^m elements, 32 B each (8 x u32, Montgomery), index bit
// (m-1-v) <-> variable v, i.e. variable 0 is the index MSB (polynomial/src/multilinear/pairing_index.rs:61-65).
// The sumcheck fold always consumes variable 0, so a round touches the two contiguous half-streams
// [0, half) and [half, 2*half): every lane reads whole 32-B elements, a wave reads 2 KiB contiguous runs.
//
// All kernels are grid-stride with 256-thread workgroups (4 waves); grids are sized >> 256 CUs by the host.
// These are HBM-streaming integer kernels: no LDS tiling is needed for the fold itself (no reuse), LDS is used
// for the workgroup reductions of the round sums.  No MFMA (256-bit modular integer ops).
#pragma once
#include "common.cuh"
#include "keccak.hpp"
#include "transcript.cuh"

namespace zk {

// pairing_index.rs:16-20 insert_bit(val, index, 0)
ZK_D uint64_t insert_zero_bit(uint64_t val, uint32_t pos) {
    const uint64_t low = val & ((1ull << pos) - 1ull);
    return ((val >> pos) << (pos + 1)) | low;
}

// ---- MultiLinearPolynomial::partial_evaluate, one assignment (evaluation_form.rs:55-70) ----------------------
// out[j] = left - r*(left - right), (left,right) = index_pair(m, initial_var)[j], pos = m-1-initial_var.
// The reference's r==0 / r==1 shortcuts (:61-62) are the same values, so they are not special-cased.
// Out of place (in != out) for general pos; in place is race-free only for the MSB fold (pos = m-1).
__global__ __launch_bounds__(kBlock) void k_fold(const uint64_t *in, uint64_t *out,
                                                 uint64_t pairs, uint32_t pos, FieldParams P, Mul29 r) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < pairs; j += stride) {
        const uint64_t l = insert_zero_bit(j, pos);
        const Fe lo = fe_load(in, l);
        const Fe hi = fe_load(in, l | (1ull << pos));
        const Fe d = fe_sub(lo, hi, P);
        fe_store(out, j, fe_sub(lo, fe_mul29(d, r, P), P));   // r is prepared on the host: mul29_prepare(challenge)
    }
}

// ---- the sumcheck fold (variable 0 = index MSB) as the bandwidth kernel --------------------------------------------
// Same arithmetic as k_fold with pos = m-1, different data movement: a wave moves 64 consecutive elements of each of
// the three streams (lo, hi, out) as two fully coalesced 1-KiB dwordx4 accesses (lane l touches bytes [16l, 16l+16) of
// the run), and lanes 2i / 2i+1 swap halves with one DPP quad_perm so each lane ends up owning one whole element
// (even lane: element i, odd lane: element 32+i).  All accesses are nontemporal: the table is streamed once.
// Measured on MI355X (tools/mb/mb_stream.hip, 2^24): 5.8-5.9 TB/s vs 5.4 TB/s for the 32-B-per-lane form.
// half = 2^(m-1) >= 64 (a multiple of 64).  Out of place or in place (a wave reads its 64 lo/hi elements before
// writing the 64 outputs at the lo positions; no other wave touches them).
__global__ __launch_bounds__(kBlock) void k_fold_msb(const uint64_t *in, uint64_t *out, uint64_t half, FieldParams P, Mul29 r) {
    const uint32_t lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * kBlock) >> 6;
    const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
    uint4 *out4 = reinterpret_cast<uint4 *>(out);
    for (uint64_t e0 = wave * 64; e0 < half; e0 += nwaves * 64) {
        const uint64_t c0 = 2 * e0 + lane;   // element e occupies uint4 slots 2e, 2e+1
        const uint4 la = nt_load16(in4 + c0), lb = nt_load16(in4 + c0 + 64);
        const uint4 ha = nt_load16(in4 + c0 + 2 * half), hb = nt_load16(in4 + c0 + 2 * half + 64);
        const Fe lo = pair_gather(la, lb, odd), hi = pair_gather(ha, hb, odd);
        const Fe o = fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P);
        uint4 oa, ob;
        pair_scatter(o, odd, oa, ob);
        nt_store16(oa, out4 + c0);
        nt_store16(ob, out4 + c0 + 64);
    }
}

// The same data movement for ANY fold position pos >= 6 (partial_evaluate with initial_var > 0, evaluation_form.rs:55-70): the pair
// partner of an element sits 2^pos >= 64 elements away, so 64 consecutive outputs still read two contiguous 64-element runs --
// block b = e / 2^pos of the output reads in[b * 2^(pos+1) + off] and in[b * 2^(pos+1) + 2^pos + off].  k_fold_msb is the case
// of one block.  Out of place only (a later block's inputs are an earlier block's output positions).  Measured (bench rows_2p24):
// the 32-byte-per-lane k_fold reaches 0.62-0.66 of the HBM peak at 2^24, this form the MSB fold's 0.79.
__global__ __launch_bounds__(kBlock) void k_fold_run(const uint64_t *in, uint64_t *out, uint64_t pairs, uint32_t pos, FieldParams P, Mul29 r) {
    const uint32_t lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * kBlock) >> 6;
    const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
    uint4 *out4 = reinterpret_cast<uint4 *>(out);
    const uint64_t span = 1ull << pos;
    for (uint64_t e0 = wave * 64; e0 < pairs; e0 += nwaves * 64) {
        const uint64_t lo0 = ((e0 >> pos) << (pos + 1)) | (e0 & (span - 1));   // insert_zero_bit(e0, pos): first element of the lo run
        const uint64_t cl = 2 * lo0 + lane, co = 2 * e0 + lane;                // element e occupies uint4 slots 2e, 2e+1
        const uint4 la = nt_load16(in4 + cl), lb = nt_load16(in4 + cl + 64);
        const uint4 ha = nt_load16(in4 + cl + 2 * span), hb = nt_load16(in4 + cl + 2 * span + 64);
        const Fe lo = pair_gather(la, lb, odd), hi = pair_gather(ha, hb, odd);
        const Fe o = fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P);
        uint4 oa, ob;
        pair_scatter(o, odd, oa, ob);
        nt_store16(oa, out4 + co);
        nt_store16(ob, out4 + co + 64);
    }
}

// The same data movement for the LOW six fold positions (pos < 6: the pair partner is 1..32 elements away, inside the wave's
// run).  A wave reads 128 consecutive elements as two coalesced 64-element runs A and B (run_load: lane l owns element
// own(l) = (l >> 1) + 32 (l & 1) of each), so a lane holds one element of a pair of A and one of a pair of B; bit pos of own(l)
// says which side.  ONE exchange with the partner lane (own index differing in bit pos) hands the "low" lane the whole A pair
// and the "high" lane the whole B pair (the low lane sends its B element and receives the partner's A element, the high lane
// the other way round), every lane folds one pair, and a second lane permutation puts output own(l) of the wave's 64 outputs
// on lane l for the coalesced store.  Both permutations are ds_bpermute pulls (8 words each): a few LDS-crossbar cycles per
// 6 KiB of HBM traffic.  Out of place only; pairs a multiple of 64.  Until round 5 these positions ran on k_fold (32 bytes
// per lane: 0.63 of the HBM peak at 2^24 against 0.72-0.73 for the run forms).
ZK_D Fe lane_pull(const Fe &v, uint32_t src_lane) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v.v[i]);
    return r;
}
__global__ __launch_bounds__(kBlock) void k_fold_low(const uint64_t *in, uint64_t *out, uint64_t pairs, uint32_t pos, FieldParams P, Mul29 r) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * kBlock) >> 6;
    const uint32_t own = pair_owned(lane);
    const bool high = (own >> pos) & 1u;
    const uint32_t partner_own = own ^ (1u << pos);
    const uint32_t partner = ((partner_own & 31u) << 1) | (partner_own >> 5);          // the lane that owns partner_own
    // this lane computes output q = 32 * high + (own without bit pos); the store wants output own(l) on lane l
    const uint32_t want = own, want_half = want >> 5, want_low = want & 31u;
    const uint32_t src_own = (((want_low >> pos) << (pos + 1)) | (want_low & ((1u << pos) - 1u))) | (want_half << pos);
    const uint32_t src = ((src_own & 31u) << 1) | (src_own >> 5);
    for (uint64_t e0 = wave * 64; e0 < pairs; e0 += nwaves * 64) {
        const Fe a = run_load_nt(in + 4 * (2 * e0), lane), b = run_load_nt(in + 4 * (2 * e0 + 64), lane);
        const Fe got = lane_pull(high ? a : b, partner);
        const Fe lo = high ? got : a, hi = high ? b : got;
        const Fe o = fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P);
        run_store_nt(out + 4 * e0, lane, lane_pull(o, src));
    }
}

// same fold with the challenge read from device memory (produced by the on-device transcript); m = variables of `in`,
// always the MSB fold.  `pairs` = 2^(m-1).
__global__ __launch_bounds__(kBlock) void k_fold_dev(const uint64_t *in, uint64_t *out, uint64_t pairs, uint32_t m,
                                                     FieldParams P, const uint64_t *__restrict__ rptr) {
    (void)m;
    const Mul29 r = load_challenge29(rptr);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < pairs; j += stride) {
        const Fe lo = fe_load(in, j);
        const Fe hi = fe_load(in, j + pairs);
        fe_store(out, j, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
    }
}

// V consecutive MSB folds in one pass (evaluate, evaluation_form.rs:83-89: n folds in a row at KNOWN assignments): a thread
// reads the 2^V elements that share the low index j = idx mod 2^(m-V) (element c = (x0..x_{V-1}) sits at c * n_out + j),
// folds variable 0 over the pairs (c, c + 2^(V-1)), then variable 1, ... exactly in the reference's order, and writes one
// element: 1/V of the launches and (2^V + 1) / (3 * (2^V - 1)) of the traffic of V single folds.  In place is safe
// (index j is only touched by its own thread).
struct FoldChallenges {
    Mul29 r[3];
};
template <int V>
__global__ __launch_bounds__(kBlock) void k_fold_multi(const uint64_t *in, uint64_t *out, uint64_t n_out, FieldParams P, FoldChallenges ch) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n_out; j += stride) {
        Fe x[1 << V];
#pragma unroll
        for (int c = 0; c < (1 << V); ++c) x[c] = fe_load(in, j + (uint64_t)c * n_out);
#pragma unroll
        for (int lvl = 0; lvl < V; ++lvl) {
#pragma unroll
            for (int c = 0; c < (1 << (V - 1 - lvl)); ++c)
                x[c] = fe_sub(x[c], fe_mul29(fe_sub(x[c], x[c + (1 << (V - 1 - lvl))], P), ch.r[lvl], P), P);
        }
        fe_store(out, j, x[0]);
    }
}

// ---- MultiLinearPolynomial::evaluate, bulk (evaluation_form.rs:83-89): the LOW L variables of the table in one pass ---------
// evaluate() folds variable after variable: n launches (or n/3 three-variable ones), each writing a table the next one reads, and
// all but the first are launch latency.  Folding variables is restriction of a multilinear polynomial, so the table after the
// low L variables (index bits 0..L-1) have been assigned is
//   out[g] = sum_{x < 2^L} in[g * 2^L + x] * eq(point_low, x),   eq(., x) = prod_p (bit p of x ? r_p : 1 - r_p)
// -- the same field elements whatever the order of the folds (exact arithmetic, canonical representatives), so the result of
// evaluate() is bit-identical.  One workgroup produces one out[g] from 2^L CONTIGUOUS elements (L = 8..12: 8-128 KiB):
// thread t owns the elements x = xm * 256 + t, all loaded before any arithmetic; waves 0-2 build three 16-entry eq tables (bits
// 0-3, 4-7, 8-L-1; chains of <= 3 multiplications, hidden under the loads); then each thread adds its <= 16 products
// in[.] * eq_m[xm] UNREDUCED (wide_mac), reduces once, multiplies by its own eq_l[t] = eq_47[t >> 4] * eq_03[t & 15] and the
// workgroup sums.  Per element: one 512-bit product and 32 bytes read; the remaining n - L variables are a 2^L times smaller table.
// what is left after a weighted bulk launch: the sum of its n <= 4096 outputs, one workgroup; result (and the completion word)
// straight into pinned host memory when the caller asks for it
__global__ __launch_bounds__(kBlock) void k_eval_sum(const uint64_t *__restrict__ in, uint32_t n, FieldParams P, uint64_t *__restrict__ out,
                                                     volatile uint32_t *flag, uint32_t seq) {
    __shared__ uint32_t red[kBlock / 64][8];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    Fe acc = fe_zero();
    Fe x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (tid + 256u * i < n) x[i] = fe_load(in, tid + 256u * i);   // all loads first: one memory round trip
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (tid + 256u * i < n) acc = fe_add(acc, x[i], P);
    acc = fe_wave_sum(acc, P);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid >> 6][i] = acc.v[i];
    }
    __syncthreads();
    if (tid == 0) {
        for (int wv = 1; wv < kBlock / 64; ++wv) {
            Fe o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = red[wv][i];
            acc = fe_add(acc, o, P);
        }
        fe_store(out, 0, acc);
        if (flag) {
            __threadfence_system();
            *flag = seq;
        }
    }
}

// Entry j (< 2^nb, nb <= 4) of the eq table over the index bits first .. first + nb - 1 of a point record PT (r[bit][8], Montgomery
// form; first and nb wave-uniform, so the coordinates are scalar loads from the argument segment):
//   (sel_0 * sel_1) * (sel_2 * sel_3),   sel_k = r_k if bit k of j is set, else 1 - r_k;   absent factors are 1
// TWO levels of multiplications, the two of the first level independent of each other (they share the multiplier's issue slots),
// instead of a chain of three: the table build is the head of every evaluate workgroup's dependent chain.  Exact field
// arithmetic: the same canonical entries whatever the association.
template <class PT>
ZK_D Fe eq_entry_depth2(const PT &pt, uint32_t first, uint32_t nb, uint32_t j, const FieldParams &P) {
    Fe one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
    Fe f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t idx = (uint32_t)k < nb ? first + k : first;   // (in range even when the factor is absent)
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = pt.r[idx][i];
        const Fe sel = (j >> k) & 1 ? r : fe_sub(one, r, P);
        f[k] = (uint32_t)k < nb ? sel : one;
    }
#ifdef ZK_EQ_CHAIN   // A/B builds only: round 4's chain of three dependent multiplications
    return fe_mul(fe_mul(fe_mul(f[0], f[1], P), f[2], P), f[3], P);
#else
    const Fe a = fe_mul(f[0], f[1], P), b = fe_mul(f[2], f[3], P);
    return fe_mul(a, b, P);
#endif
}

constexpr int kEvalLowMax = 12, kEvalLowMin = 8;   // L = 8: one element per thread
struct EvalLowPoint {   // r for index bit p (Montgomery form), p = 0 the least significant bit = the LAST variable
    uint32_t r[kEvalLowMax][8];
};
__global__ __launch_bounds__(kBlock) void k_eval_low(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t L,
                                                     EvalLowPoint pt, FieldParams P, volatile uint32_t *flag, uint32_t seq, EvalHighPoint ph) {
    __shared__ Fe eq[3][16];
    __shared__ Fe wg;   // eq(point_high, blockIdx.x) when the outputs are weighted (ph.n > 0): wave 3, which has no table to build
    __shared__ uint32_t red[kBlock / 64][8];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_m = 1u << (L - 8);   // elements per thread
    const uint64_t base = ((uint64_t)blockIdx.x << L) + tid;
    Fe x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((uint32_t)i < n_m) x[i] = fe_load(in, base + (uint64_t)i * 256);
    if (wave < 3) {   // table `wave`: bits 4*wave .. (the third table has L - 8 bits); lane j < 16 builds entry j
        const uint32_t nb = wave < 2 ? 4u : L - 8;
        const Fe acc = eq_entry_depth2(pt, wave < 2 ? 4 * wave : 8u, nb, lane, P);
        if (lane < 16) eq[wave][lane] = acc;
    } else if (ph.n) {
        const Fe f = eval_high_weight(ph, blockIdx.x, lane, P);
        if (lane == eval_high_lane(ph.n)) wg = f;
    }
    __syncthreads();
    WideAcc w;
    wide_zero(w);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((uint32_t)i < n_m) {
            const Fe m = eq[2][i];
            wide_mac(w, x[i].v, m.v);
        }
    const Fe wl = fe_mul(eq[1][tid >> 4], eq[0][tid & 15], P);
    Fe s = fe_mul(redc_wide(w, P), wl, P);
    s = fe_wave_sum(s, P);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid >> 6][i] = s.v[i];
    }
    __syncthreads();
    if (tid == 0) {
        Fe acc = s;
        for (int wv = 1; wv < kBlock / 64; ++wv) {
            Fe o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = red[wv][i];
            acc = fe_add(acc, o, P);
        }
        if (ph.n) acc = fe_mul(acc, wg, P);
        fe_store(out, blockIdx.x, acc);
        if (flag) {   // (single-workgroup launches only) completion word for the host, after the result
            __threadfence_system();
            *flag = seq;
        }
    }
}

// ---- MultiLinearPolynomial::evaluate, tail (evaluation_form.rs:83-89) -----------------------------------------------------
// Once the table is small every further fold is launch latency.  One 1024-thread workgroup finishes the last m <= 12
// variables: the first of them is folded while the 2^m elements are read from HBM (so 2^(m-1) elements = 64 KiB of LDS at
// m = 12), the rest in place in LDS with one barrier per variable.  chs: the m remaining assignments, prepared on the host.
constexpr int kEvalTailVars = 12;
constexpr int kEvalTailThreads = 1024;
constexpr int kEvalChWords = 9;
struct EvalTailChallenges {   // the m remaining assignments in prepared form, passed in the kernel arguments (no staging copy)
    uint32_t w[kEvalTailVars][kEvalChWords];
};
__global__ __launch_bounds__(kEvalTailThreads) void k_evaluate_tail(const uint64_t *__restrict__ in, uint32_t m,
                                                                    EvalTailChallenges chs, FieldParams P,
                                                                    uint64_t *__restrict__ out, volatile uint32_t *flag = nullptr, uint32_t seq = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ev_smem[];
    uint64_t *T = reinterpret_cast<uint64_t *>(ev_smem);
    const uint32_t tid = threadIdx.x;
    auto load_r = [&](uint32_t v) {   // v is wave-uniform: scalar loads from the argument segment
        Mul29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = chs.w[v][i];
        return r;
    };
    uint32_t q = 1u << (m - 1);
    {
        const Mul29 r = load_r(0);
        for (uint32_t j = tid; j < q; j += kEvalTailThreads) {
            const Fe lo = fe_load(in, j), hi = fe_load(in, j + q);
            fe_store(T, j, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
        }
    }
    __syncthreads();
    for (uint32_t v = 1; v < m; ++v) {
        q >>= 1;
        const Mul29 r = load_r(v);
        for (uint32_t j = tid; j < q; j += kEvalTailThreads) {
            const Fe lo = fe_load(T, j), hi = fe_load(T, j + q);
            fe_store(T, j, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));   // in place: j and j+q belong to this thread
        }
        __syncthreads();
    }
    if (tid == 0) {
        fe_store(out, 0, fe_load(T, 0));
        if (flag) {   // completion word for the host (capi.hip host_flag_wait): after the result, system scope
            __threadfence_system();
            *flag = seq;
        }
    }
}

// ---- ProductPoly::prod_reduce (product_poly.rs:66-74) --------------------------------------------------------
// n >= 64: the fold's wave-coalesced run access (two 1-KiB dwordx4 accesses per 64 elements and table + a DPP half swap, nontemporal:
// every table is streamed once).  Measured at 2^24: 0.62 (k = 2) / 0.57 (k = 3) of the HBM peak with 32 bytes per lane.
__global__ __launch_bounds__(kBlock) void k_prod_reduce_run(FactorPtrs fp, int k, uint64_t n, uint64_t *__restrict__ out, FieldParams P) {
    const uint32_t lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * kBlock) >> 6;
    uint4 *out4 = reinterpret_cast<uint4 *>(out);
    for (uint64_t e0 = wave * 64; e0 < n; e0 += nwaves * 64) {
        const uint64_t c0 = 2 * e0 + lane;
        Fe acc;
        {
            const uint4 *q = reinterpret_cast<const uint4 *>(fp.in[0]);
            acc = pair_gather(nt_load16(q + c0), nt_load16(q + c0 + 64), odd);
        }
        for (int f = 1; f < k; ++f) {
            const uint4 *q = reinterpret_cast<const uint4 *>(fp.in[f]);
            acc = fe_mul_tt(acc, pair_gather(nt_load16(q + c0), nt_load16(q + c0 + 64), odd), P);   // carry-free table x table product
        }
        uint4 oa, ob;
        pair_scatter(acc, odd, oa, ob);
        nt_store16(oa, out4 + c0);
        nt_store16(ob, out4 + c0 + 64);
    }
}
__global__ __launch_bounds__(kBlock) void k_prod_reduce(FactorPtrs fp, int k, uint64_t n, uint64_t *__restrict__ out,
                                                        FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        Fe acc = fe_load(fp.in[0], j);
        for (int f = 1; f < k; ++f) acc = fe_mul(acc, fe_load(fp.in[f], j), P);
        fe_store(out, j, acc);
    }
}

// Second stage of a round: one workgroup adds the per-block partials -> ns sums (Montgomery form); then, on lane 0,
//   out_rp   != null : store the sums (the round polynomial);
//   lanes    != null : store them as 8 zero-extended 32-bit digits per element (uint64 lanes) for the cross-GPU
//                      all-reduce (RCCL has no mod-p sum; integer lane sums cannot overflow);
//   sponge   != null : run the transcript step and publish the challenge.
// init (optional, a kernel ARGUMENT): the proof's initial sponge state.  The tail that closes round 0 then takes the state from its own
// arguments instead of from *sponge, and clears the two counters at zero2 -- the launch that used to do both in front of round 0
// (k_store_sponge) disappears from the single proof's serial chain.  25 selects on SGPR operands: a dynamic index into a by-value
// argument would send it through scratch.
ZK_D LaneSponge lane_sponge_from_arg(const WordSponge &w, const LaneKeccak &L) {
    LaneSponge sp;
    sp.a = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i)
        if (L.index == i) sp.a = w.s[i];
    sp.pos = w.pos;
    return sp;
}
ZK_D void round_tail_body(const uint64_t *__restrict__ partials, uint32_t nblocks, uint32_t ns, WordSponge *__restrict__ sponge,
                          uint64_t *__restrict__ out_rp, uint64_t *__restrict__ out_ch, uint64_t *__restrict__ d_challenge,
                          uint64_t *__restrict__ lanes, const FieldParams &P, const TailDerive &dv, const WordSponge *init = nullptr,
                          uint64_t *__restrict__ zero2 = nullptr) {
    __shared__ Fe fin[256];
    __shared__ Fe claim;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;
    const bool derive1 = dv.prev_rp != nullptr;   // the partials carry no t = 1 sums (k_round_kd SKIP1)
    // wave 0 fetches the sponge first, so that load is in flight while the partials are reduced
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = {0, 0};
    if (sponge && wave0) sp = init ? lane_sponge_from_arg(*init, L) : lane_sponge_load(sponge, L);
    if (init && zero2 && threadIdx.x >= 64 && threadIdx.x < 66) zero2[threadIdx.x - 64] = 0;
    // wave w owns the sums t = w, w+4, ...: lanes stride over the blocks' partials, one VALU wave reduction, one barrier
    for (uint32_t t = wave; t < ns; t += kBlock / 64) {
        if (derive1 && t == 1 && (dv.claim || dv.local_only)) {
            // this wave would own t = 1: the round kernel's claim workgroup has evaluated S_prev(r_prev) already -- or nobody needs
            // it here: a sharded round derives S(1) after the all-reduce (k_lanes_transcript reads or evaluates the claim itself)
            if (lane == 0) {
                if (dv.local_only) fin[1] = fe_zero();
                else claim = fe_load(dv.claim, 0);
            }
            continue;
        }
        if (derive1 && t == 1) {
            // this wave would own t = 1: it evaluates the previous round polynomial at the previous challenge instead,
            //   claim = sum_t prev[t] * w[t] * prod_{u != t} (r - u)   (lane t takes term t; D + 1 multiplies deep)
            const Fe r = fe_load(dv.prev_chal, 0);
            Fe term = fe_zero();
            if ((uint32_t)lane < ns) {
                Fe one;
#pragma unroll
                for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
                const Fe wt = fe_load(dv.w, lane);
                term = fe_mul(fe_load(dv.prev_rp, lane), wt, P);
                Fe node = fe_zero();   // Montgomery form of u
                for (uint32_t u = 0; u < ns; ++u) {
                    if (u != (uint32_t)lane) term = fe_mul(term, fe_sub(r, node, P), P);
                    node = fe_add(node, one, P);
                }
            }
            term = fe_wave_sum(term, P, 8);
            if (lane == 0) claim = term;
            continue;
        }
        Fe s = fe_zero();
        for (uint32_t b = lane; b < nblocks; b += 8 * 64) {   // eight independent loads in flight, then a tree of adds
            Fe x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = b + u * 64 < nblocks ? fe_load(partials, (uint64_t)(b + u * 64) * ns + t) : fe_zero();
            const Fe lo = fe_add(fe_add(x[0], x[1], P), fe_add(x[2], x[3], P), P), hi = fe_add(fe_add(x[4], x[5], P), fe_add(x[6], x[7], P), P);
            s = fe_add(s, fe_add(lo, hi, P), P);
        }
        s = fe_wave_sum(s, P);
        if (lane == 0) fin[t] = s;
    }
    __syncthreads();
    if ((derive1 || dv.lead) && !dv.local_only) {
        if (threadIdx.x == 0) {
            if (derive1) fin[1] = fe_sub(claim, fin[0], P);        // S(1) = S_prev(r_prev) - S(0)
            if (dv.lead) fin[dv.lead] = lead_rebuild(dv.lead, fin, P);   // slot D held the leading coefficient (k_round_kd LEAD)
        }
        __syncthreads();
    }
    if (threadIdx.x == kBlock - 64) {   // the last wave stores the round polynomial while wave 0 runs the transcript
        for (uint32_t t = 0; t < ns; ++t) {
            if (out_rp) fe_store(out_rp, t, fin[t]);
            if (lanes)
                for (int i = 0; i < 8; ++i) lanes[8 * t + i] = (uint64_t)fin[t].v[i];
        }
    }
    if (sponge && wave0) {
        lane_absorb_elems(sp, L, fin, ns, P);
        Mul29 ch29;
        Fe ch;
        challenge_forms(lane_squeeze_x(sp, L), (uint32_t)lane, P, ch29, ch);   // lane 0: multiplier form, lane 16: Montgomery form
        publish_challenge_forms(d_challenge, out_ch, ch, ch29, (uint32_t)lane);
        lane_sponge_store(sponge, sp, L);
    }
}
__global__ __launch_bounds__(kBlock) void k_round_tail(const uint64_t *__restrict__ partials, uint32_t nblocks, uint32_t ns,
                                                       WordSponge *__restrict__ sponge, uint64_t *__restrict__ out_rp,
                                                       uint64_t *__restrict__ out_ch, uint64_t *__restrict__ d_challenge,
                                                       uint64_t *__restrict__ lanes, FieldParams P, TailDerive dv = {}) {
    round_tail_body(partials, nblocks, ns, sponge, out_rp, out_ch, d_challenge, lanes, P, dv);
}
// the tail of round 0 of a single proof: the initial sponge state arrives as an argument (round_tail_body)
__global__ __launch_bounds__(kBlock) void k_round_tail_init(const uint64_t *__restrict__ partials, uint32_t nblocks, uint32_t ns,
                                                            WordSponge *__restrict__ sponge, uint64_t *__restrict__ out_rp,
                                                            uint64_t *__restrict__ out_ch, uint64_t *__restrict__ d_challenge, FieldParams P,
                                                            TailDerive dv, WordSponge init, uint64_t *__restrict__ zero2) {
    round_tail_body(partials, nblocks, ns, sponge, out_rp, out_ch, d_challenge, nullptr, P, dv, &init, zero2);
}
// batched form (zk_sumcheck_prove_batch): grid (1, proofs), the B transcript steps of a round side by side
struct TailSlot {
    const uint64_t *partials;
    WordSponge *sponge;
    uint64_t *out_rp, *out_ch, *d_challenge;
    TailDerive dv;
};
__global__ __launch_bounds__(kBlock) void k_round_tail_b(BatchOf<TailSlot> b, uint32_t nblocks, uint32_t ns, FieldParams P) {
    const TailSlot &a = b.a[blockIdx.y];
    round_tail_body(a.partials, nblocks, ns, a.sponge, a.out_rp, a.out_ch, a.d_challenge, nullptr, P, a.dv);
}

// ---- finisher: all remaining rounds of the prover in ONE launch, once the tables are small ---------------------------
// A round on a tiny table is pure latency (two launches, a trip through HBM for 96 bytes of sums, the serial transcript).
// One workgroup keeps the K tables (<= 2^kFinishVars elements each) in LDS and the sponge in the registers of wave 0 and
// loops: sums -> workgroup reduction -> transcript step -> fold in LDS (prover.rs:44-68, unchanged semantics).
//   pending != 0: the tables in HBM still need the previous challenge applied (prover.rs:64) while they are loaded.
//   out_final: optional, K elements: the factors evaluated at the full challenge point.
constexpr int kFinishVars = 9;
template <int K, int D>
__global__ __launch_bounds__(kBlock) void k_finish(FactorPtrs fp, uint32_t m_in, int pending, FieldParams P,
                                                   uint64_t *d_challenge, WordSponge *gsponge, uint64_t *out_rp, uint64_t *out_ch,
                                                   uint64_t *out_final) {
    constexpr int NS = D + 1;
    // all LDS is carved from the dynamic region at 16-byte aligned offsets (no static __shared__ in front of it)
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;
    uint32_t m = pending ? m_in - 1 : m_in;       // variables of the tables held in LDS
    const uint32_t n_elems = 1u << m;
    uint64_t *tab = reinterpret_cast<uint64_t *>(fin_smem);                 // K tables of 2^m elements, 4 u64 each
    unsigned char *carve = fin_smem + (size_t)K * n_elems * 32;
    uint32_t(*red)[NS][8] = reinterpret_cast<uint32_t(*)[NS][8]>(carve);    // [waves][NS][8]
    Fe *fin = reinterpret_cast<Fe *>(carve + (kBlock / 64) * NS * 32);
    Mul29 *sh_r29p = reinterpret_cast<Mul29 *>(carve + (kBlock / 64) * NS * 32 + NS * 32);
    // ---- load (and fold when pending) ----
    if (pending) {
        const Mul29 r = load_challenge29(d_challenge);
#pragma unroll
        for (int f = 0; f < K; ++f)
            for (uint32_t j = tid; j < n_elems; j += kBlock) {
                const Fe lo = fe_load(fp.in[f], j), hi = fe_load(fp.in[f], j + n_elems);
                fe_store(tab + (size_t)f * n_elems * 4, j, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
            }
    } else {
#pragma unroll
        for (int f = 0; f < K; ++f)
            for (uint32_t j = tid; j < n_elems; j += kBlock) fe_store(tab + (size_t)f * n_elems * 4, j, fe_load(fp.in[f], j));
    }
    LaneKeccak L = lane_keccak_init();
    LaneSponge sp = {0, 0};
    if (wave0) sp = lane_sponge_load(gsponge, L);
    __syncthreads();
    uint32_t round = 0;
    while (m >= 1) {
        const uint32_t q = 1u << (m - 1);
        if (q <= 128) {
            // ---- one evaluation point per WAVE: wave t forms S_t over all pairs (k - 1 multiplies per pair instead of
            // (D + 1)(k - 1) on one lane), reduces it on the VALU and owns fin[t]: no cross-wave exchange ----
            for (uint32_t t = wave; t < (uint32_t)NS; t += kBlock / 64) {
                Fe s = fe_zero();
                for (uint32_t j = lane; j < q; j += 64) {
                    Fe prod;
#pragma unroll
                    for (int f = 0; f < K; ++f) {
                        const uint64_t *T = tab + (size_t)f * n_elems * 4;
                        const Fe lo = fe_load(T, j), hi = fe_load(T, j + q);
                        Fe v = lo;
                        if (t >= 1) v = hi;
                        if (t >= 2) {
                            const Fe diff = fe_sub(hi, lo, P);
                            for (uint32_t u = 1; u < t; ++u) v = fe_add(v, diff, P);
                        }
                        prod = (f == 0) ? v : fe_mul(prod, v, P);
                    }
                    s = fe_add(s, prod, P);
                }
                s = fe_wave_sum(s, P, q < 64 ? q : 64);
                if (lane == 0) {
                    fin[t] = s;
                    fe_store(out_rp, (uint64_t)round * NS + t, s);
                }
            }
            __syncthreads();
        } else {
        // ---- sums over the pairs (j, j+q) ----
        Fe sum[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) sum[t] = fe_zero();
        for (uint32_t j = tid; j < q; j += kBlock) {
            Fe prod[NS];
#pragma unroll
            for (int f = 0; f < K; ++f) {
                const uint64_t *T = tab + (size_t)f * n_elems * 4;
                const Fe lo = fe_load(T, j), hi = fe_load(T, j + q);
                const Fe diff = fe_sub(hi, lo, P);
                Fe v = lo;
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    if (t == 1) v = hi;
                    else if (t > 1) v = fe_add(v, diff, P);
                    prod[t] = (f == 0) ? v : fe_mul(prod[t], v, P);
                }
            }
#pragma unroll
            for (int t = 0; t < NS; ++t) sum[t] = fe_add(sum[t], prod[t], P);
        }
        // ---- workgroup reduction -> fin[] ----
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            sum[t] = fe_wave_sum(sum[t], P, q < 64 ? q : 64);   // lanes >= q hold zero: empty levels are skipped
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) red[wave][t][i] = sum[t].v[i];
            }
        }
        __syncthreads();
        if (tid < NS) {
            Fe acc;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc.v[i] = red[0][tid][i];
            if (q > 64) {
                for (int w = 1; w < kBlock / 64; ++w) {
                    Fe o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o.v[i] = red[w][tid][i];
                    acc = fe_add(acc, o, P);
                }
            }
            fin[tid] = acc;
            fe_store(out_rp, (uint64_t)round * NS + tid, acc);
        }
        __syncthreads();
        }
        // ---- transcript step on wave 0, challenge to everyone through LDS ----
        if (wave0) {
            lane_absorb_elems(sp, L, fin, NS, P);
            Mul29 ch29;
            Fe ch;
            challenge_forms(lane_squeeze_x(sp, L), lane, P, ch29, ch);   // lane 0: multiplier form, lane 16: Montgomery form
            if (lane == 16) fe_store(out_ch, round, ch);
            if (lane == 0) *sh_r29p = ch29;
            if (m == 1) publish_challenge_forms(d_challenge, nullptr, ch, ch29, lane);   // last one: for the sharded tail
        }
        __syncthreads();
        // ---- fold at the challenge, in LDS (prover.rs:64); the fold after the last round is dropped by the reference
        if (m > 1) {
            Mul29 r;
#pragma unroll
            for (int i = 0; i < 9; ++i) r.l[i] = __builtin_amdgcn_readfirstlane(sh_r29p->l[i]);
            // work items (factor, index): every lane has at most ceil(k*q / 256) multiplies
            const uint32_t lq = m - 1;
            for (uint32_t i = tid; i < ((uint32_t)K << lq); i += kBlock) {
                uint64_t *T = tab + (size_t)(i >> lq) * n_elems * 4;
                const uint32_t j = i & (q - 1);
                const Fe lo = fe_load(T, j), hi = fe_load(T, j + q);
                fe_store(T, j, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
            }
            __syncthreads();
        }
        --m;
        ++round;
    }
    if (wave0) lane_sponge_store(gsponge, sp, L);
    // out_final != null: the factors at the challenge point, i.e. the fold after the last round (the reference computes
    // and drops it, prover.rs:64; a layered driver needs W(u))
    if (out_final && tid < K) {
        Mul29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = sh_r29p->l[i];
        const uint64_t *T = tab + (size_t)tid * n_elems * 4;
        const Fe lo = fe_load(T, 0), hi = fe_load(T, 1);
        fe_store(out_final, tid, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
    }
}

// The same finisher for a SUM of products (zk_sumcheck_prove_terms; a GKR layer polynomial): the factor list is grouped into
// terms at run time, each pair contributes sum_i prod_{f in term i}.
template <int D>
__global__ __launch_bounds__(kBlock) void k_finish_terms(FactorPtrs fp, TermSpec ts, uint32_t m_in, int pending, FieldParams P,
                                                   uint64_t *d_challenge, WordSponge *gsponge, uint64_t *out_rp, uint64_t *out_ch,
                                                   uint64_t *out_final) {
    constexpr int NS = D + 1;
    int K = 0;   // factors in all (runtime here)
    for (int i = 0; i < ts.n_terms; ++i) K += ts.term_k[i];
    // all LDS is carved from the dynamic region at 16-byte aligned offsets (no static __shared__ in front of it)
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;
    uint32_t m = pending ? m_in - 1 : m_in;       // variables of the tables held in LDS
    const uint32_t n_elems = 1u << m;
    uint64_t *tab = reinterpret_cast<uint64_t *>(fin_smem);                 // K tables of 2^m elements, 4 u64 each
    unsigned char *carve = fin_smem + (size_t)K * n_elems * 32;
    uint32_t(*red)[NS][8] = reinterpret_cast<uint32_t(*)[NS][8]>(carve);    // [waves][NS][8]
    Fe *fin = reinterpret_cast<Fe *>(carve + (kBlock / 64) * NS * 32);
    Mul29 *sh_r29p = reinterpret_cast<Mul29 *>(carve + (kBlock / 64) * NS * 32 + NS * 32);
    // ---- load (and fold when pending) ----
    if (pending) {
        const Mul29 r = load_challenge29(d_challenge);
#pragma unroll
        for (int f = 0; f < K; ++f)
            for (uint32_t j = tid; j < n_elems; j += kBlock) {
                const Fe lo = fe_load(fp.in[f], j), hi = fe_load(fp.in[f], j + n_elems);
                fe_store(tab + (size_t)f * n_elems * 4, j, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
            }
    } else {
#pragma unroll
        for (int f = 0; f < K; ++f)
            for (uint32_t j = tid; j < n_elems; j += kBlock) fe_store(tab + (size_t)f * n_elems * 4, j, fe_load(fp.in[f], j));
    }
    LaneKeccak L = lane_keccak_init();
    LaneSponge sp = {0, 0};
    if (wave0) sp = lane_sponge_load(gsponge, L);
    __syncthreads();
    uint32_t round = 0;
    while (m >= 1) {
        const uint32_t q = 1u << (m - 1);
        if (q <= 128) {
            // ---- one evaluation point per WAVE (see k_finish) ----
            for (uint32_t t = wave; t < (uint32_t)NS; t += kBlock / 64) {
                Fe s = fe_zero();
                for (uint32_t j = lane; j < q; j += 64) {
                    int f = 0;
                    for (int i = 0; i < ts.n_terms; ++i) {
                        Fe prod;
                        for (int g = 0; g < ts.term_k[i]; ++g, ++f) {
                            const uint64_t *T = tab + (size_t)f * n_elems * 4;
                            const Fe lo = fe_load(T, j), hi = fe_load(T, j + q);
                            Fe v = lo;
                            if (t >= 1) v = hi;
                            if (t >= 2) {
                                const Fe diff = fe_sub(hi, lo, P);
                                for (uint32_t u = 1; u < t; ++u) v = fe_add(v, diff, P);
                            }
                            prod = (g == 0) ? v : fe_mul(prod, v, P);
                        }
                        s = fe_add(s, prod, P);
                    }
                }
                s = fe_wave_sum(s, P, q < 64 ? q : 64);
                if (lane == 0) {
                    fin[t] = s;
                    fe_store(out_rp, (uint64_t)round * NS + t, s);
                }
            }
            __syncthreads();
        } else {
        // ---- sums over the pairs (j, j+q) ----
        Fe sum[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) sum[t] = fe_zero();
        for (uint32_t j = tid; j < q; j += kBlock) {
            int f = 0;
            for (int i = 0; i < ts.n_terms; ++i) {
                Fe prod[NS];
                for (int g = 0; g < ts.term_k[i]; ++g, ++f) {
                    const uint64_t *T = tab + (size_t)f * n_elems * 4;
                    const Fe lo = fe_load(T, j), hi = fe_load(T, j + q);
                    const Fe diff = fe_sub(hi, lo, P);
                    Fe v = lo;
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        if (t == 1) v = hi;
                        else if (t > 1) v = fe_add(v, diff, P);
                        prod[t] = (g == 0) ? v : fe_mul(prod[t], v, P);
                    }
                }
#pragma unroll
                for (int t = 0; t < NS; ++t) sum[t] = fe_add(sum[t], prod[t], P);
            }
        }
        // ---- workgroup reduction -> fin[] ----
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            sum[t] = fe_wave_sum(sum[t], P, q < 64 ? q : 64);   // lanes >= q hold zero: empty levels are skipped
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) red[wave][t][i] = sum[t].v[i];
            }
        }
        __syncthreads();
        if (tid < NS) {
            Fe acc;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc.v[i] = red[0][tid][i];
            if (q > 64) {
                for (int w = 1; w < kBlock / 64; ++w) {
                    Fe o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o.v[i] = red[w][tid][i];
                    acc = fe_add(acc, o, P);
                }
            }
            fin[tid] = acc;
            fe_store(out_rp, (uint64_t)round * NS + tid, acc);
        }
        __syncthreads();
        }
        // ---- transcript step on wave 0, challenge to everyone through LDS ----
        if (wave0) {
            lane_absorb_elems(sp, L, fin, NS, P);
            Mul29 ch29;
            Fe ch;
            challenge_forms(lane_squeeze_x(sp, L), lane, P, ch29, ch);   // lane 0: multiplier form, lane 16: Montgomery form
            if (lane == 16) fe_store(out_ch, round, ch);
            if (lane == 0) *sh_r29p = ch29;
            if (m == 1) publish_challenge_forms(d_challenge, nullptr, ch, ch29, lane);   // last one: for the sharded tail
        }
        __syncthreads();
        // ---- fold at the challenge, in LDS (prover.rs:64); the fold after the last round is dropped by the reference
        if (m > 1) {
            Mul29 r;
#pragma unroll
            for (int i = 0; i < 9; ++i) r.l[i] = __builtin_amdgcn_readfirstlane(sh_r29p->l[i]);
            const uint32_t lq = m - 1;
            for (uint32_t i = tid; i < ((uint32_t)K << lq); i += kBlock) {
                uint64_t *T = tab + (size_t)(i >> lq) * n_elems * 4;
                const uint32_t j = i & (q - 1);
                const Fe lo = fe_load(T, j), hi = fe_load(T, j + q);
                fe_store(T, j, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
            }
            __syncthreads();
        }
        --m;
        ++round;
    }
    if (wave0) lane_sponge_store(gsponge, sp, L);
    // out_final != null: the factors at the challenge point, i.e. the fold after the last round (the reference computes
    // and drops it, prover.rs:64; a layered driver needs W(u))
    if (out_final && (int)tid < K) {
        Mul29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = sh_r29p->l[i];
        const uint64_t *T = tab + (size_t)tid * n_elems * 4;
        const Fe lo = fe_load(T, 0), hi = fe_load(T, 1);
        fe_store(out_final, tid, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), r, P), P));
    }
}

// the prover's initial sponge state, passed by value in the kernel arguments
// zero2 (optional): two uint64 words cleared by the same launch (the pipelined rounds' last-block-done counters: a 16-byte
// hipMemsetAsync is a launch of its own on the stream)
ZK_D void store_sponge_body(const WordSponge &w, WordSponge *__restrict__ dst, uint64_t *__restrict__ zero2) {
    if (threadIdx.x < 25) dst->s[threadIdx.x] = w.s[threadIdx.x];
    if (threadIdx.x == 0) {
        dst->pos = w.pos;
        dst->pad_ = 0;
    }
    if (zero2 && threadIdx.x >= 32 && threadIdx.x < 34) zero2[threadIdx.x - 32] = 0;
}
__global__ void k_store_sponge(WordSponge w, WordSponge *__restrict__ dst, uint64_t *__restrict__ zero2) { store_sponge_body(w, dst, zero2); }
struct SpongeSlot {
    WordSponge w;
    WordSponge *dst;
    uint64_t *zero2;
};
__global__ void k_store_sponge_b(BatchOf<SpongeSlot> b) {
    const SpongeSlot &a = b.a[blockIdx.y];
    store_sponge_body(a.w, a.dst, a.zero2);
}

// Sharded prover, after the all-reduce: lanes hold sums over ranks of 32-bit digits.  Carry-propagate, reduce mod p
// (value < world * p, world <= 2^16), then the same transcript step.  One wave.
// dv: the round kernels left out the t = 1 sums (prev_rp != null: S(1) = S_prev(r_prev) - S(0), the identity holds for the GLOBAL
// sums) and / or put the leading coefficient in slot D (lead: the slot is linear in the shards, so its all-reduced value is the
// global leading coefficient) -- the same derivations k_round_tail makes on one GPU, made here on the all-reduced values.
__global__ __launch_bounds__(128) void k_lanes_transcript(const uint64_t *__restrict__ lanes, uint32_t ns, WordSponge *__restrict__ sponge,
                                                          uint64_t *__restrict__ out_rp, uint64_t *__restrict__ out_ch,
                                                          uint64_t *__restrict__ d_challenge, FieldParams P, TailDerive dv) {
    __shared__ Fe fin[256];
    __shared__ Fe claim_s;
    if (blockIdx.x != 0 || threadIdx.x >= 128) return;   // two waves, wave-uniform control flow
    const uint32_t lane = threadIdx.x & 63;
    const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    const bool fixup = dv.prev_rp != nullptr || dv.lead != 0;
    // wave 0 fetches the sponge first, so that load is in flight while the lanes are reduced (as k_round_tail does)
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = {0, 0};
    if (wave0) {
        sp = lane_sponge_load(sponge, L);
        Fe claim_early = fe_zero();
        if (dv.prev_rp && dv.claim) claim_early = fe_load(dv.claim, 0);   // evaluated by k_round_tail in front of the all-reduce; in flight with the lanes
        // lane t takes sum t (t, t + 64, ...): the carry propagation and the 17-step ladder run once for all sums of a batch instead
        // of once per sum on every lane (the ladder is ~350 dependent steps: ~1 us of the serial chain of every exchanging round)
        for (uint32_t t0 = 0; t0 < ns; t0 += 64) {
            const uint32_t t = t0 + lane < ns ? t0 + lane : ns - 1;
            uint32_t v[9];
            uint64_t carry = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                carry += lanes[8 * t + i];
                v[i] = (uint32_t)carry;
                carry >>= 32;
            }
            v[8] = (uint32_t)carry;
            if (dv.log_world <= 3) ladder9<4>(v, P);   // value < 2^log_world * p: up to eight ranks 16p .. p suffices (wave-uniform branch)
            else ladder9<16>(v, P);
            Fe r;
#pragma unroll
            for (int i = 0; i < 8; ++i) r.v[i] = v[i];
            if (t0 + lane < ns) {
                fin[t] = r;
                if (!fixup) fe_store(out_rp, t, r);
            }
        }
        if (dv.prev_rp && dv.claim && lane == 0) claim_s = claim_early;
    } else if (dv.prev_rp && !dv.claim) {
        // wave 1, beside the ladder (nothing here depends on the lanes): claim = S_prev(r_prev) = sum_t prev[t] * w[t] * prod_{u != t} (r_prev - u),
        // lane t takes term t (fast degrees only: ns <= 5)
        const Fe r = fe_load(dv.prev_chal, 0);
        Fe term = fe_zero();
        if (lane < ns) {
            Fe one;
#pragma unroll
            for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
            term = fe_mul(fe_load(dv.prev_rp, lane), fe_load(dv.w, lane), P);
            Fe node = fe_zero();   // Montgomery form of u
            for (uint32_t u = 0; u < ns; ++u) {
                if (u != lane) term = fe_mul(term, fe_sub(r, node, P), P);
                node = fe_add(node, one, P);
            }
        }
        const Fe claim = fe_wave_sum(term, P, 8);   // valid in lane 0
        if (lane == 0) claim_s = claim;
    }
    __syncthreads();
    if (!wave0) return;
    if (fixup) {
        if (lane == 0) {
            if (dv.prev_rp) fin[1] = fe_sub(claim_s, fin[0], P);             // S(1) = S_prev(r_prev) - S(0), on the GLOBAL sums
            if (dv.lead) fin[dv.lead] = lead_rebuild(dv.lead, fin, P);        // slot D held the (global) leading coefficient
        }
        __builtin_amdgcn_wave_barrier();   // same wave: LDS operations execute in order
        if (lane < ns) fe_store(out_rp, lane, fin[lane]);
    }
    lane_absorb_elems(sp, L, fin, ns, P);
    Mul29 ch29;
    Fe ch;
    challenge_forms(lane_squeeze_x(sp, L), (uint32_t)L.lane, P, ch29, ch);
    publish_challenge_forms(d_challenge, out_ch, ch, ch29, (uint32_t)L.lane);
    lane_sponge_store(sponge, sp, L);
}

// interleave the all-gathered shard tables [rank][factor][2^s] into k tables of world * 2^s elements:
// table_f[local * world + rank] = gathered[rank][f][local]  (global index = local * world + rank)
__global__ __launch_bounds__(kBlock) void k_gather_to_tables(const uint64_t *__restrict__ gathered, FactorPtrs fp, uint32_t k,
                                                             uint32_t world, uint32_t s) {
    const uint64_t per_rank = (uint64_t)k << s, total = per_rank * world, stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const uint64_t rank = i / per_rank, rem = i % per_rank;
        const uint32_t f = (uint32_t)(rem >> s);
        const uint64_t local = rem & ((1ull << s) - 1);
        fe_store(fp.out[f], local * world + rank, fe_load(gathered, i));
    }
}

// ---- MultiLinearPolynomial::to_bytes (evaluation_form.rs:97-103): 32-byte big-endian canonical integers ----------
__global__ __launch_bounds__(kBlock) void k_to_bytes(const uint64_t *__restrict__ in, uint8_t *__restrict__ out,
                                                     uint64_t n, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const Fe c = fe_to_canonical(fe_load(in, j), P);
        uint4 *o = reinterpret_cast<uint4 *>(out + 32 * j);
        o[0] = make_uint4(__builtin_bswap32(c.v[7]), __builtin_bswap32(c.v[6]), __builtin_bswap32(c.v[5]),
                          __builtin_bswap32(c.v[4]));
        o[1] = make_uint4(__builtin_bswap32(c.v[3]), __builtin_bswap32(c.v[2]), __builtin_bswap32(c.v[1]),
                          __builtin_bswap32(c.v[0]));
    }
}

// ---- CoeffMultilinearPolynomial::to_evaluation_form (coefficient_form.rs:340-347) ---------------------------------------
// eval[idx] = sum over keys that are subsets of the point's variable set.  With key bit v <-> variable v and table index
// bit (n-1-v) <-> variable v, that is a zeta (subset-sum) transform of the coefficient vector placed at bit-reversed
// positions: scatter, then one in-place pass per variable: T[x | b] += T[x].
// (the round-4 global-pass form, kept behind ZK_ZETA_GLOBAL=1 for A/B; the shipped transform is zeta_kernels.cuh)
__global__ __launch_bounds__(kBlock) void k_scatter_terms(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ coeffs,
                                                          uint64_t n_terms, uint64_t *__restrict__ table) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < n_terms; t += stride)
        fe_store(table, idx[t], fe_load(coeffs, t));   // idx = bit-reversed keys, unique (merged on the host)
}
__global__ __launch_bounds__(kBlock) void k_zeta_pass(uint64_t *table, uint64_t pairs, uint32_t pos, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < pairs; j += stride) {
        const uint64_t lo = insert_zero_bit(j, pos), hi = lo | (1ull << pos);
        fe_store(table, hi, fe_add(fe_load(table, hi), fe_load(table, lo), P));
    }
}

// V consecutive bit positions pos .. pos+V-1 in one pass: a thread owns the 2^V elements that differ in those bits, applies the V
// levels of the subset-sum butterfly (hi += lo: additions only) in registers and writes the 2^V - 1 elements that changed --
// a third of the launches and 0.4 of the traffic of three single passes at V = 3.
template <int V>
__global__ __launch_bounds__(kBlock) void k_zeta_multi(uint64_t *table, uint64_t groups, uint32_t pos, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t low_mask = (1ull << pos) - 1;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += stride) {
        const uint64_t base = ((g >> pos) << (pos + V)) | (g & low_mask);
        Fe x[1 << V];
#pragma unroll
        for (int c = 0; c < (1 << V); ++c) x[c] = fe_load(table, base | ((uint64_t)c << pos));
#pragma unroll
        for (int b = 0; b < V; ++b)
#pragma unroll
            for (int c = 0; c < (1 << V); ++c)
                if (c & (1 << b)) x[c] = fe_add(x[c], x[c ^ (1 << b)], P);
#pragma unroll
        for (int c = 1; c < (1 << V); ++c) fe_store(table, base | ((uint64_t)c << pos), x[c]);
    }
}

// ---- synthetic inputs: element i = first hash(seed, i, attempt) < p, stored in Montgomery form --------
ZK_D uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void k_fill_random(uint64_t *__restrict__ out, uint64_t n, uint64_t seed,
                                                        uint64_t first, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint32_t topbits = P.bits & 31u;
    const uint32_t topmask = topbits ? ((1u << topbits) - 1u) : 0xffffffffu;
    const int toplimb = (int)((P.bits + 31u) / 32u) - 1;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const uint64_t h0 = splitmix64(seed ^ splitmix64(first + j));
        Fe c;
        for (uint64_t attempt = 0;; ++attempt) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint64_t x = splitmix64(h0 + 4 * attempt + (uint64_t)w);
                c.v[2 * w] = (uint32_t)x;
                c.v[2 * w + 1] = (uint32_t)(x >> 32);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i == toplimb) c.v[i] &= topmask;
                else if (i > toplimb) c.v[i] = 0;
            }
            Fe d;
            if (sub8(d.v, c.v, P.p)) break;   // borrow => c < p
        }
        fe_store(out, j, fe_from_canonical(c, P));
    }
}

// ---- fft crate, first form: bit-reversal + one radix-2 DIT stage per launch (fft/src/lib.rs:21-46 computes the
// same DFT recursively).  tw[i] = omega^i, i < n/2.
__global__ __launch_bounds__(kBlock) void k_bitrev_copy(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                        uint32_t log_n) {
    const uint64_t n = 1ull << log_n, stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const uint64_t rj = log_n ? (__brevll(j) >> (64 - log_n)) : 0;
        fe_store(out, rj, fe_load(in, j));
    }
}
__global__ __launch_bounds__(kBlock) void k_ntt_stage(uint64_t *__restrict__ data, const uint64_t *__restrict__ tw,
                                                      uint32_t log_n, uint32_t s, FieldParams P) {
    const uint64_t half = 1ull << (log_n - 1), stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t h = 1ull << s;   // butterflies span h within blocks of 2h
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < half; b += stride) {
        const uint64_t j = b & (h - 1), base = (b >> s) << (s + 1);
        const Fe w = fe_load(tw, j << (log_n - 1 - s));
        const Fe u = fe_load(data, base + j);
        const Fe t = fe_mul(w, fe_load(data, base + j + h), P);
        fe_store(data, base + j, fe_add(u, t, P));
        fe_store(data, base + j + h, fe_sub(u, t, P));
    }
}
// fft_internal with an omega that is NOT a primitive n-th root (fft/src/lib.rs:39-43 computes even[i] + omega^i * odd[i] and
// even[i] + omega^(i + m/2) * odd[i] literally; omega^(m/2) = -1 only for primitive roots): both twiddles are read from a
// full table tw[i] = omega^i, i < n.  Stage s has sub-transforms of size m = 2h, h = 2^s, whose omega is omega^(n/m).
__global__ __launch_bounds__(kBlock) void k_ntt_stage_generic(uint64_t *__restrict__ data, const uint64_t *__restrict__ tw,
                                                              uint32_t log_n, uint32_t s, FieldParams P) {
    const uint64_t half = 1ull << (log_n - 1), stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t h = 1ull << s;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < half; b += stride) {
        const uint64_t j = b & (h - 1), base = (b >> s) << (s + 1);
        const Fe w0 = fe_load(tw, j << (log_n - 1 - s)), w1 = fe_load(tw, (j + h) << (log_n - 1 - s));
        const Fe u = fe_load(data, base + j), o = fe_load(data, base + j + h);
        fe_store(data, base + j, fe_add(u, fe_mul(w0, o, P), P));
        fe_store(data, base + j + h, fe_add(u, fe_mul(w1, o, P), P));
    }
}
// tw[i] = omega^i for i < count: chunked -- each thread starts from omega^(first) via square-and-multiply
__global__ __launch_bounds__(kBlock) void k_twiddle_table(uint64_t *__restrict__ tw, uint64_t count, Fe omega,
                                                          FieldParams P) {
    constexpr uint64_t kChunk = 64;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t chunks = (count + kChunk - 1) / kChunk;
    for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < chunks; c += stride) {
        uint64_t e = c * kChunk;
        Fe acc = fe_one(P), base = omega;
        while (e) {
            if (e & 1) acc = fe_mul(acc, base, P);
            base = fe_sqr(base, P);
            e >>= 1;
        }
        const uint64_t end = (c * kChunk + kChunk < count) ? c * kChunk + kChunk : count;
        for (uint64_t i = c * kChunk; i < end; ++i) {
            fe_store(tw, i, acc);
            acc = fe_mul(acc, omega, P);
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_scale(uint64_t *__restrict__ data, uint64_t n, Fe s, FieldParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride)
        fe_store(data, j, fe_mul(fe_load(data, j), s, P));
}

// ---- PartialEq on tables: any differing 16-byte word raises the flag ------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_compare(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint64_t n16,
                                                    uint32_t *__restrict__ flag) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    bool diff = false;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n16; j += stride) {
        const uint4 x = a[j], y = b[j];
        diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    if (diff) atomicOr(flag, 1u);
}

// ---- measurement kernels -----------------------------------------------------------------------------------------
// dependent fe_mul chain entirely in registers: 2 independent chains per thread
__global__ __launch_bounds__(kBlock) void k_bench_modmul(uint64_t *__restrict__ out, int iters, FieldParams P, Fe seed, Mul29 c29,
                                                         int variant) {
    Fe a = seed, b = seed;
    a.v[0] ^= threadIdx.x;
    b.v[1] ^= blockIdx.x + 7u * threadIdx.x;   // both chains per-lane (a uniform chain would be scalarised)
    a = fe_from_canonical(fe_to_canonical(a, P), P);
    if (variant == 0) {          // saturated Montgomery product of two variable operands
        for (int i = 0; i < iters; ++i) {
            a = fe_mul(a, b, P);
            b = fe_mul(b, a, P);
        }
    } else {                     // unsaturated product by a prepared (wave-uniform) operand
        for (int i = 0; i < iters; ++i) {
            a = fe_mul29(a, c29, P);
            b = fe_mul29(b, c29, P);
        }
    }
    if (a.v[0] == 0x12345678u && b.v[3] == 0x9abcdef0u) fe_store(out, 0, a);   // keep the chain live
}
__global__ __launch_bounds__(kBlock) void k_bench_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n16; j += stride) out[j] = in[j];
}

}  // namespace zk
