// round_kernels.cuh -- the sumcheck round kernels (sumcheck/src/prover.rs:44-68 on the GPU).  Own translation unit
// (rounds.hip): these are the heavy instantiations.
#pragma once
#include "common.cuh"
#include "quad.cuh"

namespace zk {

// ---- workgroup reduction of NS field elements per thread -> partials[block][NS] -------------------------------
// SKIP1: sum[1] is not computed by the caller (the tail derives S(1) from the previous round's claim): not reduced, not stored.
template <int NS, bool SKIP1 = false>
ZK_D void block_reduce_store(Fe (&sum)[NS], uint64_t *__restrict__ partials, const FieldParams &P) {
    __shared__ uint32_t red[kBlock / 64][NS][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        if (SKIP1 && t == 1) continue;
        sum[t] = fe_wave_sum(sum[t], P);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) red[wave][t][i] = sum[t].v[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < NS && !(SKIP1 && threadIdx.x == 1)) {
        const int t = threadIdx.x;
        Fe acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = red[0][t][i];
        for (int w = 1; w < kBlock / 64; ++w) {
            Fe o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = red[w][t][i];
            acc = fe_add(acc, o, P);
        }
        fe_store(partials, (uint64_t)blockIdx.x * NS + t, acc);
    }
}

// ---- one sumcheck round (sumcheck/src/prover.rs:44-68), fused ---------------------------------------------------
// Computes the round polynomial in evaluation form, S_t = sum_x prod_f P_f(t, x) for t = 0..D, over the table pairs
// (lo, hi) = (T[j], T[j+q]).  With FUSED the pairs are first produced by folding the PREVIOUS round's table at its
// challenge r (prover.rs:64: T'[j] = T[j] - r*(T[j] - T[j+2q])), written back for the next round and used from
// registers -- one pass over HBM per round instead of the reference's (D+2)*k folds + (D+1) prod_reduce + sums.
//   P_f(t, x) = lo + t*(hi - lo)  ==  left - F::from(t)*(left - right)   (evaluation_form.rs:68; exact in F_p)
// For k >= 2 the last factor's products are accumulated UNREDUCED (512+ bits) and Montgomery-reduced once per
// kMaxLazy pairs; sums of Montgomery products are exact, so the reduced result is the same canonical element.
// In-place (out == in) is race-free: index j and j+q are read and written only by the thread that owns j.
template <int D, bool FUSED>
__global__ __launch_bounds__(kBlock) void k_round(FactorPtrs fp, int k, uint64_t q, FieldParams P,
                                                  const uint64_t *__restrict__ rptr, uint64_t *__restrict__ partials) {
    constexpr int NS = D + 1;
    // the previous round's challenge is produced on the device by k_round_tail (no host round trip)
    Mul29 r = {};
    if (FUSED) r = load_challenge29(rptr);
    Fe sum[NS];
    WideAcc acc[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        sum[t] = fe_zero();
        wide_zero(acc[t]);
    }
    int lazy = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < q; j += stride) {
        Fe prod[NS];
        for (int f = 0; f < k; ++f) {
            Fe lo, hi;
            if (FUSED) {
                const Fe a0 = fe_load(fp.in[f], j), a1 = fe_load(fp.in[f], j + q);
                const Fe a2 = fe_load(fp.in[f], j + 2 * q), a3 = fe_load(fp.in[f], j + 3 * q);
                lo = fe_sub(a0, fe_mul29(fe_sub(a0, a2, P), r, P), P);
                hi = fe_sub(a1, fe_mul29(fe_sub(a1, a3, P), r, P), P);
                fe_store(fp.out[f], j, lo);
                fe_store(fp.out[f], j + q, hi);
            } else {
                lo = fe_load(fp.in[f], j);
                hi = fe_load(fp.in[f], j + q);
            }
            const Fe diff = fe_sub(hi, lo, P);
            Fe v = lo;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                if (t == 1) v = hi;
                else if (t > 1) v = fe_add(v, diff, P);
                if (k == 1) sum[t] = fe_add(sum[t], v, P);
                else if (f == 0) prod[t] = v;
                else if (f < k - 1) prod[t] = fe_mul(prod[t], v, P);
                else wide_mac(acc[t], prod[t].v, v.v);
            }
        }
        if (k > 1 && ++lazy == kMaxLazy) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                sum[t] = fe_add(sum[t], redc_wide(acc[t], P), P);
                wide_zero(acc[t]);
            }
            lazy = 0;
        }
    }
    if (k > 1 && lazy) {
#pragma unroll
        for (int t = 0; t < NS; ++t) sum[t] = fe_add(sum[t], redc_wide(acc[t], P), P);
    }
    block_reduce_store<NS>(sum, partials, P);
}

// ---- the same round, specialised: K factors and degree D at compile time ------------------------------------------
// This is the kernel the prover normally runs (the common (k, D) shapes).  Differences from k_round above, all scheduling:
//  * every load of a pair-index j is issued before any arithmetic, and the loads of the thread's NEXT j are issued the
//    moment a factor's registers have been consumed, so a wave hides its own HBM latency (the arithmetic of one j is
//    ~7.7k cycles per wave, several times the memory latency) even at 2 waves per SIMD;
//  * the host sizes the grid so that a thread handles at most kMaxLazy pairs: the unreduced accumulators are reduced
//    once, after the loop (no flush path, no separate running sums in registers);
//  * factor indices are template parameters (round_factor<F>), so every array index is static and nothing spills.
// EXTRA = 1 adds a second, single-factor term to the sum (table index K): S_t = sum_x (prod_{f<K} P_f(t,x) + P_K(t,x)) --
// the GKR layer polynomial W*H + B in ONE pass instead of a second launch per round (zk_sumcheck_prove_terms, terms {K, 1}).
// the inner products of a K >= 3 term: table x table.  A/B build switch (-DZK_KD_INNER_MUL=fe_mul_tt: the carry-free form)
#ifndef ZK_KD_INNER_MUL
#define ZK_KD_INNER_MUL fe_mul
#endif
template <int K, int D, bool FUSED, int EXTRA = 0>
struct RoundRegs {
    static constexpr int NS = D + 1, NL = FUSED ? 4 : 2;
    // sums-only kernels over two tables prefetch TWO pair indices ahead (see round_factor); over three tables the second buffer
    // (48 registers) is what pushed them past 256 registers, i.e. down to one wave per SIMD: they prefetch one ahead, as the fused
    // kernels do, and run two waves per SIMD.  (Over two tables it makes no difference either way: 150 instead of 182 registers,
    // three waves instead of two, n = 24 within noise -- profiles/r03_round_kd_occupancy_ab.log.)
    static constexpr bool DEEP = !FUSED && K + EXTRA <= 2;
    Fe cur[K + EXTRA][NL];
    Fe nxt[DEEP ? K + EXTRA : 1][DEEP ? NL : 1];
    Fe prod[NS];
    Fe sum[NS];
    Fe sum_b[EXTRA ? NS : 1];
    WideAcc acc[K > 1 ? NS : 1];
};
// LEAD (K == D, big rounds): the last evaluation point is not formed.  The round polynomial has degree D with leading
// coefficient L = sum_x prod_f (hi_f - lo_f), so slot D accumulates prod_f diff_f instead of prod_f (v_f(D-1) + diff_f) -- one
// modular addition per factor and pair index fewer -- and the tail rebuilds S(D) = sum_{i<D} (-1)^(D-1-i) C(D,i) S(i) + D! L
// (the D-th finite difference of a degree-D polynomial is D! times its leading coefficient; exact in F_p, so the same
// canonical element the reference's direct sum gives).  The single-factor extra term is linear: it adds nothing to L.
template <int F, int K, int D, bool FUSED, int EXTRA, bool SKIP1 = false, bool LEAD = false>
ZK_D void round_factor(RoundRegs<K, D, FUSED, EXTRA> &R, const FactorPtrs &fp, uint64_t j, uint64_t jn, bool more, uint64_t q,
                       const Mul29 &r, const FieldParams &P, uint64_t jnn = 0, bool more2 = false) {
    static_assert(!LEAD || (K == D && D >= 1), "the leading-coefficient slot needs deg = K = D");
    constexpr int NS = D + 1, NL = FUSED ? 4 : 2;
    // the second factor of a three-table degree-3 product with all four slots live (round 0, LEAD): see the end of this function
    constexpr bool PAIR3 = K == 3 && D == 3 && F == 1 && LEAD && !SKIP1;
    Fe lo, hi;
    if (FUSED) {
        lo = fe_sub(R.cur[F][0], fe_mul29(fe_sub(R.cur[F][0], R.cur[F][2], P), r, P), P);
        hi = fe_sub(R.cur[F][1], fe_mul29(fe_sub(R.cur[F][1], R.cur[F][3], P), r, P), P);
        fe_store(fp.out[F], j, lo);
        fe_store(fp.out[F], j + q, hi);
    } else {
        lo = R.cur[F][0];
        hi = R.cur[F][1];
    }
    constexpr bool DEEP = RoundRegs<K, D, FUSED, EXTRA>::DEEP;
    if (!DEEP) {
        if (more) {   // this factor's input registers are free: start the next pair's loads now
#pragma unroll
            for (int l = 0; l < NL; ++l) R.cur[F][l] = fe_load(fp.in[F], jn + (uint64_t)l * q);
        }
    } else {
        // sums only: ~600 instructions per pair index do not cover the HBM latency at 2 waves per SIMD, so the loads run
        // two pair indices ahead (the second buffer costs 16 register moves per factor and iteration)
#pragma unroll
        for (int l = 0; l < NL; ++l) R.cur[F][l] = R.nxt[F][l];
        if (more2) {
#pragma unroll
            for (int l = 0; l < NL; ++l) R.nxt[F][l] = fe_load(fp.in[F], jnn + (uint64_t)l * q);
        }
    }
    const Fe diff = fe_sub(hi, lo, P);
    Fe v = lo;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        if (LEAD && t == D) v = diff;                                   // slot D: the leading coefficient
        else if (t == 1) v = hi;
        else if (t > 1) v = fe_add(v, diff, P);
        if (SKIP1 && t == 1) continue;                                  // S(1) = claim - S(0): derived in the tail
        if (LEAD && t == D && F == K) continue;                         // a linear term has no degree-D coefficient
        if (F == K) R.sum_b[t] = fe_add(R.sum_b[t], v, P);            // the extra single-factor term
        else if (K == 1) R.sum[t] = fe_add(R.sum[t], v, P);
        else if (F == 0) R.prod[t] = v;
        else if (PAIR3 && t == 2) continue;                             // (formed below from the other three pair products)
        else if (F < K - 1) R.prod[t] = ZK_KD_INNER_MUL(R.prod[t], v, P);   // (K >= 3: the products in front of the last, unreduced one)
        else wide_mac(R.acc[t], R.prod[t].v, v.v);
    }
    if (PAIR3) {
        // slots 0, 1, 3 hold A = lo0 lo1, B = hi0 hi1, C = d0 d1 (d = hi - lo); hi0 hi1 = A + (lo0 d1 + d0 lo1) + C, so the pair product at
        // t = 2, (lo0 + 2 d0)(lo1 + 2 d1) = A + 2 (B - A - C) + 4 C = 2 (B + C) - A: three modular additions instead of a multiplication
        const Fe bc = fe_add(R.prod[1], R.prod[3], P);
        R.prod[2] = fe_sub(fe_add(bc, bc, P), R.prod[0], P);
    }
}
// K <= 2 fits 2 waves per SIMD (<= 256 VGPRs); K >= 3 (or an extra term) keeps >= 12 elements in flight and gets the whole
// register file.
// SKIP1 (big fused rounds): the products for t = 1 are not formed at all -- S_i(0) + S_i(1) = S_{i-1}(r_{i-1}) holds
// identically for the sums the prover itself computed in the previous round (exact field arithmetic, so the derived S_i(1)
// is bit-identical to the computed one); k_round_tail rebuilds it from the previous round polynomial.
// Workgroups per CU the register budget is set for (two = at most 256 VGPRs = two waves per SIMD).  Two tables: two.  Three
// tables as ONE product (K = 3): two -- the fused kernels then spill 52-184 bytes per lane, but a second wave per SIMD hides more
// of the loads than the spills cost (k = 3, n = 20: 0.527 -> 0.515 ms).  A product plus a single-factor term (the GKR layer
// polynomial, EXTRA = 1): two for the fused SKIP1 kernels (12-44 bytes spilled; GKR proof 7.95 -> 7.76 ms), one for the fused
// kernels without SKIP1 (212 bytes).  The sums-only kernels over three tables need 212-222 registers since they prefetch one pair
// index ahead instead of two (RoundRegs::DEEP) and run two waves per SIMD on their own.  profiles/r03_round_kd_occupancy_ab.log
#ifndef ZK_KD_MIN_BLOCKS
#define ZK_KD_MIN_BLOCKS(K, EXTRA, FUSED, SKIP1) (((K) + (EXTRA) <= 2 || ((K) == 3 && (EXTRA) == 0) || ((EXTRA) == 1 && (FUSED) && (SKIP1))) ? 2 : 1)
#endif
template <int K, int D, bool FUSED, int EXTRA = 0, bool SKIP1 = false, bool LEAD = false>
ZK_D void round_kd_body(const FactorPtrs &fp, uint64_t q, const FieldParams &P, const uint64_t *__restrict__ rptr, uint64_t *__restrict__ partials,
                        const ClaimJob &cj) {
    constexpr int NS = D + 1, NL = FUSED ? 4 : 2;
    // SKIP1 with a claim job: the LAST workgroup of the grid is not a work block -- its first wave evaluates S_prev(r_prev) for the
    // tail (common.cuh ClaimJob) while the others stream the table
    uint32_t nblk = gridDim.x;
    if constexpr (SKIP1) {
        if (cj.out) {
            nblk -= 1;
            if (blockIdx.x == nblk) {
                if (threadIdx.x < 64) {
                    const Fe claim = claim_eval(cj, P);
                    if (threadIdx.x == 0) fe_store(cj.out, 0, claim);
                }
                return;
            }
        }
    }
    Mul29 r = {};
    if (FUSED) r = load_challenge29(rptr);
    RoundRegs<K, D, FUSED, EXTRA> R;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        R.sum[t] = fe_zero();
        if (EXTRA) R.sum_b[t] = fe_zero();
        if (K > 1) wide_zero(R.acc[t]);
    }
    const uint64_t stride = (uint64_t)nblk * kBlock;
    uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < q) {
#pragma unroll
        for (int f = 0; f < K + EXTRA; ++f)
#pragma unroll
            for (int l = 0; l < NL; ++l) R.cur[f][l] = fe_load(fp.in[f], j + (uint64_t)l * q);
    }
    if (RoundRegs<K, D, FUSED, EXTRA>::DEEP && j + stride < q) {
#pragma unroll
        for (int f = 0; f < K + EXTRA; ++f)
#pragma unroll
            for (int l = 0; l < NL; ++l) R.nxt[f][l] = fe_load(fp.in[f], j + stride + (uint64_t)l * q);
    }
    while (j < q) {
        const uint64_t jn = j + stride, jnn = jn + stride;
        const bool more = jn < q, more2 = jnn < q;
        round_factor<0, K, D, FUSED, EXTRA, SKIP1, LEAD>(R, fp, j, jn, more, q, r, P, jnn, more2);
        if constexpr (K + EXTRA > 1) round_factor<1, K, D, FUSED, EXTRA, SKIP1, LEAD>(R, fp, j, jn, more, q, r, P, jnn, more2);
        if constexpr (K + EXTRA > 2) round_factor<2, K, D, FUSED, EXTRA, SKIP1, LEAD>(R, fp, j, jn, more, q, r, P, jnn, more2);
        if constexpr (K + EXTRA > 3) round_factor<3, K, D, FUSED, EXTRA, SKIP1, LEAD>(R, fp, j, jn, more, q, r, P, jnn, more2);
        j = jn;
    }
    if (K > 1) {
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if (!(SKIP1 && t == 1)) R.sum[t] = redc_wide(R.acc[t], P);
    }
    if (EXTRA) {
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if (!(SKIP1 && t == 1)) R.sum[t] = fe_add(R.sum[t], R.sum_b[t], P);
    }
    block_reduce_store<NS, SKIP1>(R.sum, partials, P);
}
template <int K, int D, bool FUSED, int EXTRA = 0, bool SKIP1 = false, bool LEAD = false>
__global__ __launch_bounds__(kBlock, ZK_KD_MIN_BLOCKS(K, EXTRA, FUSED, SKIP1)) void k_round_kd(FactorPtrs fp, uint64_t q, FieldParams P,
                                                        const uint64_t *__restrict__ rptr, uint64_t *__restrict__ partials, ClaimJob cj = {}) {
    round_kd_body<K, D, FUSED, EXTRA, SKIP1, LEAD>(fp, q, P, rptr, partials, cj);
}
// per-proof arguments of a batched round launch (k_round_kd_b, k_round0_dot29_b, k_round_quad_b: grid (work blocks [+ claim], proofs))
struct RoundSlot {
    FactorPtrs4 fp;
    const uint64_t *rptr;
    uint64_t *partials;
    ClaimJob cj;
};
template <int K, int D, bool FUSED, int EXTRA = 0, bool SKIP1 = false, bool LEAD = false>
__global__ __launch_bounds__(kBlock, ZK_KD_MIN_BLOCKS(K, EXTRA, FUSED, SKIP1)) void k_round_kd_b(BatchOf<RoundSlot> b, uint64_t q, FieldParams P) {
    const RoundSlot &a = b.a[blockIdx.y];
    const FactorPtrs fp = factor_ptrs_of(a.fp);
    round_kd_body<K, D, FUSED, EXTRA, SKIP1, LEAD>(fp, q, P, a.rptr, a.partials, a.cj);
}

// ---- round 0 (sums only) of the two-table degree-2 shapes on carry-free 29-bit columns (round 4) ---------------------------------
// The first round of a proof has no fold: per pair index it is three products of table values (LEAD form: S(0) = sum lo0 lo1,
// S(1) = sum hi0 hi1, L = sum (hi0 - lo0)(hi1 - lo1)) and nothing else, i.e. the kernel IS its multiplier.  wide_mac spends 80
// v_mad_u64_u32 + 80 v_addc + shifts (~195 instructions) on a 256 x 256-bit product; with both operands split into nine 29-bit
// limbs (2 x ~20 instructions) the 81 limb products go straight into seventeen 64-bit column sums with no carry instruction at
// all (dot29_mac, the form k_eval_stream uses): ~130 per product.  A column takes 7 pair indices (63 products < 2^58) between
// carry normalisations; the three sums are converted to 512-bit integers and Montgomery-reduced once per thread, exactly as
// before, so the partials are the same canonical elements (exact integer sums of the same products).
// c (17 columns of weight 2^(29 k)) += x * y, both nine 29-bit limbs: 81 v_mad_u64_u32 and nothing else.  Two asm statements (an asm
// statement takes at most 30 operands): limbs 0-4 of x (columns 0..12), then limbs 5-8 (columns 5..16); i-major, so consecutive
// instructions write different columns.
ZK_D void dot29_mac(uint64_t (&c)[17], const uint32_t (&x)[9], const uint32_t (&y)[9]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    asm("v_mad_u64_u32 %0, vcc, %13, %18, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %13, %19, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %13, %20, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %13, %21, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %13, %22, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %13, %23, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %13, %24, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %13, %25, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %13, %26, %8\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %18, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %14, %19, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %14, %20, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %14, %21, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %14, %22, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %14, %23, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %14, %24, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %14, %25, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %14, %26, %9\n\t"
        "v_mad_u64_u32 %2, vcc, %15, %18, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %15, %19, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %15, %20, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %15, %21, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %15, %22, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %15, %23, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %15, %24, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %15, %25, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %15, %26, %10\n\t"
        "v_mad_u64_u32 %3, vcc, %16, %18, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %16, %19, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %16, %20, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %16, %21, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %16, %22, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %16, %23, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %16, %24, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %16, %25, %10\n\t"
        "v_mad_u64_u32 %11, vcc, %16, %26, %11\n\t"
        "v_mad_u64_u32 %4, vcc, %17, %18, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %17, %19, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %17, %20, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %17, %21, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %17, %22, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %17, %23, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %17, %24, %10\n\t"
        "v_mad_u64_u32 %11, vcc, %17, %25, %11\n\t"
        "v_mad_u64_u32 %12, vcc, %17, %26, %12"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %12, %16, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %17, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %12, %18, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %12, %19, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %12, %20, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %12, %21, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %12, %22, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %12, %23, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %12, %24, %8\n\t"
        "v_mad_u64_u32 %1, vcc, %13, %16, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %13, %17, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %13, %18, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %13, %19, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %13, %20, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %13, %21, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %13, %22, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %13, %23, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %13, %24, %9\n\t"
        "v_mad_u64_u32 %2, vcc, %14, %16, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %14, %17, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %14, %18, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %14, %19, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %14, %20, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %14, %21, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %14, %22, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %14, %23, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %14, %24, %10\n\t"
        "v_mad_u64_u32 %3, vcc, %15, %16, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %15, %17, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %15, %18, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %15, %19, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %15, %20, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %15, %21, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %15, %22, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %15, %23, %10\n\t"
        "v_mad_u64_u32 %11, vcc, %15, %24, %11"
        : "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]), "+v"(c[16])
        : "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8])
        : "vcc");
#else
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)x[i] * y[j];
#endif
}

ZK_D void dot29_normalise(uint64_t (&c)[17]) {
    constexpr uint64_t M = (1ull << 29) - 1;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        c[k + 1] += c[k] >> 29;
        c[k] &= M;
    }
}
// normalised columns (the top one unbounded) -> the 17-word integer redc_wide takes
ZK_D void dot29_to_wide(const uint64_t (&c)[17], WideAcc &w) {
    constexpr uint64_t M = (1ull << 29) - 1;
    uint64_t window = 0;
    int have = 0, wi = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const uint64_t limb = k < 16 ? c[k] : (k == 16 ? (c[16] & M) : (c[16] >> 29));   // 18 limbs below 2^29 (the last: what is left)
        window |= limb << have;
        have += 29;
        if (have >= 32) {
            if (wi < 17) w.v[wi] = (uint32_t)window;
            ++wi;
            window >>= 32;
            have -= 32;
        }
    }
    if (wi < 17) w.v[wi++] = (uint32_t)window;
#pragma unroll
    for (int i = 0; i < 17; ++i)
        if (i >= wi) w.v[i] = 0;
}
template <int EXTRA>
ZK_D void round0_dot29_body(const FactorPtrs &fp, uint64_t q, const FieldParams &P, uint64_t *__restrict__ partials) {
    constexpr int NT = 2 + EXTRA;
    uint64_t c0[17], c1[17], cL[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) c0[k] = c1[k] = cL[k] = 0;
    Fe sb0 = fe_zero(), sb1 = fe_zero();   // the single-factor term (EXTRA): linear, so it only feeds S(0) and S(1)
    constexpr bool DEEP = EXTRA == 0;   // two tables: loads two pair indices ahead (as k_round_kd); three: one ahead, or the registers run out
    Fe cur[NT][2], nxt[DEEP ? NT : 1][2];
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < q) {
#pragma unroll
        for (int f = 0; f < NT; ++f) cur[f][0] = fe_load(fp.in[f], j), cur[f][1] = fe_load(fp.in[f], j + q);
    }
    if (DEEP && j + stride < q) {
#pragma unroll
        for (int f = 0; f < NT; ++f) nxt[f][0] = fe_load(fp.in[f], j + stride), nxt[f][1] = fe_load(fp.in[f], j + stride + q);
    }
    int since = 0;
    while (j < q) {
        const uint64_t jn = j + stride, jnn = jn + stride;
        Fe v[NT][2];
#pragma unroll
        for (int f = 0; f < NT; ++f) v[f][0] = cur[f][0], v[f][1] = cur[f][1];
        if (DEEP) {
#pragma unroll
            for (int f = 0; f < NT; ++f) cur[f][0] = nxt[f][0], cur[f][1] = nxt[f][1];
            if (jnn < q) {
#pragma unroll
                for (int f = 0; f < NT; ++f) nxt[f][0] = fe_load(fp.in[f], jnn), nxt[f][1] = fe_load(fp.in[f], jnn + q);
            }
        } else if (jn < q) {
#pragma unroll
            for (int f = 0; f < NT; ++f) cur[f][0] = fe_load(fp.in[f], jn), cur[f][1] = fe_load(fp.in[f], jn + q);
        }
        uint32_t a[9], b[9];
        split29(v[0][0].v, a);
        split29(v[1][0].v, b);
        dot29_mac(c0, a, b);                                  // S(0): lo0 * lo1
        split29(v[0][1].v, a);
        split29(v[1][1].v, b);
        dot29_mac(c1, a, b);                                  // S(1): hi0 * hi1
        const Fe d0 = fe_sub(v[0][1], v[0][0], P), d1 = fe_sub(v[1][1], v[1][0], P);
        split29(d0.v, a);
        split29(d1.v, b);
        dot29_mac(cL, a, b);                                  // leading coefficient: (hi0 - lo0)(hi1 - lo1)
        if (EXTRA) {
            sb0 = fe_add(sb0, v[NT - 1][0], P);
            sb1 = fe_add(sb1, v[NT - 1][1], P);
        }
        if (++since == 7) {
            dot29_normalise(c0);
            dot29_normalise(c1);
            dot29_normalise(cL);
            since = 0;
        }
        j = jn;
    }
    dot29_normalise(c0);
    dot29_normalise(c1);
    dot29_normalise(cL);
    Fe sum[3];
    {
        WideAcc w;
        dot29_to_wide(c0, w);
        sum[0] = redc_wide(w, P);
        dot29_to_wide(c1, w);
        sum[1] = redc_wide(w, P);
        dot29_to_wide(cL, w);
        sum[2] = redc_wide(w, P);
    }
    if (EXTRA) {
        sum[0] = fe_add(sum[0], sb0, P);
        sum[1] = fe_add(sum[1], sb1, P);
    }
    block_reduce_store<3>(sum, partials, P);
}
template <int EXTRA>
__global__ __launch_bounds__(kBlock, 2) void k_round0_dot29(FactorPtrs fp, uint64_t q, FieldParams P, uint64_t *__restrict__ partials) {
    round0_dot29_body<EXTRA>(fp, q, P, partials);
}
template <int EXTRA>
__global__ __launch_bounds__(kBlock, 2) void k_round0_dot29_b(BatchOf<RoundSlot> b, uint64_t q, FieldParams P) {
    const RoundSlot &a = b.a[blockIdx.y];
    const FactorPtrs fp = factor_ptrs_of(a.fp);
    round0_dot29_body<EXTRA>(fp, q, P, a.partials);
}

// ---- the big rounds with their rows arriving by LDS-DMA (round 6) ------------------------------------------------------------------
// k_round_kd and k_round0_dot29 move 32 bytes per lane and access (fe_load / fe_store): every wave instruction touches half of
// sixteen 128-byte lines.  Here a wave's 64-pair-index RUN of a row (2 KiB) arrives as two global_load_lds_dwordx4 -- whole 1-KiB
// nontemporal pieces, no VGPR destination, lane l's 16 bytes at LDS offset 16 l -- into a per-wave ring, a lane reads its element back
// with two ds_read_b128 right before it needs it, and the folded half tables leave through pair_scatter (one DPP half swap) as whole
// 1-KiB stores: the fold kernel's data movement (kernels.cuh k_fold_msb) under the round kernels' arithmetic.  What that buys
// (tools/mb/mb_round0_glds.hip, mb_fused_glds.hip, mb_rows_pattern.hip; in the prover: profiles/r06_glds_sizes.log):
//  * the sums-only round of two tables was held at 0.63 of HBM by its access SHAPE, not by its prefetch depth (a ring of one unit does
//    what a ring of four does, at two, three or four waves per SIMD): 212 -> 168 us at 2^24 per table;
//  * the three-table FUSED kernels keep their prefetch in LDS instead of 48-96 registers and stop spilling: 8-13 % from the sizes in
//    rounds.hip (round 0 of three tables was built too -- carry-free columns, shared pair products -- and lost its lead when the shared
//    pair products went into k_round_kd as well: PAIR3 in round_factor; HISTORY.md);
//  * the fused round of TWO tables gains nothing in any form (eight read + four written streams reach 0.69-0.72 of HBM with every
//    multiplication removed) and has no instantiation here.
// The DMA instructions live in asm statements, so the compiler does not count them: every wait is a counted s_waitcnt vmcnt(N) placed
// by hand.  vmcnt retires in issue order and counts stores as well on this part; N is always "what was issued after the pieces I need".
// q must be a multiple of 64 (the host sends other sizes to the kernels above); a wave's runs are r0, r0 + rs, ... (wave-uniform).
extern __shared__ __attribute__((aligned(1024))) uint8_t glds_ring[];   // [4 waves][8 or 12 KiB]
constexpr uint32_t kGldsRingBytes = 4 * 8192, kGldsRing3Bytes = 4 * 12288;   // per workgroup: two 4-KiB units or one 8-KiB unit per wave; three 4-KiB units
template <int N>
ZK_D void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define ZK_GLDS_PIECES(ADDR)                      \
    "global_load_lds_dwordx4 %1, " ADDR " nt\n\t" \
    "global_load_lds_dwordx4 %1, " ADDR " offset:1024 nt\n\t"
#define ZK_GLDS_ROW(M0, ADDR) "s_mov_b32 m0, " M0 "\n\ts_nop 0\n\t" ZK_GLDS_PIECES(ADDR)
// two rows (lo, hi) of one run: four 1-KiB pieces; voff = 16 * lane, LDS rows 2 KiB apart from lds_dst (a wave-uniform byte address)
ZK_D void glds_rows2(uint64_t a0, uint64_t a1, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" ZK_GLDS_ROW("%4", "%2") ZK_GLDS_ROW("%5", "%3") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(a0), "s"(a1), "s"(lds_dst), "s"(lds_dst + 2048u)
                 : "memory");
}
// four rows (j, j + q, j + 2q, j + 3q) of one run: eight pieces
ZK_D void glds_rows4(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t" ZK_GLDS_ROW("%6", "%2") ZK_GLDS_ROW("%7", "%3") ZK_GLDS_ROW("%8", "%4") ZK_GLDS_ROW("%9", "%5") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(lds_dst), "s"(lds_dst + 2048u), "s"(lds_dst + 4096u), "s"(lds_dst + 6144u)
                 : "memory");
}
#undef ZK_GLDS_ROW
#undef ZK_GLDS_PIECES
ZK_D Fe glds_elem(const uint8_t *row, uint32_t elem) {
    const uint4 *p = reinterpret_cast<const uint4 *>(row + elem * 32);
    const uint4 a = p[0], b = p[1];
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
struct GldsWave {
    uint8_t *my;         // this wave's 8 KiB of the ring
    uint32_t my_lds;     // its LDS byte address (wave-uniform)
    uint32_t lane, voff, own;
    uint64_t r0, rs, K;  // first run, run stride, number of runs of this wave
};
ZK_D GldsWave glds_wave(uint64_t q, uint32_t nblk, uint32_t bytes_per_wave = 8192) {
    GldsWave w;
    w.lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.my = glds_ring + wave * bytes_per_wave;
    w.my_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)w.my);
    w.voff = w.lane * 16;
    w.own = pair_owned(w.lane);
    const uint64_t runs = q >> 6;
    w.r0 = (uint64_t)blockIdx.x * (kBlock / 64) + wave;
    w.rs = (uint64_t)nblk * (kBlock / 64);
    w.K = w.r0 < runs ? (runs - w.r0 + w.rs - 1) / w.rs : 0;
    return w;
}

// Round 0 of the two-table degree-2 product (k_round0_dot29's sums: S(0), S(1), leading coefficient), EXTRA = 1: plus a single-factor
// term (the GKR layer polynomial's round 0, k_round_kd<2, 2, false, 1, false, true>).  A unit = one table's (lo, hi) rows of a run =
// 4 KiB; the wave's ring holds one unit per table, each re-issued for the next run the moment a lane has copied it to registers.  Any
// number of runs per wave: the columns are reduced every kMaxLazy pair indices.
// waits: a unit's four pieces are followed by the other tables' units (this run's or the next one's): 4 (NU - 1) pieces, fewer in the
// last run
template <int NU, int F>
ZK_D void round0_glds_wait(bool last) {
    if (!last) wait_vm<4 * (NU - 1)>();
    else wait_vm<4 * (NU - 1 - F)>();
}
ZK_D void dot29_flush(uint64_t (&c)[17], Fe &sum, const FieldParams &P) {   // sum += c (normalised) * R^-1; c = 0
    WideAcc wa;
    dot29_to_wide(c, wa);
    sum = fe_add(sum, redc_wide(wa, P), P);
#pragma unroll
    for (int i = 0; i < 17; ++i) c[i] = 0;
}
template <int EXTRA>
ZK_D void round0_glds_body(const FactorPtrs &fp, uint64_t q, const FieldParams &P, uint64_t *__restrict__ partials) {
    constexpr int NU = 2 + EXTRA;
    const GldsWave w = glds_wave(q, gridDim.x, NU * 4096);
    const uint64_t hi_off = q * 32;
    uint64_t c0[17], c1[17], cL[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) c0[k] = c1[k] = cL[k] = 0;
    Fe sum[3] = {fe_zero(), fe_zero(), fe_zero()};
    Fe sb0 = fe_zero(), sb1 = fe_zero();   // the single-factor term: linear, so it only feeds S(0) and S(1)
    if (w.K) {
#pragma unroll
        for (int f = 0; f < NU; ++f) {
            const uint64_t a = (uint64_t)(uintptr_t)fp.in[f] + w.r0 * 2048;
            glds_rows2(a, a + hi_off, w.voff, w.my_lds + f * 4096);
        }
        int since = 0, lazy = 0;
        for (uint64_t k = 0; k < w.K; ++k) {
            const bool last = k + 1 == w.K;
            const uint64_t next = (w.r0 + (k + 1) * w.rs) * 2048;
            uint32_t la0[9], la1[9], ld0[9];
            round0_glds_wait<NU, 0>(last);
            {
                const Fe lo = glds_elem(w.my, w.lane), hi = glds_elem(w.my + 2048, w.lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) glds_rows2((uint64_t)(uintptr_t)fp.in[0] + next, (uint64_t)(uintptr_t)fp.in[0] + next + hi_off, w.voff, w.my_lds);
                split29(lo.v, la0);
                split29(hi.v, la1);
                const Fe d = fe_sub(hi, lo, P);
                split29(d.v, ld0);
            }
            round0_glds_wait<NU, 1>(last);
            {
                const Fe lo = glds_elem(w.my + 4096, w.lane), hi = glds_elem(w.my + 6144, w.lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) glds_rows2((uint64_t)(uintptr_t)fp.in[1] + next, (uint64_t)(uintptr_t)fp.in[1] + next + hi_off, w.voff, w.my_lds + 4096);
                uint32_t b[9];
                split29(lo.v, b);
                dot29_mac(c0, la0, b);   // S(0): lo0 * lo1
                split29(hi.v, b);
                dot29_mac(c1, la1, b);   // S(1): hi0 * hi1
                const Fe d = fe_sub(hi, lo, P);
                split29(d.v, b);
                dot29_mac(cL, ld0, b);   // leading coefficient: (hi0 - lo0)(hi1 - lo1)
            }
            if constexpr (EXTRA) {
                round0_glds_wait<NU, 2>(last);
                const Fe lo = glds_elem(w.my + 8192, w.lane), hi = glds_elem(w.my + 10240, w.lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) glds_rows2((uint64_t)(uintptr_t)fp.in[2] + next, (uint64_t)(uintptr_t)fp.in[2] + next + hi_off, w.voff, w.my_lds + 8192);
                sb0 = fe_add(sb0, lo, P);
                sb1 = fe_add(sb1, hi, P);
            }
            ++lazy;
            if (++since == 7 || lazy == kMaxLazy || last) {
                dot29_normalise(c0);
                dot29_normalise(c1);
                dot29_normalise(cL);
                since = 0;
            }
            if (lazy == kMaxLazy || last) {   // redc_wide takes the sum of at most kMaxLazy products
                dot29_flush(c0, sum[0], P);
                dot29_flush(c1, sum[1], P);
                dot29_flush(cL, sum[2], P);
                lazy = 0;
            }
        }
    }
    if (EXTRA) {
        sum[0] = fe_add(sum[0], sb0, P);
        sum[1] = fe_add(sum[1], sb1, P);
    }
    block_reduce_store<3>(sum, partials, P);
}
template <int EXTRA>
__global__ __launch_bounds__(kBlock, 2) void k_round0_glds(FactorPtrs fp, uint64_t q, FieldParams P, uint64_t *__restrict__ partials) {
    round0_glds_body<EXTRA>(fp, q, P, partials);
}
template <int EXTRA>
__global__ __launch_bounds__(kBlock, 2) void k_round0_glds_b(BatchOf<RoundSlot> b, uint64_t q, FieldParams P) {
    const RoundSlot &a = b.a[blockIdx.y];
    const FactorPtrs fp = factor_ptrs_of(a.fp);
    round0_glds_body<EXTRA>(fp, q, P, a.partials);
}

// The big fused rounds of a K-table product of degree D = K with SKIP1 and LEAD, EXTRA = 1: plus a single-factor term
// (k_round_kd<K, K, true, EXTRA, true, true>: same sums, same claim workgroup, same half tables).  A unit = one table's four rows of a
// run = 8 KiB = the wave's whole ring: the next unit's DMA is issued the moment this unit's rows are in registers, and is waited for
// with vmcnt(4) -- its eight pieces are older than this unit's four stores.  STORE_NT: the half tables bypass the caches (they are
// larger than what the next round could find there).  Three tables need no scratch here (198-210 registers; k_round_kd spills 12-184
// bytes per lane at two waves per SIMD), which is where the gain comes from: 8-13 % from 2^16 pairs up for K = 3; two tables gain
// nothing at any size and stay on k_round_kd (profiles/r06_glds_sizes.log).
template <int F, int K, int EXTRA, bool STORE_NT>
ZK_D void fused_glds_unit(const GldsWave &w, const FactorPtrs &fp, uint64_t q, uint64_t run, bool last, const Mul29 &r, const FieldParams &P,
                          Fe (&prod)[K + 1], WideAcc (&acc)[K + 1], Fe (&sum_b)[K + 1]) {
    constexpr int D = K, NT = K + EXTRA;
    const Fe c0 = glds_elem(w.my, w.own), c1 = glds_elem(w.my + 2048, w.own), c2 = glds_elem(w.my + 4096, w.own), c3 = glds_elem(w.my + 6144, w.own);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t rowb = q * 32;
    if (F + 1 < NT) {
        const uint64_t a = (uint64_t)(uintptr_t)fp.in[F + 1 < NT ? F + 1 : 0] + run * 2048;
        glds_rows4(a, a + rowb, a + 2 * rowb, a + 3 * rowb, w.voff, w.my_lds);
    } else if (!last) {
        const uint64_t a = (uint64_t)(uintptr_t)fp.in[0] + (run + w.rs) * 2048;
        glds_rows4(a, a + rowb, a + 2 * rowb, a + 3 * rowb, w.voff, w.my_lds);
    }
    const Fe lo = fe_sub(c0, fe_mul29(fe_sub(c0, c2, P), r, P), P);   // prover.rs:64 on rows (0, 2) and (1, 3)
    const Fe hi = fe_sub(c1, fe_mul29(fe_sub(c1, c3, P), r, P), P);
    uint64_t *out = fp.out[F];
    if (STORE_NT) {
        run_store_nt(out + run * 256, w.lane, lo);
        run_store_nt(out + (run * 64 + q) * 4, w.lane, hi);
    } else {
        run_store(out + run * 256, w.lane, lo);
        run_store(out + (run * 64 + q) * 4, w.lane, hi);
    }
    const Fe diff = fe_sub(hi, lo, P);
    Fe v = lo;
#pragma unroll
    for (int t = 0; t <= D; ++t) {
        if (t == 1) {
            v = hi;
            continue;   // SKIP1: S(1) = claim - S(0), derived in the tail
        }
        if (t == D) v = diff;   // LEAD: slot D accumulates the leading coefficient
        else if (t > 1) v = fe_add(v, diff, P);
        if (F == K) {
            if (t < D) sum_b[t] = fe_add(sum_b[t], v, P);   // the single-factor term: linear, nothing for the leading coefficient
        } else if (F == 0) prod[t] = v;
        else if (F < K - 1) prod[t] = ZK_KD_INNER_MUL(prod[t], v, P);
        else wide_mac(acc[t], prod[t].v, v.v);
    }
    if (F + 1 < NT || !last) wait_vm<4>();
}
template <int K, int EXTRA, bool STORE_NT>
ZK_D void fused_glds_body(const FactorPtrs &fp, uint64_t q, const FieldParams &P, const uint64_t *__restrict__ rptr, uint64_t *__restrict__ partials,
                          const ClaimJob &cj) {
    constexpr int NS = K + 1;
    uint32_t nblk = gridDim.x;
    if (cj.out) {   // the claim workgroup (k_round_kd, SKIP1)
        nblk -= 1;
        if (blockIdx.x == nblk) {
            if (threadIdx.x < 64) {
                const Fe claim = claim_eval(cj, P);
                if (threadIdx.x == 0) fe_store(cj.out, 0, claim);
            }
            return;
        }
    }
    const Mul29 r = load_challenge29(rptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of the compiler's is in flight when the counting starts
    const GldsWave w = glds_wave(q, nblk);
    Fe prod[NS], sum[NS], sum_b[NS];
    WideAcc acc[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        sum[t] = fe_zero();
        sum_b[t] = fe_zero();
        wide_zero(acc[t]);
    }
    if (w.K) {
        {
            const uint64_t a = (uint64_t)(uintptr_t)fp.in[0] + w.r0 * 2048, rowb = q * 32;
            glds_rows4(a, a + rowb, a + 2 * rowb, a + 3 * rowb, w.voff, w.my_lds);
        }
        wait_vm<0>();
        for (uint64_t k = 0; k < w.K; ++k) {   // (the host sizes the grid for at most kMaxLazy runs per wave)
            const uint64_t run = w.r0 + k * w.rs;
            const bool last = k + 1 == w.K;
            fused_glds_unit<0, K, EXTRA, STORE_NT>(w, fp, q, run, last, r, P, prod, acc, sum_b);
            fused_glds_unit<1, K, EXTRA, STORE_NT>(w, fp, q, run, last, r, P, prod, acc, sum_b);
            if constexpr (K + EXTRA > 2) fused_glds_unit<2, K, EXTRA, STORE_NT>(w, fp, q, run, last, r, P, prod, acc, sum_b);
            if constexpr (K + EXTRA > 3) fused_glds_unit<3, K, EXTRA, STORE_NT>(w, fp, q, run, last, r, P, prod, acc, sum_b);
        }
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if (t != 1) sum[t] = redc_wide(acc[t], P);
        if (EXTRA) {
#pragma unroll
            for (int t = 0; t < K; ++t)
                if (t != 1) sum[t] = fe_add(sum[t], sum_b[t], P);
        }
    }
    block_reduce_store<NS, true>(sum, partials, P);
}
template <int K, int EXTRA, bool STORE_NT>
__global__ __launch_bounds__(kBlock, 2) void k_round_fused_glds(FactorPtrs fp, uint64_t q, FieldParams P, const uint64_t *__restrict__ rptr,
                                                                 uint64_t *__restrict__ partials, ClaimJob cj) {
    fused_glds_body<K, EXTRA, STORE_NT>(fp, q, P, rptr, partials, cj);
}
template <int K, int EXTRA, bool STORE_NT>
__global__ __launch_bounds__(kBlock, 2) void k_round_fused_glds_b(BatchOf<RoundSlot> b, uint64_t q, FieldParams P) {
    const RoundSlot &a = b.a[blockIdx.y];
    const FactorPtrs fp = factor_ptrs_of(a.fp);
    fused_glds_body<K, EXTRA, STORE_NT>(fp, q, P, a.rptr, a.partials, a.cj);
}

// ---- small fused rounds: one FACTOR per lane, one EVALUATION POINT per lane, four lanes per pair index -----------------
// Between ~2^10 and ~2^15 pairs a round is pure latency: k_round_kd gives a lane the whole pair index (7 dependent-ish
// multiplies for k = 2, 14 for k = 3: 4-8 us of a single wave's issue time) while most of the machine idles.  Here the four
// lanes of a quad share pair index j: lane f folds factor f (2 multiplies), the quad transposes the (factor, point) values
// with DPP quad_perm broadcasts, and lane t forms the product for evaluation point t (k - 1 multiplies).  Same arithmetic
// per pair index, spread over 4x the lanes: the per-lane chain drops to ~1100-1400 instructions.  K + EXTRA <= 4, D <= 3.
template <int K, int D, int EXTRA>
ZK_D void round_quad_body(const FactorPtrs &fp, uint64_t q, const FieldParams &P, const uint64_t *__restrict__ rptr, uint64_t *__restrict__ partials) {
    constexpr int NF = K + EXTRA, NS = D + 1;
    static_assert(NF <= 4 && NS <= 4, "four lanes per pair index");
    __shared__ uint32_t red[kBlock / 64][4][8];
    const Mul29 r = load_challenge29(rptr);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l4 = lane & 3;
    const bool has_factor = l4 < (uint32_t)NF;
    // this lane's factor (lanes >= NF alias factor 0 and keep their hands off memory)
    const uint64_t *in = fp.in[0];
    uint64_t *out = fp.out[0];
#pragma unroll
    for (int f = 1; f < NF; ++f)
        if (l4 == (uint32_t)f) {
            in = fp.in[f];
            out = fp.out[f];
        }
    const uint64_t stride = (uint64_t)gridDim.x * (kBlock / 4);
    uint64_t j = (uint64_t)blockIdx.x * (kBlock / 4) + (threadIdx.x >> 2);
    WideAcc acc;
    wide_zero(acc);
    Fe sum = fe_zero();   // K == 1 products and the extra single-factor term
    Fe cur[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) cur[l] = fe_zero();
    if (j < q && has_factor) {
#pragma unroll
        for (int l = 0; l < 4; ++l) cur[l] = fe_load(in, j + (uint64_t)l * q);
    }
    while (j < q) {
        const uint64_t jn = j + stride;
        Fe v[NS];
        v[0] = fe_sub(cur[0], fe_mul29(fe_sub(cur[0], cur[2], P), r, P), P);   // lo'
        const Fe hi = fe_sub(cur[1], fe_mul29(fe_sub(cur[1], cur[3], P), r, P), P);
        if (has_factor) {
            fe_store(out, j, v[0]);
            fe_store(out, j + q, hi);
            if (jn < q) {
#pragma unroll
                for (int l = 0; l < 4; ++l) cur[l] = fe_load(in, jn + (uint64_t)l * q);
            }
        }
        if constexpr (NS > 1) v[1] = hi;
        if constexpr (NS > 2) {
            const Fe diff = fe_sub(hi, v[0], P);
#pragma unroll
            for (int t = 2; t < NS; ++t) v[t] = fe_add(v[t - 1], diff, P);
        }
        Fe w[NF];
#pragma unroll
        for (int g = 0; g < NF; ++g) w[g] = fe_zero();
        quad_transpose<0, NF, NS>(v, w, l4);
        // lane t: product over the K factors at point t (+ the extra term's value); lanes > D compute on zeros
        if constexpr (K == 1) {
            sum = fe_add(sum, w[0], P);
        } else {
            Fe prod = w[0];
#pragma unroll
            for (int g = 1; g + 1 < K; ++g) prod = ZK_KD_INNER_MUL(prod, w[g], P);
            wide_mac(acc, prod.v, w[K - 1].v);
        }
        if constexpr (EXTRA) sum = fe_add(sum, w[K], P);
        j = jn;
    }
    Fe s = sum;
    if constexpr (K > 1) s = fe_add(redc_wide(acc, P), sum, P);
    // wave sum that keeps the four quad positions apart: every level of fe_wave_sum except the two inside a quad
    {
        Fe a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto x = __builtin_amdgcn_permlane32_swap(s.v[i], s.v[i], false, false);
            a.v[i] = x[0];
            b.v[i] = x[1];
        }
        s = fe_add(a, b, P);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto x = __builtin_amdgcn_permlane16_swap(s.v[i], s.v[i], false, false);
            a.v[i] = x[0];
            b.v[i] = x[1];
        }
        s = fe_add(a, b, P);
        s = fe_add(s, fe_dpp<0x128>(s), P);
        s = fe_add(s, fe_dpp<0x12C>(s), P);
    }
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[wave][lane][i] = s.v[i];
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)NS) {
        Fe tot = fe_zero();
        for (int wv = 0; wv < kBlock / 64; ++wv) {
            Fe o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = red[wv][threadIdx.x][i];
            tot = fe_add(tot, o, P);
        }
        fe_store(partials, (uint64_t)blockIdx.x * NS + threadIdx.x, tot);
    }
}
template <int K, int D, int EXTRA>
__global__ __launch_bounds__(kBlock) void k_round_quad(FactorPtrs fp, uint64_t q, FieldParams P, const uint64_t *__restrict__ rptr,
                                                       uint64_t *__restrict__ partials) {
    round_quad_body<K, D, EXTRA>(fp, q, P, rptr, partials);
}
template <int K, int D, int EXTRA>
__global__ __launch_bounds__(kBlock) void k_round_quad_b(BatchOf<RoundSlot> b, uint64_t q, FieldParams P) {
    const RoundSlot &a = b.a[blockIdx.y];
    const FactorPtrs fp = factor_ptrs_of(a.fp);
    round_quad_body<K, D, EXTRA>(fp, q, P, a.rptr, a.partials);
}

// Generic-degree fallback: one evaluation point t per launch (any D up to 255, any k <= kMaxFactors).
__global__ __launch_bounds__(kBlock) void k_round_single_t(FactorPtrs fp, int k, uint64_t q, FieldParams P, Fe tval,
                                                           uint64_t *__restrict__ partials) {
    Fe sum[1] = {fe_zero()};
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < q; j += stride) {
        Fe prod = fe_zero();
        for (int f = 0; f < k; ++f) {
            const Fe lo = fe_load(fp.in[f], j), hi = fe_load(fp.in[f], j + q);
            const Fe v = fe_sub(lo, fe_mul(tval, fe_sub(lo, hi, P), P), P);
            prod = (f == 0) ? v : fe_mul(prod, v, P);
        }
        sum[0] = fe_add(sum[0], prod, P);
    }
    block_reduce_store<1>(sum, partials, P);
}

}  // namespace zk
