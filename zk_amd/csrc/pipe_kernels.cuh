// pipe_kernels.cuh -- the small and middle rounds of the prover with the transcript step OFF the critical path.
//
// A sumcheck round (sumcheck/src/prover.rs:44-68) is  sums S_s  ->  absorb, squeeze r_s  ->  fold at r_s,  and the squeeze is
// serial (one Keccak permutation + field conversions, ~6 us on one wave).  Below ~2^15 pairs the sums take about as long, and
// doing them one after the other is what made these rounds cost ~18 us each.  Here the sums of round s are computed BEFORE
// the challenge r_{s-1} that their table depends on is known, as a polynomial in it:
//
//   table of round s:   T_s[x] = T_{s-1}[x] + r_{s-1} * (T_{s-1}[x + half] - T_{s-1}[x])          (evaluation_form.rs:68)
//   so every factor value at evaluation point t is  u_f(t) + r_{s-1} * v_f(t),  u from the low half, u + v from the high half
//   S_s(t) = sum_x [ prod_{f<K} (u_f + r v_f)  (+ the single-factor term u_B + r v_B) ]  = polynomial of degree K in r.
//
// The work lanes evaluate that polynomial at K+1 fixed nodes, E_s(t; rho) for rho in {0, inf, 1, -1} (rho = inf is the
// leading coefficient prod v_f; K = 1 uses {0, inf}, K = 2 adds 1, K = 3 adds -1), while ONE wave runs the transcript step
// of round s-1; when r_{s-1} appears, S_s(t) is two multiplications away.  Field arithmetic is exact, so the interpolated
// S_s(t) is the same canonical element the reference's sum produces (bit-exact; every prover test compares with the oracle).
//
// k_round_pipe: a launch = [work blocks: (fold T_{s-2} -> T_{s-1} at r_{s-2} +) E_s partials] + [one transcript block:
// closes round s-1 from its partials (plain sums, or E_{s-1} partials + r_{s-2})].  Nothing waits on another block: the
// transcript block's output (r_{s-1}) is only read by the NEXT launch.  Challenges live in two alternating slots so the
// work blocks read r_{s-2} while the transcript block writes r_{s-1}.
// k_finish_pipe: all remaining rounds in one 1024-thread workgroup with the same overlap (wave 0 = transcript, the other
// waves fold in LDS and prepare the next round's E), one barrier per round.
#pragma once
#include "common.cuh"
#include "quad.cuh"
#include "transcript.cuh"

namespace zk {

constexpr int kPipeNodeZero = 0, kPipeNodeInf = 1, kPipeNodeOne = 2, kPipeNodeMinus = 3;
constexpr int kPipeThreads = 256;
constexpr int kFinishPipeThreads = 1024;

template <int K, int D, int EXTRA>
struct PipeShape {
    static_assert(K >= 1 && K <= 3 && D >= 1 && D <= 3 && K + EXTRA <= 4 && (EXTRA == 0 || K >= 2), "shape");
    static constexpr int NF = K + EXTRA, NS = D + 1, NR = K + 1, NE = NS * NR;
};

// Per-lane accumulators of the work lanes (lane l4 = evaluation point t of its quad).
template <int K, int D, int EXTRA, bool PLAIN>
struct PipeAcc {
    WideAcc e[K >= 2 ? K + 1 : 1];      // unreduced E(t; node) products (K >= 2)
    WideAcc s[(PLAIN && K >= 2) ? 1 : 1];   // plain S(t) of the table itself (fresh / pending entry of the finisher)
    Fe su, sw;                          // K == 1: sum u, sum w of factor 0;  EXTRA: of the single-factor term
    Fe ps;                              // PLAIN, K == 1 or EXTRA: modular part of the plain sum
};
template <int K, int D, int EXTRA, bool PLAIN>
ZK_D void pipe_acc_zero(PipeAcc<K, D, EXTRA, PLAIN> &A) {
#pragma unroll
    for (int i = 0; i < (K >= 2 ? K + 1 : 1); ++i) wide_zero(A.e[i]);
    wide_zero(A.s[0]);
    A.su = fe_zero();
    A.sw = fe_zero();
    A.ps = fe_zero();
}

// value of a0 + t*(a1 - a0) on the lane whose evaluation point is t = l4 (0..3)
ZK_D Fe pipe_point(const Fe &a0, const Fe &a1, uint32_t l4, const FieldParams &P) {
    const Fe d = fe_sub(a1, a0, P);
    Fe v = l4 == 0 ? a0 : a1;
    const Fe v2 = fe_add(a1, d, P);
    v = l4 >= 2 ? v2 : v;
    const Fe v3 = fe_add(v2, d, P);
    v = l4 >= 3 ? v3 : v;
    return v;
}

// One pair index of round s on a quad: lane f holds a[0..4) = T_{s-1}[j + l*q] of factor f (already folded); every lane ends
// up with its evaluation point's contribution added to A.  EMIT: the E(t; rho) part;  PLAIN: the plain sums of T_{s-1} itself
// (its pairs are (a0, a2) and (a1, a3)).
template <int K, int D, int EXTRA, bool PLAIN, int F = 0>
ZK_D void pipe_products(PipeAcc<K, D, EXTRA, PLAIN> &A, const Fe (&a)[4], uint32_t l4, bool emit, const FieldParams &P, Fe (&pe)[4],
                        Fe (&pp)[2]) {
    constexpr int NF = K + EXTRA;
    const Fe b0 = quad_bcast<F>(a[0]), b1 = quad_bcast<F>(a[1]), b2 = quad_bcast<F>(a[2]), b3 = quad_bcast<F>(a[3]);
    if (emit) {
        const Fe u = pipe_point(b0, b1, l4, P), w = pipe_point(b2, b3, l4, P);
        if constexpr (F < K) {
            const Fe v = fe_sub(w, u, P);
            if constexpr (K == 1) {
                A.su = fe_add(A.su, u, P);
                A.sw = fe_add(A.sw, w, P);
            } else if constexpr (F == 0) {
                pe[kPipeNodeZero] = u;
                pe[kPipeNodeInf] = v;
                pe[kPipeNodeOne] = w;
                if constexpr (K == 3) pe[kPipeNodeMinus] = fe_sub(u, v, P);
            } else if constexpr (F < K - 1) {
                pe[kPipeNodeZero] = fe_mul(pe[kPipeNodeZero], u, P);
                pe[kPipeNodeInf] = fe_mul(pe[kPipeNodeInf], v, P);
                pe[kPipeNodeOne] = fe_mul(pe[kPipeNodeOne], w, P);
                if constexpr (K == 3) pe[kPipeNodeMinus] = fe_mul(pe[kPipeNodeMinus], fe_sub(u, v, P), P);
            } else {
                wide_mac(A.e[kPipeNodeZero], pe[kPipeNodeZero].v, u.v);
                wide_mac(A.e[kPipeNodeInf], pe[kPipeNodeInf].v, v.v);
                wide_mac(A.e[kPipeNodeOne], pe[kPipeNodeOne].v, w.v);
                if constexpr (K == 3) {
                    const Fe m = fe_sub(u, v, P);
                    wide_mac(A.e[kPipeNodeMinus], pe[kPipeNodeMinus].v, m.v);
                }
            }
        } else {   // the single-factor term: linear in r, only sums
            A.su = fe_add(A.su, u, P);
            A.sw = fe_add(A.sw, w, P);
        }
    }
    if constexpr (PLAIN) {
        const Fe x = pipe_point(b0, b2, l4, P), y = pipe_point(b1, b3, l4, P);   // the table's own pairs (j, j+2q), (j+q, j+3q)
        if constexpr (F < K) {
            if constexpr (K == 1) {
                A.ps = fe_add(A.ps, fe_add(x, y, P), P);
            } else if constexpr (F == 0) {
                pp[0] = x;
                pp[1] = y;
            } else if constexpr (F < K - 1) {
                pp[0] = fe_mul(pp[0], x, P);
                pp[1] = fe_mul(pp[1], y, P);
            } else {
                wide_mac(A.s[0], pp[0].v, x.v);
                wide_mac(A.s[0], pp[1].v, y.v);
            }
        } else {
            A.ps = fe_add(A.ps, fe_add(x, y, P), P);
        }
    }
    if constexpr (F + 1 < NF) pipe_products<K, D, EXTRA, PLAIN, F + 1>(A, a, l4, emit, P, pe, pp);
}

// Close the accumulators: out[node] = this lane's E(t; node) (NR values), plain = its plain S(t).
template <int K, int D, int EXTRA, bool PLAIN>
ZK_D void pipe_acc_close(const PipeAcc<K, D, EXTRA, PLAIN> &A, Fe (&out)[K + 1], Fe &plain, const FieldParams &P) {
    if constexpr (K == 1) {
        out[kPipeNodeZero] = A.su;
        out[kPipeNodeInf] = fe_sub(A.sw, A.su, P);
    } else {
#pragma unroll
        for (int i = 0; i <= K; ++i) out[i] = redc_wide(A.e[i], P);
        if constexpr (EXTRA) {
            out[kPipeNodeZero] = fe_add(out[kPipeNodeZero], A.su, P);
            out[kPipeNodeOne] = fe_add(out[kPipeNodeOne], A.sw, P);
            if constexpr (K == 3) out[kPipeNodeMinus] = fe_add(out[kPipeNodeMinus], fe_sub(fe_add(A.su, A.su, P), A.sw, P), P);   // u - v = 2u - w
        }
    }
    if constexpr (PLAIN) {
        if constexpr (K == 1) plain = A.ps;
        else plain = fe_add(redc_wide(A.s[0], P), A.ps, P);
    } else {
        plain = fe_zero();
    }
}

// wave sum that keeps the four quad positions apart: every level of fe_wave_sum except the two inside a quad
ZK_D Fe quad_wave_sum(Fe s, const FieldParams &P) {
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
    return s;
}

// S(t) from E(t; .) once the challenge r is known.  r29 = prepared multiplier form of r; inv2 = 1/2 (K == 3).
template <int K>
ZK_D Fe pipe_eval(const Fe (&e)[K + 1], const Mul29 &r29, const Fe &inv2, const FieldParams &P) {
    if constexpr (K == 1) {
        return fe_add(e[kPipeNodeZero], fe_mul29(e[kPipeNodeInf], r29, P), P);
    } else if constexpr (K == 2) {
        const Fe c1 = fe_sub(fe_sub(e[kPipeNodeOne], e[kPipeNodeZero], P), e[kPipeNodeInf], P);
        return fe_add(e[kPipeNodeZero], fe_mul29(fe_add(c1, fe_mul29(e[kPipeNodeInf], r29, P), P), r29, P), P);
    } else {
        // p(1) + p(-1) = 2(c0 + c2),  p(1) - p(-1) = 2(c1 + c3)
        const Fe h = fe_mul(fe_add(e[kPipeNodeOne], e[kPipeNodeMinus], P), inv2, P), g = fe_mul(fe_sub(e[kPipeNodeOne], e[kPipeNodeMinus], P), inv2, P);
        const Fe c2 = fe_sub(h, e[kPipeNodeZero], P), c1 = fe_sub(g, e[kPipeNodeInf], P);
        Fe acc = fe_add(c2, fe_mul29(e[kPipeNodeInf], r29, P), P);
        acc = fe_add(c1, fe_mul29(acc, r29, P), P);
        return fe_add(e[kPipeNodeZero], fe_mul29(acc, r29, P), P);
    }
}

// ---- the transcript block ----------------------------------------------------------------------------------------------
struct PipeTailArgs {
    const uint64_t *partials;   // per-block partials of the round being closed: [block][n_in] elements
    uint32_t nblocks;
    uint32_t n_in;              // values per block: NS (mode 0; SKIP1 leaves slot 1 unwritten) or NS * NR (mode 1)
    int mode;                   // 0: plain sums   1: E(t; rho), evaluated at *chal_in
    const uint64_t *chal_in;    // challenge record of the previous round (mode 1, and SKIP1's derive)
    uint64_t *chal_out;         // challenge record this round's challenge is published in
    WordSponge *sponge;
    uint64_t *out_rp, *out_ch;
    Fe inv2;
    TailDerive dv;              // mode 0 after a SKIP1 round kernel
};
// Reduce the block partials: value idx (< n_in <= 16) summed over the blocks -> red[idx].  256 threads: thread (idx =
// tid % 16, slice = tid / 16) adds its share, lanes 16/32 apart combine on the VALU, the four waves through LDS.
ZK_D void pipe_reduce_partials(const uint64_t *__restrict__ partials, uint32_t nblocks, uint32_t n_in, Fe *red /* LDS, 16 */, Fe (*stage)[16] /* LDS [4][16] */,
                               const FieldParams &P) {
    const uint32_t tid = threadIdx.x, idx = tid & 15, slice = tid >> 4, lane = tid & 63, wave = tid >> 6;
    Fe s = fe_zero();
    if (idx < n_in)
        for (uint32_t b = slice; b < nblocks; b += kPipeThreads / 16) s = fe_add(s, fe_load(partials, (uint64_t)b * n_in + idx), P);
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
    }
    if (lane < 16) stage[wave][lane] = s;
    __syncthreads();
    if (tid < 16) red[tid] = fe_add(fe_add(stage[0][tid], stage[1][tid], P), fe_add(stage[2][tid], stage[3][tid], P), P);
    __syncthreads();
}

template <int K, int D>
ZK_D void pipe_tail_block(const PipeTailArgs &ta, const FieldParams &P) {
    constexpr int NS = D + 1, NR = K + 1;
    __shared__ Fe red[16];
    __shared__ Fe stage[4][16];
    __shared__ Fe fin[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = {0, 0};
    if (wave0) sp = lane_sponge_load(ta.sponge, L);   // in flight while the partials are reduced
    pipe_reduce_partials(ta.partials, ta.nblocks, ta.n_in, red, stage, P);
    if (wave0) {
        Fe s = fe_zero();
        if (ta.mode == 1) {
            const Mul29 r29 = load_challenge29(ta.chal_in);
            Fe e[NR];
            const uint32_t t = lane < (uint32_t)NS ? lane : 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) e[i] = red[t * NR + i];
            s = pipe_eval<K>(e, r29, ta.inv2, P);
        } else {
            s = red[lane < (uint32_t)NS ? lane : 0];
            if (ta.dv.prev_rp) {
                // S(1) = S_prev(r_prev) - S(0): the round kernel left the t = 1 products out (k_round_kd SKIP1)
                const Fe r = fe_load(ta.chal_in, 0);
                Fe term = fe_zero();
                if (lane < (uint32_t)NS) {
                    Fe wt = ta.dv.w[0];
#pragma unroll
                    for (int i = 1; i <= kMaxSkipDegree; ++i)
                        if (lane == (uint32_t)i) wt = ta.dv.w[i];
                    term = fe_mul(fe_load(ta.dv.prev_rp, lane), wt, P);
                    Fe node = fe_zero();
                    const Fe one = fe_one(P);
                    for (uint32_t u = 0; u < (uint32_t)NS; ++u) {
                        if (u != lane) term = fe_mul(term, fe_sub(r, node, P), P);
                        node = fe_add(node, one, P);
                    }
                }
                const Fe claim = fe_wave_sum(term, P, 8);   // valid in lane 0
                Fe c0;
#pragma unroll
                for (int i = 0; i < 8; ++i) c0.v[i] = __builtin_amdgcn_readfirstlane(claim.v[i]);
                const Fe s0 = red[0];
                if (lane == 1) s = fe_sub(c0, s0, P);
            }
        }
        if (lane < (uint32_t)NS) {
            fin[lane] = s;
            fe_store(ta.out_rp, lane, s);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): fin[] written by this wave is read back by transcript_step
        Mul29 ch29;
        const Fe ch = transcript_step(sp, L, fin, NS, P, ch29);
        publish_challenge(ta.chal_out, ta.out_ch, ch, ch29, L.lane);
        lane_sponge_store(ta.sponge, sp, L);
    }
}

// ---- the pipelined round kernel ------------------------------------------------------------------------------------------
// q = pairs of round s.  Work blocks 0 .. gridDim.x-2, transcript block gridDim.x-1.
//   FOLD : fp.in = tables of round s-2 (8q elements), folded at *chal_fold (= r_{s-2}) into fp.out (tables of round s-1, 4q
//          elements, may alias fp.in: a lane reads and writes only its own positions);
//   !FOLD: fp.in = tables of round s-1 (4q elements), already materialised.
//   emit : compute the E_s partials (0: fold only -- the launch that leaves the pipeline).
template <int K, int D, int EXTRA, bool FOLD>
__global__ __launch_bounds__(kPipeThreads) void k_round_pipe(FactorPtrs fp, uint64_t q, int emit, FieldParams P, const uint64_t *__restrict__ chal_fold,
                                                             uint64_t *__restrict__ e_partials, PipeTailArgs ta) {
    using S = PipeShape<K, D, EXTRA>;
    if (blockIdx.x == gridDim.x - 1) {
        pipe_tail_block<K, D>(ta, P);
        return;
    }
    __shared__ uint32_t redw[kPipeThreads / 64][4][S::NR][8];
    Mul29 r = {};
    if (FOLD) r = load_challenge29(chal_fold);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l4 = lane & 3;
    const bool has_factor = l4 < (uint32_t)S::NF;
    const uint64_t *in = fp.in[0];
    uint64_t *out = fp.out[0];
#pragma unroll
    for (int f = 1; f < S::NF; ++f)
        if (l4 == (uint32_t)f) {
            in = fp.in[f];
            out = fp.out[f];
        }
    constexpr int NL = FOLD ? 8 : 4;
    const uint64_t nwork = gridDim.x - 1, stride = nwork * (kPipeThreads / 4);
    uint64_t j = (uint64_t)blockIdx.x * (kPipeThreads / 4) + (threadIdx.x >> 2);
    PipeAcc<K, D, EXTRA, false> A;
    pipe_acc_zero(A);
    Fe cur[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) cur[l] = fe_zero();
    if (j < q && has_factor) {
#pragma unroll
        for (int l = 0; l < NL; ++l) cur[l] = fe_load(in, j + (uint64_t)l * q);
    }
    while (j < q) {
        const uint64_t jn = j + stride;
        Fe a[4];
        if constexpr (FOLD) {
#pragma unroll
            for (int l = 0; l < 4; ++l) a[l] = fe_sub(cur[l], fe_mul29(fe_sub(cur[l], cur[l + 4], P), r, P), P);   // evaluation_form.rs:68
            if (has_factor) {
#pragma unroll
                for (int l = 0; l < 4; ++l) fe_store(out, j + (uint64_t)l * q, a[l]);
            }
        } else {
#pragma unroll
            for (int l = 0; l < 4; ++l) a[l] = cur[l];
        }
        if (has_factor && jn < q) {
#pragma unroll
            for (int l = 0; l < NL; ++l) cur[l] = fe_load(in, jn + (uint64_t)l * q);
        }
        Fe pe[4], pp[2];
        pipe_products<K, D, EXTRA, false>(A, a, l4, emit != 0, P, pe, pp);
        j = jn;
    }
    if (!emit) return;
    Fe e[S::NR], plain;
    pipe_acc_close<K, D, EXTRA, false>(A, e, plain, P);
#pragma unroll
    for (int i = 0; i < S::NR; ++i) {
        const Fe s = quad_wave_sum(e[i], P);
        if (lane < 4) {
#pragma unroll
            for (int w = 0; w < 8; ++w) redw[wave][lane][i][w] = s.v[w];
        }
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)S::NE) {
        const uint32_t t = threadIdx.x / S::NR, i = threadIdx.x % S::NR;
        Fe tot = fe_zero();
        for (int wv = 0; wv < kPipeThreads / 64; ++wv) {
            Fe o;
#pragma unroll
            for (int w = 0; w < 8; ++w) o.v[w] = redw[wv][t][i][w];
            tot = fe_add(tot, o, P);
        }
        fe_store(e_partials, (uint64_t)blockIdx.x * S::NE + threadIdx.x, tot);
    }
}

}  // namespace zk
