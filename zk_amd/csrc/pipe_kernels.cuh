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
//
// Work layout ("hex"): these rounds are latency-bound, so ONE pair index gets the 16 lanes of a DPP row and every lane does
// one multiplication per phase, exchanging through a per-row LDS buffer (same wave: LDS operations execute in order):
//   A  lane (factor f, slot l): loads T[j + l*q] (and T[j + (l+4)*q], folding them at the challenge: 1 multiply), the four
//      values a_l = table of the round before the prepared one, restricted to the pair index;
//   B1 lane (factor f, point t): u = a0 + t(a1 - a0) (value if the pending challenge were 0), w = a2 + t(a3 - a2) (if it
//      were 1), and from them the factor's value at every node: u, w - u (inf), w, 2u - w (-1);
//   B2 lane (point t, node rho): the product over the factors (1 multiply, unreduced accumulation).
#pragma once
#include "common.cuh"
#include "quad.cuh"
#include "transcript.cuh"
#include "pipe_args.hpp"

namespace zk {

constexpr int kPipeNodeZero = 0, kPipeNodeInf = 1, kPipeNodeOne = 2, kPipeNodeMinus = 3;
constexpr int kPipeThreads = 256;
constexpr int kFinishPipeThreads = 256;

template <int K, int D, int EXTRA>
struct PipeShape {
    static_assert(K >= 1 && K <= 3 && D >= 1 && D <= 3 && K + EXTRA <= 4 && (EXTRA == 0 || K >= 2), "shape");
    static constexpr int NF = K + EXTRA, NS = D + 1, NR = K + 1, NE = NS * NR;
};


// ---- the work lanes --------------------------------------------------------------------------------------------------------
template <int K, int D, int EXTRA, bool PLAIN>
struct HexCfg {
    static constexpr int NF = K + EXTRA, NS = D + 1, NR = K + 1;
    static constexpr int XW = NR + (PLAIN ? 2 : 0);          // values a (factor, point) lane publishes
    static constexpr int ROW_FE = NF * NS * XW > NF * 4 ? NF * NS * XW : NF * 4;   // the a's alias the x's (read before written)
    static constexpr int ROW_BYTES = ROW_FE * 32;
};
template <bool PLAIN>
struct HexAcc {
    WideAcc e;            // K >= 2: unreduced products of this lane's (t, node)
    WideAcc p;            // PLAIN: products of the table's own pairs (node 0: pairs (a0, a2); node 1: pairs (a1, a3))
    Fe s, ps;             // K == 1 / the single-factor term: modular sums
};
template <bool PLAIN>
ZK_D void hex_acc_zero(HexAcc<PLAIN> &A) {
    wide_zero(A.e);
    wide_zero(A.p);
    A.s = fe_zero();
    A.ps = fe_zero();
}
ZK_D Fe fe_and(const Fe &a, uint32_t mask) {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = a.v[i] & mask;
    return o;
}
// a0 + t * (a1 - a0) for a lane-dependent t in 0..D, as D masked additions (no selects: the compiler turns chains of
// selects over lane-dependent conditions into scratch-memory tables)
template <int D>
ZK_D Fe hex_point(const Fe &a0, const Fe &a1, uint32_t t, const FieldParams &P) {
    const Fe d = fe_sub(a1, a0, P);
    Fe v = a0;
#pragma unroll
    for (int s = 1; s <= D; ++s) v = fe_add(v, fe_and(d, t >= (uint32_t)s ? 0xffffffffu : 0u), P);
    return v;
}
ZK_D Fe lds_fe_load(const Fe *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1];
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
ZK_D void lds_fe_store(Fe *p, const Fe &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// One pair index j (of q) on one 16-lane row.  src / dst: this lane's phase-A factor table (lanes of quad f: factor f).
// live: j < q (dead rows carry zeros through every phase and add nothing).
template <int K, int D, int EXTRA, bool FOLD, bool PLAIN, bool EMIT>
ZK_D void hex_step(HexAcc<PLAIN> &A, const uint64_t *src, uint64_t *dst, uint64_t j, uint64_t q, bool live, const Mul29 &r, const FieldParams &P,
                   Fe *row /* LDS: this row's exchange buffer */, uint32_t lane16) {
    using C = HexCfg<K, D, EXTRA, PLAIN>;
    const uint32_t hi2 = lane16 >> 2, lo2 = lane16 & 3;
    // ---- A: (factor hi2, slot lo2) ----
    {
        Fe a = fe_zero();
        if (live && hi2 < (uint32_t)C::NF) {
            const Fe x = fe_load(src, j + (uint64_t)lo2 * q);
            if constexpr (FOLD) {
                const Fe y = fe_load(src, j + (uint64_t)(lo2 + 4) * q);
                a = fe_sub(x, fe_mul29(fe_sub(x, y, P), r, P), P);   // evaluation_form.rs:68
                fe_store(dst, j + (uint64_t)lo2 * q, a);
            } else {
                a = x;
            }
        }
        if (hi2 < (uint32_t)C::NF) lds_fe_store(row + hi2 * 4 + lo2, a);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- B1: (factor hi2, point lo2) ----
    {
        const bool role = hi2 < (uint32_t)C::NF && lo2 < (uint32_t)C::NS;
        const uint32_t f = role ? hi2 : 0, t = lo2;
        const Fe a0 = lds_fe_load(row + f * 4 + 0), a1 = lds_fe_load(row + f * 4 + 1), a2 = lds_fe_load(row + f * 4 + 2),
                 a3 = lds_fe_load(row + f * 4 + 3);
        __builtin_amdgcn_wave_barrier();   // every lane has read its a's before anybody overwrites them with x's
        Fe *x = row + (f * C::NS + (role ? t : 0)) * C::XW;
        if constexpr (EMIT) {
            const Fe u = hex_point<D>(a0, a1, t, P), w = hex_point<D>(a2, a3, t, P);
            Fe v = fe_sub(w, u, P);
            if constexpr (EXTRA) v = fe_and(v, f == (uint32_t)K ? 0u : 0xffffffffu);   // the single-factor term has no r^K part
            if (role) {
                lds_fe_store(x + kPipeNodeZero, u);
                lds_fe_store(x + kPipeNodeInf, v);
                if constexpr (K >= 2) lds_fe_store(x + kPipeNodeOne, w);
                if constexpr (K == 3) lds_fe_store(x + kPipeNodeMinus, fe_sub(fe_add(u, u, P), w, P));   // u - v = 2u - w
            }
        }
        if constexpr (PLAIN) {
            const Fe xp = hex_point<D>(a0, a2, t, P), yp = hex_point<D>(a1, a3, t, P);   // the table's own pairs (j, j+2q), (j+q, j+3q)
            if (role) {
                lds_fe_store(x + C::NR, xp);
                lds_fe_store(x + C::NR + 1, yp);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- B2: (point hi2, node lo2) ----
    {
        const uint32_t t = hi2 < (uint32_t)C::NS ? hi2 : 0, node = lo2 < (uint32_t)C::NR ? lo2 : 0;
        const Fe *x = row + t * C::XW + node;
        constexpr int FS = C::NS * C::XW;   // stride between factors
        if constexpr (EMIT) {
            const Fe x0 = lds_fe_load(x);
            if constexpr (K == 1) {
                A.s = fe_add(A.s, x0, P);
            } else if constexpr (K == 2) {
                const Fe x1 = lds_fe_load(x + FS);
                wide_mac(A.e, x0.v, x1.v);
            } else {
                const Fe x1 = lds_fe_load(x + FS), x2 = lds_fe_load(x + 2 * FS);
                const Fe m = fe_mul(x0, x1, P);
                wide_mac(A.e, m.v, x2.v);
            }
            if constexpr (EXTRA) A.s = fe_add(A.s, lds_fe_load(x + K * FS), P);
        }
        if constexpr (PLAIN) {
            const Fe *y = row + t * C::XW + C::NR + (lo2 & 1);
            const Fe y0 = lds_fe_load(y);
            if constexpr (K == 1) {
                A.ps = fe_add(A.ps, y0, P);
            } else if constexpr (K == 2) {
                const Fe y1 = lds_fe_load(y + FS);
                wide_mac(A.p, y0.v, y1.v);
            } else {
                const Fe y1 = lds_fe_load(y + FS), y2 = lds_fe_load(y + 2 * FS);
                const Fe m = fe_mul(y0, y1, P);
                wide_mac(A.p, m.v, y2.v);
            }
            if constexpr (EXTRA) A.ps = fe_add(A.ps, lds_fe_load(y + K * FS), P);
        }
    }
    __builtin_amdgcn_wave_barrier();   // the row buffer is free for the next pair index
}
// this lane's E(t; node) (lane16 = 4t + node) and, PLAIN, its share of the plain S(t) (nodes 0 and 1 hold the two halves)
template <int K, bool PLAIN>
ZK_D void hex_close(const HexAcc<PLAIN> &A, Fe &e, Fe &plain, const FieldParams &P) {
    if constexpr (K == 1) e = A.s;
    else e = fe_add(redc_wide(A.e, P), A.s, P);
    if constexpr (PLAIN) {
        if constexpr (K == 1) plain = A.ps;
        else plain = fe_add(redc_wide(A.p, P), A.ps, P);
    } else {
        plain = fe_zero();
    }
}
// sum over the four rows of a wave: lanes 0..15 end up with the totals of their (t, node)
ZK_D Fe hex_rows_sum(Fe s, const FieldParams &P) {
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
    return fe_add(a, b, P);
}

// S(t) from E(t; .) once the challenge r is known, in CANONICAL form, with the fewest DEPENDENT multiplications: the transcript
// wave is the prover's serial chain and a lone wave pays ~0.8 us per carry-free multiplication whatever else it could overlap.
//   * a Montgomery-form value times the prepared CANONICAL challenge (rc29) is a canonical product, so the LAST Horner step
//     delivers canonical S directly -- no Montgomery -> canonical conversion in front of the absorb;
//   * everything that does not depend on a previous product shares ONE multiplication, lane-parallel: row 0 (lanes t) takes the
//     first Horner product e_inf * r, row 1 (lanes 16 + t) the canonical e_0 (times 2^5: a Montgomery reduction), rows 2 / 3 the
//     halvings (e_1 +- e_-1) / 2 of K = 3; their per-lane multiplier comes from a small LDS table.
// Dependent multiplications per round: K (plus the challenge's own conversion) instead of K + 2 (+ 1 for K = 3's halvings).
// Exact field arithmetic: the canonical S is the reference's element (every prover test compares with the oracle).
struct alignas(16) PipeEvalLds {
    uint32_t cop[3][12];   // per-row multipliers, 48-byte slots (read 16 bytes at a time): [0] the challenge (prepared, Montgomery;
                           // K = 1: canonical), [1] the constant 2^5 (a Montgomery reduction), [2] 1/2
    Fe xch[3][4];          // what rows 1..3 hand to row 0: [0] canonical e_0, [1] (e_1 + e_-1)/2, [2] (e_1 - e_-1)/2
};
ZK_D Mul29 lds_mul29_load(const uint32_t *slot) {
    const uint4 *q = reinterpret_cast<const uint4 *>(slot);
    const uint4 a = q[0], b = q[1];
    Mul29 m = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, slot[8]}};
    return m;
}
ZK_D void pipe_eval_set(PipeEvalLds &W, int slot, const Mul29 &m) {
#pragma unroll
    for (int i = 0; i < 9; ++i) W.cop[slot][i] = m.l[i];
}
// wave-uniform constants of the table (any lanes may call it; lanes 1, 2 write)
ZK_D void pipe_eval_consts(PipeEvalLds &W, const PipeConsts &pc, uint32_t lane) {
    if (lane == 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) W.cop[1][i] = i == 0 ? 32u : 0u;
    }
    if (lane == 2) pipe_eval_set(W, 2, pc.inv2p);
}
// red: E totals in LDS, [t * STRIDE + node].  rh29 / rc29: the two prepared forms of the challenge (wave-uniform).  Every lane of
// the wave executes this; the result is valid on lanes t < D + 1.  Callers make W.cop[0] = (K == 1 ? rc29 : rh29) beforehand.
template <int K, int D, int STRIDE = K + 1>
ZK_D Fe pipe_eval_canon(const Fe *red, PipeEvalLds &W, const Mul29 &rh29, const Mul29 &rc29, uint32_t lane, const FieldParams &P) {
    constexpr int NS = D + 1, NR = K + 1;
    const uint32_t t = (lane & 15) < (uint32_t)NS ? (lane & 15) : 0, row = lane >> 4;
    Fe e[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) e[i] = red[t * STRIDE + i];
    // the first, shared multiplication
    Fe a = row == 0 ? e[kPipeNodeInf] : e[kPipeNodeZero];
    if constexpr (K == 3) {
        const Fe sm = fe_add(e[kPipeNodeOne], e[kPipeNodeMinus], P), df = fe_sub(e[kPipeNodeOne], e[kPipeNodeMinus], P);
#pragma unroll
        for (int i = 0; i < 8; ++i) a.v[i] = row == 2 ? sm.v[i] : (row == 3 ? df.v[i] : a.v[i]);
    }
    const Mul29 cm = lds_mul29_load(W.cop[row == 0 ? 0 : (row == 1 ? 1 : 2)]);
    const Fe m1 = fe_mul29(a, cm, P);
    if (row >= 1 && row <= (K == 3 ? 3u : 1u) && (lane & 15) < (uint32_t)NS) lds_fe_store(&W.xch[row - 1][t], m1);
    __builtin_amdgcn_wave_barrier();   // same wave: LDS operations execute in order
    const Fe e0c = lds_fe_load(&W.xch[0][t]);
    if constexpr (K == 1) {
        return fe_add(e0c, m1, P);                                                        // e0 + r * e_inf
    } else if constexpr (K == 2) {
        const Fe c1 = fe_sub(fe_sub(e[kPipeNodeOne], e[kPipeNodeZero], P), e[kPipeNodeInf], P);
        return fe_add(e0c, fe_mul29(fe_add(c1, m1, P), rc29, P), P);                      // e0 + r (c1 + r e_inf)
    } else {
        // p(1) + p(-1) = 2 (c0 + c2),  p(1) - p(-1) = 2 (c1 + c3)
        const Fe h = lds_fe_load(&W.xch[1][t]), g = lds_fe_load(&W.xch[2][t]);
        const Fe c2 = fe_sub(h, e[kPipeNodeZero], P), c1 = fe_sub(g, e[kPipeNodeInf], P);
        const Fe acc = fe_add(c1, fe_mul29(fe_add(c2, m1, P), rh29, P), P);
        return fe_add(e0c, fe_mul29(acc, rc29, P), P);                                    // e0 + r (c1 + r (c2 + r e_inf))
    }
}
// the proof records Montgomery form: canonical -> Montgomery is one multiplication by R^2 (any wave, off the serial chain)
ZK_D Fe fe_from_canonical29(const Fe &c, const FieldParams &P) {
    Mul29 k0;
#pragma unroll
    for (int i = 0; i < 9; ++i) k0.l[i] = P.r2_29[i];
    return fe_mul29(c, k0, P);
}

// Write-through / L1-bypassing element accesses (global_store / global_load ... sc1 = relaxed agent-scope atomics, 8 bytes each):
// the EXPERIMENTAL fence-free hand-off of the block partials (A/B builds with -DZK_PIPE_SC1_HANDOFF=1 only; see k_round_pipe)
ZK_D void fe_store_sc1(uint64_t *base, uint64_t idx, const Fe &r) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(base + 4 * idx);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __hip_atomic_store(q + i, (unsigned long long)r.v[2 * i] | ((unsigned long long)r.v[2 * i + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ZK_D Fe fe_load_sc1(const uint64_t *base, uint64_t idx) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(base + 4 * idx);
    Fe r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned long long w = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.v[2 * i] = (uint32_t)w;
        r.v[2 * i + 1] = (uint32_t)(w >> 32);
    }
    return r;
}
ZK_D void dbg_stamp(uint64_t *dbg, int slot) {
    if (dbg && (threadIdx.x & 63) == 0) dbg[slot] = wall_clock64();
}

// ---- the transcript block ----------------------------------------------------------------------------------------------
// Reduce the block partials: value idx (< n_in <= 16) summed over the blocks -> red[idx].  256 threads: thread (idx =
// tid % 16, slice = tid / 16) adds its share, lanes 16/32 apart combine on the VALU, the four waves through LDS.
template <bool SC1 = false>
ZK_D void pipe_reduce_partials(const uint64_t *__restrict__ partials, uint32_t nblocks, uint32_t n_in, Fe *red /* LDS, 16 */, Fe (*stage)[16] /* LDS [4][16] */,
                               const FieldParams &P) {
    const uint32_t tid = threadIdx.x, idx = tid & 15, slice = tid >> 4, lane = tid & 63, wave = tid >> 6;
    Fe s = fe_zero();
    if (idx < n_in) {
        constexpr uint32_t kStep = kPipeThreads / 16;
        for (uint32_t b = slice; b < nblocks; b += 16 * kStep) {   // sixteen independent loads in flight (the partials of other
            Fe x[16];                                               // XCDs come from the memory side: ~1 us per dependent batch)
#pragma unroll
            for (int u = 0; u < 16; ++u)
                x[u] = b + u * kStep < nblocks ? (SC1 ? fe_load_sc1(partials, (uint64_t)(b + u * kStep) * n_in + idx) : fe_load(partials, (uint64_t)(b + u * kStep) * n_in + idx))
                                               : fe_zero();
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) x[u] = fe_add(x[u], x[u + w], P);
            s = fe_add(s, x[0], P);
        }
    }
    s = hex_rows_sum(s, P);
    if (lane < 16) stage[wave][lane] = s;
    __syncthreads();
    if (tid < 16) red[tid] = fe_add(fe_add(stage[0][tid], stage[1][tid], P), fe_add(stage[2][tid], stage[3][tid], P), P);
    __syncthreads();
}

template <int K, int D>
ZK_D void pipe_tail_block(const PipeTailArgs &ta, const FieldParams &P) {
    constexpr int NS = D + 1;
    __shared__ Fe red[16];
    __shared__ Fe stage[4][16];
    __shared__ Fe fin[4];
    __shared__ PipeEvalLds W;
    __shared__ int fin_canonical;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = {0, 0};
    if (wave0) dbg_stamp(ta.dbg, 0);
    if (wave0) sp = lane_sponge_load(ta.sponge, L);   // in flight while the partials are reduced
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (ta.nblocks == 1) {   // already reduced (the last work block of the launch before did it)
        if (wave >= 2) return;
        if (wave0 && lane < 16 && lane < ta.n_in) red[lane] = fe_load(ta.partials, lane);
    } else {
        pipe_reduce_partials(ta.partials, ta.nblocks, ta.n_in, red, stage, P);
        if (wave >= 2) return;
    }
    // wave 0: the transcript step.  wave 1: waits for the digest and converts it to the Montgomery form the proof records,
    // while wave 0 goes on with the multiplier form the next launch needs (one carry-free multiplication each)
    __shared__ Fe digest_x;
    Fe x = fe_zero();
    if (wave0) {
        __builtin_amdgcn_s_setprio(3);
        dbg_stamp(ta.dbg, 1);
        Fe s = fe_zero();
        if (ta.mode == 1) {
            const Mul29 rh29 = load_challenge29(ta.chal_in), rc29 = load_challenge29c(ta.chal_in);
            pipe_eval_consts(W, ta.pc, lane);
            if (lane == 0) pipe_eval_set(W, 0, K == 1 ? rc29 : rh29);
            __builtin_amdgcn_wave_barrier();
            s = pipe_eval_canon<K, D>(red, W, rh29, rc29, lane, P);   // CANONICAL
        } else {
            s = red[lane < (uint32_t)NS ? lane : 0];
            if (ta.dv.prev_rp && ta.dv.claim) {
                // S(1) = S_prev(r_prev) - S(0): the round kernel left the t = 1 products out (k_round_kd SKIP1) and its claim workgroup
                // has evaluated S_prev(r_prev)
                const Fe c0 = fe_load(ta.dv.claim, 0);
                const Fe s0 = red[0];
                if (lane == 1) s = fe_sub(c0, s0, P);
            } else if (ta.dv.prev_rp) {
                const Fe r = fe_load(ta.dv.prev_chal, 0);
                Fe term = fe_zero();
                if (lane < (uint32_t)NS) {
                    const Fe wt = fe_load(ta.dv.w, lane);
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
            if (ta.dv.lead) {   // slot D holds the leading coefficient (k_round_kd LEAD): lane D rebuilds S(D) from the others
                if (lane < (uint32_t)NS) fin[lane] = s;
                __builtin_amdgcn_wave_barrier();
                if (lane == ta.dv.lead) s = lead_rebuild(ta.dv.lead, fin, P);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane < (uint32_t)NS) {
            fin[lane] = s;
            if (ta.mode != 1) fe_store(ta.out_rp, lane, s);   // (mode 1: canonical here; wave 1 stores the Montgomery form)
        }
        if (lane == 0) fin_canonical = ta.mode == 1;
        dbg_stamp(ta.dbg, 2);
        if (ta.mode == 1) lane_absorb_elems<true>(sp, L, fin, NS, P);
        else lane_absorb_elems(sp, L, fin, NS, P);
        x = lane_squeeze_x(sp, L);
        if (lane == 0) digest_x = x;
        dbg_stamp(ta.dbg, 3);
    }
    __syncthreads();   // waves 0 and 1 (the others have left)
    if (wave0) {
        publish_challenge29_both(ta.chal_out, challenge29_both(x, ta.pc.k266, lane, P), lane);   // lane 0 / lane 16: the two forms
        lane_sponge_store(ta.sponge, sp, L);
        dbg_stamp(ta.dbg, 4);
    } else {
        // wave 1, one multiplication by R^2 for all of it: lanes t < NS the round polynomial (canonical -> Montgomery, mode 1),
        // lane 16 the challenge the proof records
        const bool rp = fin_canonical && lane < (uint32_t)NS;
        const Fe v = fe_from_canonical29(rp ? fin[lane] : digest_x, P);
        if (rp) fe_store(ta.out_rp, lane, v);
        if (lane == 16) {
            fe_store(ta.chal_out, 0, v);
            fe_store(ta.out_ch, 0, v);
        }
    }
}

// ---- the pipelined round kernel ------------------------------------------------------------------------------------------
// q = pairs of round s.  Transcript block 0, work blocks 1 .. gridDim.x-1 (16 pair indices per block and pass).
//   FOLD : fp.in = tables of round s-2 (8q elements), folded at *chal_fold (= r_{s-2}) into fp.out (tables of round s-1, 4q
//          elements, may alias fp.in: a lane reads and writes only positions of its own pair index);
//   !FOLD: fp.in = tables of round s-1 (4q elements), already materialised.
//   emit : compute the E_s partials (0: fold only -- the launch that leaves the pipeline).
// e_partials: slot 0 = the total over the work blocks (written by the block that finishes last), slot 1 + b = block b;
// each slot [t * (K+1) + node].  done_counter: zero on entry, zero again on exit.
template <int K, int D, int EXTRA>
constexpr int pipe_block_threads() {
    return kPipeThreads;   // one wave per SIMD: a work wave is a latency-bound chain, four of them on a SIMD run 4x slower each
}
template <int K, int D, int EXTRA, bool FOLD>
ZK_D void round_pipe_body(const FactorPtrs &fp, uint64_t q, int emit, const FieldParams &P, const uint64_t *__restrict__ chal_fold,
                          uint64_t *__restrict__ e_partials, uint32_t *done_counter, const PipeTailArgs &ta, int sc1_handoff) {
    using S = PipeShape<K, D, EXTRA>;
    using C = HexCfg<K, D, EXTRA, false>;
    constexpr int kThreads = pipe_block_threads<K, D, EXTRA>(), kPipeRows = kThreads / 16, kWaves = kThreads / 64;
    if (blockIdx.x == 0) {   // dispatched first: the serial transcript step is the launch's critical path
        if (threadIdx.x < (uint32_t)kPipeThreads) pipe_tail_block<K, D>(ta, P);   // (its barriers: the other waves have left)
        return;
    }
    __shared__ __attribute__((aligned(16))) Fe rows[kPipeRows][C::ROW_FE];
    __shared__ Fe redw[kWaves][16];
    const uint32_t wb = blockIdx.x - 1, nwork = gridDim.x - 1;
    Mul29 r = {};
    if (FOLD) r = load_challenge29(chal_fold);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lane16 = threadIdx.x & 15, rowi = threadIdx.x >> 4;
    const uint32_t fa = lane16 >> 2;   // phase-A factor of this lane
    const uint64_t *src = fp.in[0];
    uint64_t *dst = fp.out[0];
#pragma unroll
    for (int f = 1; f < S::NF; ++f) {   // pointer select by arithmetic masks (see hex_point)
        const uint64_t m = fa == (uint32_t)f ? ~0ull : 0ull;
        src = reinterpret_cast<const uint64_t *>((reinterpret_cast<uint64_t>(src) & ~m) | (reinterpret_cast<uint64_t>(fp.in[f]) & m));
        dst = reinterpret_cast<uint64_t *>((reinterpret_cast<uint64_t>(dst) & ~m) | (reinterpret_cast<uint64_t>(fp.out[f]) & m));
    }
    uint64_t *wdbg = (wb == 0 && ta.dbg) ? ta.dbg + 8 : nullptr;
    if (threadIdx.x < 64) dbg_stamp(wdbg, 0);
    HexAcc<false> A;
    hex_acc_zero(A);
    const uint64_t stride = (uint64_t)nwork * kPipeRows;
    for (uint64_t j0 = (uint64_t)wb * kPipeRows; j0 < q; j0 += stride) {   // j0 is block-uniform
        const uint64_t j = j0 + rowi;
        if (emit) hex_step<K, D, EXTRA, FOLD, false, true>(A, src, dst, j, q, j < q, r, P, rows[rowi], lane16);
        else hex_step<K, D, EXTRA, FOLD, false, false>(A, src, dst, j, q, j < q, r, P, rows[rowi], lane16);
    }
    if (threadIdx.x < 64) dbg_stamp(wdbg, 1);
    if (!emit) return;
    Fe e, plain;
    hex_close<K, false>(A, e, plain, P);
    e = hex_rows_sum(e, P);
    if (lane < 16) redw[wave][lane] = e;
    if (threadIdx.x < 64) dbg_stamp(wdbg, 2);
    __syncthreads();
    if (threadIdx.x < 16) {
        const uint32_t t = threadIdx.x >> 2, node = threadIdx.x & 3;
        if (t < (uint32_t)S::NS && node < (uint32_t)S::NR) {
            Fe tot = redw[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) tot = fe_add(tot, redw[w][threadIdx.x], P);
            if (sc1_handoff) fe_store_sc1(e_partials, (uint64_t)(wb + 1) * S::NE + t * S::NR + node, tot);
            else fe_store(e_partials, (uint64_t)(wb + 1) * S::NE + t * S::NR + node, tot);   // slot 0 is the total
        }
    }
    if (threadIdx.x < 64) dbg_stamp(wdbg, 3);
    // The block that finishes last adds the partials up (slot 0), so the NEXT launch's transcript block -- the critical path --
    // reads NE values instead of reducing nwork * NE.  Release / acquire at agent scope: the other blocks may sit on other XCDs.
    // EXPERIMENTAL (A/B builds with -DZK_PIPE_SC1_HANDOFF=1 only; the shipped library always passes 0): the same hand-off without the two fences -- write-through (sc1)
    // partial stores, the storing wave's s_waitcnt vmcnt(0), ONE lane's agent-scope counter add, and sc1 loads in the block whose
    // add came last (MI355X_MICROARCH.md "Valid forms": measured valid on gfx950 / ROCm 7.2, not an architectural guarantee).
    __shared__ uint32_t is_last;
    __shared__ Fe lred[16];
    __shared__ Fe lstage[4][16];
    if (sc1_handoff) {
        if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing lanes (tid < 16) are all in wave 0
        if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwork - 1 ? 1u : 0u;
        __syncthreads();
        if (!is_last) return;
        pipe_reduce_partials<true>(e_partials + S::NE * 4, nwork, S::NE, lred, lstage, P);
    } else {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            is_last = atomicAdd(done_counter, 1u) == nwork - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (!is_last) return;
        __threadfence();
        pipe_reduce_partials(e_partials + S::NE * 4, nwork, S::NE, lred, lstage, P);
    }
    if (threadIdx.x < 16) {
        const uint32_t t = threadIdx.x / S::NR, node = threadIdx.x % S::NR;
        if (threadIdx.x < (uint32_t)S::NE) fe_store(e_partials, t * S::NR + node, lred[threadIdx.x]);
    }
    if (threadIdx.x == 0) *done_counter = 0;   // ready for the next launch that uses this buffer
    if (threadIdx.x < 64) dbg_stamp(wdbg, 4);
}
template <int K, int D, int EXTRA, bool FOLD>
__global__ __launch_bounds__((pipe_block_threads<K, D, EXTRA>())) void k_round_pipe(FactorPtrs fp, uint64_t q, int emit, FieldParams P,
                                                                                  const uint64_t *__restrict__ chal_fold,
                                                                                  uint64_t *__restrict__ e_partials, uint32_t *done_counter,
                                                                                  PipeTailArgs ta, int sc1_handoff) {
    round_pipe_body<K, D, EXTRA, FOLD>(fp, q, emit, P, chal_fold, e_partials, done_counter, ta, sc1_handoff);
}
// batched form (zk_sumcheck_prove_batch): grid (1 + work blocks, proofs) -- B transcript blocks and B sets of work blocks in one launch
struct PipeSlot {
    FactorPtrs4 fp;
    const uint64_t *chal_fold;
    uint64_t *e_partials;
    uint32_t *done_counter;
    PipeTailArgs ta;
};
template <int K, int D, int EXTRA, bool FOLD>
__global__ __launch_bounds__((pipe_block_threads<K, D, EXTRA>())) void k_round_pipe_b(BatchOf<PipeSlot> b, uint64_t q, int emit, FieldParams P,
                                                                                    int sc1_handoff) {
    const PipeSlot &a = b.a[blockIdx.y];
    const FactorPtrs fp = factor_ptrs_of(a.fp);
    round_pipe_body<K, D, EXTRA, FOLD>(fp, q, emit, P, a.chal_fold, a.e_partials, a.done_counter, a.ta, sc1_handoff);
}

// ZK_SHARD_FAKE_ALLREDUCE_US: a stand-in for the latency of a multi-rank all-reduce on a one-rank communicator (spins on the 100 MHz clock)
__global__ void k_spin_us(uint32_t us) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)us * 100) {
    }
}

// ---- the pipelined finisher: all remaining rounds in ONE 1024-thread workgroup --------------------------------------------
// Wave 0 runs the transcript steps; waves 1..15 (60 rows) fold the tables in LDS and prepare the next round's E while it
// does; one barrier per round.  Entry (the tables in fp.in have m_in >= 3 variables):
//   kFinEntryFresh  : fp.in = tables of round s, nothing pending: a first, unpipelined round computes their plain sums
//   kFinEntryPending: fp.in = tables of round s-1 with r_{s-1} (*chal_in) pending: the same after folding them into LDS
//   kFinEntryPipe   : fp.in = tables of round s-1, r_{s-1} pending, E_s partials ready (from k_round_pipe): fully pipelined
// Rounds are numbered from 0 at the first round this kernel closes (out_rp / out_ch point at that round's slots).
enum { kFinEntryFresh = 0, kFinEntryPending = 1, kFinEntryPipe = 2 };
constexpr int kFinWorkWaves = kFinishPipeThreads / 64 - 1;
constexpr int kFinStageBytes = 24 * 1024;   // exchange buffers of the work rows
// The finisher takes over when the tables it folds have at most 2^8 elements (2^7 kept in LDS): its first prepared round then
// has 2^5 pair indices = three passes of its twelve work rows, about one transcript step; bigger rounds are faster as
// k_round_pipe launches spread over many CUs.
constexpr int finish_pipe_vars(int /*n_factors*/) { return 7; }

struct FinLds {   // carved from the dynamic region (32-byte aligned offsets)
    uint64_t *tab[4];
    Fe *stage;                       // kFinStageBytes
    Fe (*ew)[kFinWorkWaves][16];     // [parity][wave][4t + node]
    Fe (*pw)[16];                    // [wave][4t + node] plain sums (entry rounds)
    Fe (*st16)[16];                  // [16 waves][16]
    Fe *red;                         // [16]
    Fe *fin;                         // [4]
    Fe *xr;                          // [2] by round parity: the raw digest of the round's squeeze (for the Montgomery conversion)
    Fe (*finc)[4];                   // [2] by round parity: the round polynomial in CANONICAL form (converted for the proof a round later)
    PipeEvalLds *W;
    Mul29 *r29;                      // [2] by round parity: prepared Montgomery form of the challenge
    Mul29 *rc29;                     // [2] by round parity: prepared canonical form
    ZK_D uint32_t *red_u32() const { return reinterpret_cast<uint32_t *>(red); }
};
constexpr size_t kFinMiscBytes = kFinStageBytes + sizeof(Fe) * (2 * kFinWorkWaves * 16 + kFinWorkWaves * 16 + 16 * 16 + 16 + 4 + 2 + 8) +
                                 sizeof(PipeEvalLds) + 16 + 4 * 48 + 64;

// One pass set of the work rows over the pair indices j < q of the NEXT round.  src: tables with 8q (FOLD: folded at r into
// the 4q-element tables dst first) or 4q elements per factor.
template <int K, int D, int EXTRA, bool FOLD, bool PLAIN>
ZK_D uint32_t fin_rows(uint32_t n_work_waves) {
    using C = HexCfg<K, D, EXTRA, PLAIN>;
    constexpr uint32_t kBuf = kFinStageBytes / C::ROW_BYTES;   // rows with an exchange buffer
    return n_work_waves * 4 < kBuf ? n_work_waves * 4 : kBuf;
}
template <int K, int D, int EXTRA, bool FOLD, bool PLAIN>
ZK_D void fin_work(const uint64_t *src, uint64_t *dst, uint32_t q, const Mul29 &r, const FieldParams &P, Fe *stage, Fe (*ew)[16], Fe (*pw)[16],
                   uint32_t wl /* work lane */, uint32_t n_work_waves) {
    using C = HexCfg<K, D, EXTRA, PLAIN>;
    const uint32_t kRows = fin_rows<K, D, EXTRA, FOLD, PLAIN>(n_work_waves);
    const uint32_t lane = wl & 63, wwave = wl >> 6, lane16 = wl & 15, rowi = wl >> 4;
    if (__builtin_amdgcn_readfirstlane(wwave * 4) >= (q < kRows ? q : kRows)) return;   // no row of this wave ever works: its slots are not read
    const bool has_buf = rowi < kRows;
    Fe *row = stage + (size_t)(has_buf ? rowi : 0) * C::ROW_FE;
    HexAcc<PLAIN> A;
    hex_acc_zero(A);
    for (uint32_t j0 = 0; j0 < q; j0 += kRows) {
        const uint32_t j = j0 + rowi;
        if (has_buf) hex_step<K, D, EXTRA, FOLD, PLAIN, true>(A, src, dst, j, q, j < q, r, P, row, lane16);
    }
    Fe e, plain;
    hex_close<K, PLAIN>(A, e, plain, P);
    e = hex_rows_sum(e, P);
    if (lane < 16) ew[wwave][lane] = e;
    if constexpr (PLAIN) {
        plain = hex_rows_sum(plain, P);
        if (lane < 16) pw[wwave][lane] = plain;
    }
}
template <int K, int D, int EXTRA, bool PLAIN>
ZK_D uint32_t fin_active_waves(uint32_t q, uint32_t n_work_waves) {
    const uint32_t kRows = fin_rows<K, D, EXTRA, true, PLAIN>(n_work_waves);
    const uint32_t rows = q < kRows ? q : kRows;
    return (rows + 3) / 4;
}
// wave 0: total over the active work waves of ew[.][4t + node] -> e[node] on lane t (< NS); through LDS `red`
template <int K, int D>
ZK_D void fin_gather_e(Fe (*ew)[16], uint32_t nactive, Fe *red, Fe (&e)[K + 1], uint32_t lane, const FieldParams &P) {
    constexpr int NS = D + 1, NR = K + 1;
    const uint32_t idx = lane & 15, g = lane >> 4;
    Fe s = fe_zero();
    for (uint32_t w = g; w < nactive; w += 4) s = fe_add(s, ew[w][idx], P);
    s = hex_rows_sum(s, P);
    if (lane < 16) red[lane] = s;
    const uint32_t t = lane < (uint32_t)NS ? lane : 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) e[i] = red[4 * t + i];
}

template <int K, int D, int EXTRA>
ZK_D void finish_pipe_body(const FactorPtrs &fp, uint32_t m_in, int entry, const uint64_t *__restrict__ e_partials, uint32_t e_blocks, const FieldParams &P,
                           const PipeConsts &pc, const uint64_t *__restrict__ chal_in, uint64_t *__restrict__ chal_last, WordSponge *gsponge,
                           uint64_t *out_rp, uint64_t *out_ch, uint64_t *out_final, uint64_t *dbg, const FinishPublish &pub) {
    constexpr int NF = K + EXTRA, NS = D + 1, NR = K + 1, NE = NS * NR;
    extern __shared__ __attribute__((aligned(32))) unsigned char fp_smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;
    const uint32_t cap = 1u << (m_in - 1);   // elements per factor ever held in LDS
    FinLds S;
    unsigned char *carve = fp_smem;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        S.tab[f] = reinterpret_cast<uint64_t *>(carve);
        carve += (size_t)cap * 32;
    }
    S.stage = reinterpret_cast<Fe *>(carve);
    carve += kFinStageBytes;
    S.ew = reinterpret_cast<Fe(*)[kFinWorkWaves][16]>(carve);
    carve += sizeof(Fe) * 2 * kFinWorkWaves * 16;
    S.pw = reinterpret_cast<Fe(*)[16]>(carve);
    carve += sizeof(Fe) * kFinWorkWaves * 16;
    S.st16 = reinterpret_cast<Fe(*)[16]>(carve);
    carve += sizeof(Fe) * 16 * 16;
    S.red = reinterpret_cast<Fe *>(carve);
    carve += sizeof(Fe) * 16;
    S.fin = reinterpret_cast<Fe *>(carve);
    carve += sizeof(Fe) * 4;
    S.xr = reinterpret_cast<Fe *>(carve);
    carve += sizeof(Fe) * 2;
    S.finc = reinterpret_cast<Fe(*)[4]>(carve);
    carve += sizeof(Fe) * 8;
    S.W = reinterpret_cast<PipeEvalLds *>(carve);          // (32-byte aligned so far: the multiplier records are read 16 bytes at a time)
    carve += (sizeof(PipeEvalLds) + 15) / 16 * 16;
    // two Mul29 records each (indexed by round parity: sizeof(Mul29) = 36 bytes apart, NOT 16-byte aligned -- they are read one
    // 32-bit limb at a time); 96 bytes are reserved per pair, 4 * 48 in kFinMiscBytes
    static_assert(2 * sizeof(Mul29) <= 2 * 48, "the r29 / rc29 pairs overlap");
    S.r29 = reinterpret_cast<Mul29 *>(carve);
    S.rc29 = reinterpret_cast<Mul29 *>(carve + 2 * 48);
    if (wave0) pipe_eval_consts(*S.W, pc, lane);

    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = {0, 0};
    if (wave0) sp = lane_sponge_load(gsponge, L);
    // The transcript wave must have its SIMD to itself: a wave issues at most one VALU instruction per 4 cycles, and four
    // busy waves on one SIMD get a quarter of that each (measured: the transcript step took 6.8 us next to three work waves,
    // 3.7 us alone).  Waves that landed on wave 0's SIMD (HW_ID.SIMD_ID) stay idle; the others take work-wave numbers.
    {
        const uint32_t simd = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));   // hwreg(HW_REG_HW_ID, 4, 2)
        if (lane == 0) S.red_u32()[wave] = simd;
    }
    __syncthreads();
    uint32_t my_work_wave = 0xffffffffu, n_work_waves = 0;   // dense numbering of the waves that work
    {
        const uint32_t simd0 = S.red_u32()[0];
        for (uint32_t w = 1; w < (uint32_t)kFinishPipeThreads / 64; ++w) {
            if (S.red_u32()[w] == simd0) continue;
            if (w == wave) my_work_wave = n_work_waves;
            ++n_work_waves;
        }
    }
    __syncthreads();
    const bool worker = my_work_wave != 0xffffffffu;
    const uint32_t wl = my_work_wave * 64 + lane;   // work lane
    // this lane's phase-A factor: its global source and its LDS table
    const uint32_t fa = (wl & 15) >> 2;
    const uint64_t *gsrc = fp.in[0];
    uint64_t *ltab = S.tab[0];
#pragma unroll
    for (int f = 1; f < NF; ++f) {
        const uint64_t msk = fa == (uint32_t)f ? ~0ull : 0ull;
        gsrc = reinterpret_cast<const uint64_t *>((reinterpret_cast<uint64_t>(gsrc) & ~msk) | (reinterpret_cast<uint64_t>(fp.in[f]) & msk));
        ltab = reinterpret_cast<uint64_t *>((reinterpret_cast<uint64_t>(ltab) & ~msk) | (reinterpret_cast<uint64_t>(S.tab[f]) & msk));
    }
    uint32_t m;            // variables of the table the NEXT work pass reads (the "previous" table of the uniform loop)
    bool src_global;       // ... and whether it is still in global memory
    uint32_t round = 0;    // rounds closed so far
    int par = 0;           // parity of the ew buffer that holds the E of round `round`
    uint32_t nact = 0;     // work waves that wrote it
    Mul29 rprev = {}, rcprev = {};   // prepared Montgomery / canonical forms of the challenge of the round before
    if (entry != kFinEntryFresh) rprev = load_challenge29(chal_in);
    if (entry == kFinEntryPipe) rcprev = load_challenge29c(chal_in);

    // ---- entry ----
    if (entry == kFinEntryPipe) {
        // E of round 0 (of this kernel) comes from the pipelined launch before: reduce its block partials with everybody
        {
            const uint32_t idx = tid & 15, slice = tid >> 4;
            const uint32_t t = idx >> 2, node = idx & 3;
            Fe s = fe_zero();
            if (t < (uint32_t)NS && node < (uint32_t)NR)
                for (uint32_t b = slice; b < e_blocks; b += kFinishPipeThreads / 16) s = fe_add(s, fe_load(e_partials, (uint64_t)b * NE + t * NR + node), P);
            s = hex_rows_sum(s, P);
            if (lane < 16) S.st16[wave][lane] = s;
        }
        __syncthreads();
        if (tid < 16) {
            Fe tot = fe_zero();
            for (int w = 0; w < kFinishPipeThreads / 64; ++w) tot = fe_add(tot, S.st16[w][tid], P);
            S.ew[0][0][tid] = tot;   // as if ONE work wave had produced it
        }
        __syncthreads();
        par = 0;
        nact = 1;
        m = m_in;
        src_global = true;
    } else {
        // unpipelined first round: plain sums of the round's own table (+ E of the next round)
        const uint32_t q = 1u << (m_in - (entry == kFinEntryPending ? 3 : 2));
        if (worker) {
            if (entry == kFinEntryPending) fin_work<K, D, EXTRA, true, true>(gsrc, ltab, q, rprev, P, S.stage, S.ew[1], S.pw, wl, n_work_waves);
            else fin_work<K, D, EXTRA, false, true>(gsrc, ltab, q, rprev, P, S.stage, S.ew[1], S.pw, wl, n_work_waves);
        }
        __syncthreads();
        const uint32_t na = fin_active_waves<K, D, EXTRA, true>(q, n_work_waves);
        if (wave0) {
            // plain S(t) = (node 0 part) + (node 1 part), summed over the active waves
            Fe s = fe_zero();
            if (lane < (uint32_t)NS)
                for (uint32_t w = 0; w < na; ++w) s = fe_add(s, fe_add(S.pw[w][4 * lane], S.pw[w][4 * lane + 1], P), P);
            if (lane < (uint32_t)NS) {
                S.fin[lane] = s;
                fe_store(out_rp, lane, s);
            }
            lane_absorb_elems(sp, L, S.fin, NS, P);
            const Fe x = lane_squeeze_x(sp, L);
            if (lane == 0) S.xr[0] = x;
            if (!n_work_waves) publish_challenge_fe(nullptr, out_ch, challenge_fe_of(x, P), (int)lane);   // (else: the converter wave, below)
            const Mul29 both = challenge29_both(x, pc.k266, lane, P);
            if (lane == 0) S.r29[0] = both;
            if (lane == 16) S.rc29[0] = both;
        }
        __syncthreads();
        round = 1;
        par = 1;
        nact = na;
        if (entry == kFinEntryPending) {
            m = m_in - 1;
            src_global = false;
        } else {
            m = m_in;
            src_global = true;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            rprev.l[i] = __builtin_amdgcn_readfirstlane(S.r29[0].l[i]);
            rcprev.l[i] = __builtin_amdgcn_readfirstlane(S.rc29[0].l[i]);
        }
    }

    // ---- uniform loop: close round `round` (its table has m - 1 variables) while the next one is prepared ----
    // rprev = challenge of round - 1 (folds the m-variable table into this round's)
    if (wave0) __builtin_amdgcn_s_setprio(3);   // the transcript wave is the critical path: first pick of issue slots / LDS
    // the wave that turns digests into Montgomery-form challenges: the LAST work wave (the rows of a small round fill the work
    // waves from the first one up, so it is the one most often idle); with no work wave at all, the transcript wave itself
    const bool converter = n_work_waves ? (worker && my_work_wave == n_work_waves - 1) : wave0;
    // a round has squeezed a digest (and, in the loop, left a canonical round polynomial) that the converter has not turned into
    // the proof's Montgomery form yet; conv_rp: the round polynomial too (the entry round stored its own)
    bool conv_pending = entry != kFinEntryPipe && n_work_waves != 0, conv_rp = false;
    // converter: one multiplication by R^2 for the challenge (lane 16) and the round polynomial (lanes t < NS) of round `rd`
    auto convert_round = [&](uint32_t rd, bool with_rp, uint64_t *also_chal) {
        const bool rp = with_rp && lane < (uint32_t)NS;
        const Fe v = fe_from_canonical29(rp ? S.finc[rd & 1][lane < (uint32_t)NS ? lane : 0] : S.xr[rd & 1], P);
        if (rp) fe_store(out_rp, (uint64_t)rd * NS + lane, v);
        if (lane == 16) {
            fe_store(out_ch, rd, v);
            if (also_chal) fe_store(also_chal, 0, v);
        }
    };
    while (m >= 2) {
        uint64_t *rdbg = dbg ? dbg + 16 * round : nullptr;
        if (wave0) {
            dbg_stamp(rdbg, 0);
            Fe e[NR];
            fin_gather_e<K, D>(S.ew[par], nact, S.red, e, lane, P);   // (leaves the totals in S.red, [4 t + node])
            dbg_stamp(rdbg, 1);
            if (lane == 0) pipe_eval_set(*S.W, 0, K == 1 ? rcprev : rprev);
            __builtin_amdgcn_wave_barrier();
            const Fe s = pipe_eval_canon<K, D, 4>(S.red, *S.W, rprev, rcprev, lane, P);   // CANONICAL
            dbg_stamp(rdbg, 2);
            if (lane < (uint32_t)NS) S.finc[round & 1][lane] = s;
            // the transcript wave keeps only what the next round needs: the two multiplier forms of the challenge.  The Montgomery
            // forms the proof records (round polynomial, challenge) are converted by the last work wave, one round later (below)
            lane_absorb_elems<true>(sp, L, S.finc[round & 1], NS, P);
            const Fe x = lane_squeeze_x(sp, L);
            if (lane == 0) S.xr[round & 1] = x;
            if (converter) convert_round(round, true, nullptr);   // no work wave exists
            const Mul29 both = challenge29_both(x, pc.k266, lane, P);
            if (lane == 0) S.r29[round & 1] = both;
            if (lane == 16) S.rc29[round & 1] = both;
            if (m == 2) publish_challenge29_both(chal_last, both, lane);
            dbg_stamp(rdbg, 3);
        } else if (worker) {
            if (converter && conv_pending) convert_round(round - 1, conv_rp, nullptr);
            if (m >= 3) {
            if (my_work_wave == 0) dbg_stamp(rdbg, 8);
            const uint32_t q = 1u << (m - 3);
            if (src_global) fin_work<K, D, EXTRA, true, false>(gsrc, ltab, q, rprev, P, S.stage, S.ew[par ^ 1], S.pw, wl, n_work_waves);
            else fin_work<K, D, EXTRA, true, false>(ltab, ltab, q, rprev, P, S.stage, S.ew[par ^ 1], S.pw, wl, n_work_waves);
            if (my_work_wave == 0) dbg_stamp(rdbg, 9);
            }
        }
        conv_pending = true;
        conv_rp = true;
        __syncthreads();
        if (wave0) dbg_stamp(rdbg, 4);
        if (m >= 3) nact = fin_active_waves<K, D, EXTRA, false>(1u << (m - 3), n_work_waves);
        if (m > 2 || out_final) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                rprev.l[i] = __builtin_amdgcn_readfirstlane(S.r29[round & 1].l[i]);
                rcprev.l[i] = __builtin_amdgcn_readfirstlane(S.rc29[round & 1].l[i]);
            }
        }
        par ^= 1;
        src_global = false;
        ++round;
        --m;
    }
    if (wave0) lane_sponge_store(gsponge, sp, L);
    // the last round's challenge and round polynomial (the barrier that ended the loop has published them)
    if (converter && conv_pending && n_work_waves) convert_round(round - 1, conv_rp, chal_last);
    else if (converter && conv_pending && lane == 16) fe_store(chal_last, 0, fe_load(out_ch, round - 1));   // (converted in the loop already)
    // out_final: the factors at the whole challenge point.  LDS holds the 4-element tables of the second-to-last round; they
    // are folded at the last two challenges (rprev = the last one; the one before sits in the other slot)
    if (out_final && tid < (uint32_t)NF) {
        Mul29 r2;
#pragma unroll
        for (int i = 0; i < 9; ++i) r2.l[i] = S.r29[round & 1].l[i];   // round - 2 has the parity of round
        const uint64_t *T = S.tab[0];
#pragma unroll
        for (int f = 1; f < NF; ++f)
            if (tid == (uint32_t)f) T = S.tab[f];
        const Fe x0 = fe_load(T, 0), x1 = fe_load(T, 1), x2 = fe_load(T, 2), x3 = fe_load(T, 3);
        const Fe lo = fe_sub(x0, fe_mul29(fe_sub(x0, x2, P), r2, P), P), hi = fe_sub(x1, fe_mul29(fe_sub(x1, x3, P), r2, P), P);
        fe_store(out_final, tid, fe_sub(lo, fe_mul29(fe_sub(lo, hi, P), rprev, P), P));
    }
    if (pub.flag) {   // this launch ends the call: the proof block goes to pinned host memory from here (pipe_args.hpp FinishPublish)
        __syncthreads();   // the round polynomials / challenges / factor values stored above by the other waves
        if (wave0) {
            for (uint32_t i = 2 * lane; i < pub.n_u64; i += 2 * 64)
                *reinterpret_cast<uint4 *>(pub.dst_host + i) = *reinterpret_cast<const uint4 *>(pub.src + i);
            __threadfence_system();
            if (lane == 0) *pub.flag = pub.seq;
        }
    }
}
template <int K, int D, int EXTRA>
__global__ __launch_bounds__(kFinishPipeThreads) void k_finish_pipe(FactorPtrs fp, uint32_t m_in, int entry, const uint64_t *__restrict__ e_partials,
                                                                    uint32_t e_blocks, FieldParams P, PipeConsts pc, const uint64_t *__restrict__ chal_in,
                                                                    uint64_t *__restrict__ chal_last, WordSponge *gsponge, uint64_t *out_rp,
                                                                    uint64_t *out_ch, uint64_t *out_final, uint64_t *dbg, FinishPublish pub) {
    finish_pipe_body<K, D, EXTRA>(fp, m_in, entry, e_partials, e_blocks, P, pc, chal_in, chal_last, gsponge, out_rp, out_ch, out_final, dbg, pub);
}
// batched form: grid (1, proofs), one finisher workgroup per proof (each on its own CU: the dynamic LDS is per workgroup)
struct FinishSlot {
    FactorPtrs4 fp;
    const uint64_t *e_partials, *chal_in;
    uint64_t *chal_last;
    WordSponge *sponge;
    uint64_t *out_rp, *out_ch, *out_final;
    FinishPublish pub;
};
template <int K, int D, int EXTRA>
__global__ __launch_bounds__(kFinishPipeThreads) void k_finish_pipe_b(BatchOf<FinishSlot> b, uint32_t m_in, int entry, uint32_t e_blocks, FieldParams P,
                                                                      PipeConsts pc) {
    const FinishSlot &a = b.a[blockIdx.y];
    const FactorPtrs fp = factor_ptrs_of(a.fp);
    finish_pipe_body<K, D, EXTRA>(fp, m_in, entry, a.e_partials, e_blocks, P, pc, a.chal_in, a.chal_last, a.sponge, a.out_rp, a.out_ch, a.out_final, nullptr, a.pub);
}

}  // namespace zk
