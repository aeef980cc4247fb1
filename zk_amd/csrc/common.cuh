// common.cuh -- constants and argument structs shared by the kernel translation units.
#pragma once
#include "field.cuh"

namespace zk {

constexpr int kBlock = 256;
constexpr int kMaxFactors = 8;

struct FactorPtrs {
    const uint64_t *in[kMaxFactors];
    uint64_t *out[kMaxFactors];
};

// ---- batched launches (zk_sumcheck_prove_batch): ONE launch per round for up to kMaxBatch independent proofs of the same shape and
// size.  blockIdx.y selects the proof; its pointers come from slot blockIdx.y of an argument array passed by value (kernarg), every
// other argument (sizes, field, grid) is shared.  The per-proof kernel bodies are the single-proof kernels' own bodies.
constexpr int kMaxBatch = 8;
template <class Slot>
struct BatchOf {
    Slot a[kMaxBatch];
};
struct FactorPtrs4 {   // the batched shapes have at most four tables per proof (keeps the slot arrays small)
    const uint64_t *in[4];
    uint64_t *out[4];
};
ZK_HD FactorPtrs4 factor_ptrs4(const FactorPtrs &f) {
    FactorPtrs4 r;
    for (int i = 0; i < 4; ++i) r.in[i] = f.in[i], r.out[i] = f.out[i];
    return r;
}
ZK_HD FactorPtrs factor_ptrs_of(const FactorPtrs4 &f) {
    FactorPtrs r = {};
    for (int i = 0; i < 4; ++i) r.in[i] = f.in[i], r.out[i] = f.out[i];
    return r;
}

// Challenge record in device memory (written by the on-device transcript): 8 words of r (Montgomery form), the 9 words of
// its prepared multiplier form (Mul29 of r: what folds multiply by), and the 9 words of the prepared form of the CANONICAL
// challenge (Mul29 of r * R^-1: a Montgomery-form value times it is a canonical product -- the pipelined rounds close with
// it, pipe_kernels.cuh pipe_eval_canon).  kChallengeBytes is the allocation size.
constexpr int kChallengeBytes = 128;
constexpr int kChalCanonWord = 17;   // 32-bit word offset of the canonical prepared form
// A sum of products sum_i prod_{f in term i} T_f: the factors are listed flat (FactorPtrs), term after term.
constexpr int kMaxTerms = 4;
struct TermSpec {
    int n_terms;
    int term_k[kMaxTerms];
};
#if defined(__HIPCC__)
ZK_D Mul29 load_challenge29c(const uint64_t *rptr) {   // the prepared canonical form (wave-uniform -> SGPRs)
    const uint32_t *w = reinterpret_cast<const uint32_t *>(rptr) + kChalCanonWord;
    Mul29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = __builtin_amdgcn_readfirstlane(w[i]);
    return r;
}
ZK_D Mul29 load_challenge29(const uint64_t *rptr) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(rptr) + 8;
    Mul29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = __builtin_amdgcn_readfirstlane(w[i]);   // wave-uniform -> SGPRs
    return r;
}
// ---- wave-coalesced element access (the fold's and the big rounds' data movement) ------------------------------------------
// A wave moves a run of 64 consecutive elements (2 KiB) as two fully coalesced 1-KiB dwordx4 accesses: lane l touches bytes
// [16l, 16l+16) of each half-run, so it holds half (l & 1) of element (l >> 1) of chunk A (elements 0..31) and of chunk B
// (32..63); lanes 2i / 2i+1 swap one half with a DPP quad_perm and each lane owns a whole element: even lane 2i element i,
// odd lane 2i+1 element 32+i.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
ZK_D uint4 nt_load16(const uint4 *p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
ZK_D void nt_store16(uint4 v, uint4 *p) {
    const u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(p));
}
ZK_D uint32_t swap_pair_lane(uint32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true); }
// chunks A (elements 0..31 of the run) and B (elements 32..63): lane l holds half (l & 1) of element (l >> 1) of each
ZK_D Fe pair_gather(const uint4 &A, const uint4 &B, bool odd) {
    const uint4 send = odd ? A : B;   // what the neighbour needs from me
    const uint4 recv = make_uint4(swap_pair_lane(send.x), swap_pair_lane(send.y), swap_pair_lane(send.z), swap_pair_lane(send.w));
    Fe r;
    if (!odd) r = {{A.x, A.y, A.z, A.w, recv.x, recv.y, recv.z, recv.w}};
    else r = {{recv.x, recv.y, recv.z, recv.w, B.x, B.y, B.z, B.w}};
    return r;
}
ZK_D void pair_scatter(const Fe &e, bool odd, uint4 &A, uint4 &B) {
    const uint4 lo = make_uint4(e.v[0], e.v[1], e.v[2], e.v[3]), hi = make_uint4(e.v[4], e.v[5], e.v[6], e.v[7]);
    const uint4 send = odd ? lo : hi;
    const uint4 recv = make_uint4(swap_pair_lane(send.x), swap_pair_lane(send.y), swap_pair_lane(send.z), swap_pair_lane(send.w));
    if (!odd) {
        A = lo;
        B = recv;
    } else {
        A = recv;
        B = hi;
    }
}
// the element a lane owns after pair_gather, relative to the start of the wave's run
ZK_D uint32_t pair_owned(uint32_t lane) { return (lane >> 1) + ((lane & 1) << 5); }
// plain (cached) forms for tables that the next round re-reads
ZK_D Fe run_load(const uint64_t *run /* first element of the wave's run */, uint32_t lane) {
    const uint4 *p = reinterpret_cast<const uint4 *>(run) + lane;
    return pair_gather(p[0], p[64], lane & 1);
}
ZK_D void run_store(uint64_t *run, uint32_t lane, const Fe &e) {
    uint4 A, B;
    pair_scatter(e, lane & 1, A, B);
    uint4 *p = reinterpret_cast<uint4 *>(run) + lane;
    p[0] = A;
    p[64] = B;
}

// nontemporal forms (tables streamed once)
ZK_D Fe run_load_nt(const uint64_t *run, uint32_t lane) {
    const uint4 *p = reinterpret_cast<const uint4 *>(run) + lane;
    return pair_gather(nt_load16(p), nt_load16(p + 64), lane & 1);
}
ZK_D void run_store_nt(uint64_t *run, uint32_t lane, const Fe &e) {
    uint4 A, B;
    pair_scatter(e, lane & 1, A, B);
    uint4 *p = reinterpret_cast<uint4 *>(run) + lane;
    nt_store16(A, p);
    nt_store16(B, p + 64);
}

// Big fused rounds leave out the t = 1 sums; the tail derives S_i(1) = S_{i-1}(r_{i-1}) - S_i(0) (k_round_kd, SKIP1).
// prev_rp: the previous round polynomial (D + 1 elements, device); w[t] = 1 / prod_{u != t} (t - u), the Lagrange weights
// on the nodes 0..D (Montgomery form, computed by the host once per context and degree, kept in device memory).
constexpr int kMaxSkipDegree = 4;
struct TailDerive {
    const uint64_t *prev_rp;   // null: nothing to derive
    const uint64_t *prev_chal; // challenge record of the previous round (r_{i-1})
    const uint64_t *w;         // the D + 1 weights, device memory (one buffer per context and degree)
    uint32_t lead;             // D > 0: slot D of the sums holds the leading coefficient L (k_round_kd LEAD); rebuild S(D)
    uint32_t local_only;       // sharded prover: these are ONE rank's sums -- k_round_tail derives nothing (slot 1 -> 0, slot D keeps L:
                               // both are linear in the shards); k_lanes_transcript derives from the all-reduced values
    const uint64_t *claim;     // one element of device memory: S_prev(r_prev), evaluated by the round kernel's claim workgroup (ClaimJob) --
                               // the tails read it instead of running the Lagrange chain themselves (null: they evaluate it)
    uint32_t log_world;        // k_lanes_transcript: the all-reduced value is below 2^log_world * p (how far its reduction ladder must reach)
};
// The claim S_prev(r_prev) a SKIP1 round needs (S(1) = claim - S(0)) depends on the round BEFORE only, so the round kernel itself
// evaluates it -- one extra workgroup, first wave, beside the work blocks -- and parks it in device memory (`out`); the tail that
// closes the round reads one element instead of running a chain of D + 1 dependent multiplications on the prover's serial path.
struct ClaimJob {
    const uint64_t *prev_rp;   // previous round polynomial (ns elements); null: no job
    const uint64_t *prev_chal; // challenge record of the previous round
    const uint64_t *w;         // Lagrange weights on 0..ns-1
    uint64_t *out;             // one element
    uint32_t ns;
};
// S(D) of a degree-D round polynomial from S(0..D-1) and its leading coefficient L (in S[D]): D <= 3
ZK_HD Fe lead_rebuild(uint32_t D, const Fe *S, const FieldParams &P) {
    const Fe L = S[D];
    if (D == 1) return fe_add(S[0], L, P);                                       // S1 = S0 + L
    if (D == 2) {                                                                // S2 = 2 (S1 + L) - S0
        const Fe a = fe_add(S[1], L, P);
        return fe_sub(fe_add(a, a, P), S[0], P);
    }
    Fe b = fe_add(fe_sub(S[2], S[1], P), fe_add(L, L, P), P);                    // S3 = S0 + 3 (S2 - S1 + 2 L)
    b = fe_add(fe_add(b, b, P), b, P);
    return fe_add(S[0], b, P);
}

// Sum of one field element per lane over a wave, entirely on the VALU: v_permlane32_swap / v_permlane16_swap (gfx950)
// across wave halves and 16-lane rows, DPP row rotations inside a row -- no ds_bpermute round trips.  Lanes >= width
// must hold zero (their levels are skipped; width is wave-uniform); the total is valid in lane 0, and in every lane when
// width == 64.  Addition in F_p is commutative and exact, so the order of the levels does not change the result.
template <int CTRL>
ZK_D Fe fe_dpp(const Fe &s) {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = __builtin_amdgcn_update_dpp(0u, s.v[i], CTRL, 0xF, 0xF, true);
    return o;
}
ZK_D Fe fe_wave_sum(Fe s, const FieldParams &P, uint32_t width = 64) {
    if (width > 32) {
        Fe a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto r = __builtin_amdgcn_permlane32_swap(s.v[i], s.v[i], false, false);
            a.v[i] = r[0];
            b.v[i] = r[1];
        }
        s = fe_add(a, b, P);
    }
    if (width > 16) {
        Fe a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const auto r = __builtin_amdgcn_permlane16_swap(s.v[i], s.v[i], false, false);
            a.v[i] = r[0];
            b.v[i] = r[1];
        }
        s = fe_add(a, b, P);
    }
    // row_ror:n makes lane i read lane (i - n) mod 16: rotate by 16 - k so that lane 0 reads lane k (matters when width < 16)
    if (width > 8) s = fe_add(s, fe_dpp<0x128>(s), P);
    if (width > 4) s = fe_add(s, fe_dpp<0x12C>(s), P);
    if (width > 2) s = fe_add(s, fe_dpp<0x12E>(s), P);
    if (width > 1) s = fe_add(s, fe_dpp<0x12F>(s), P);
    return s;
}
// one wave: claim = sum_t prev[t] * w[t] * prod_{u != t} (r - u); lane t takes term t (ns <= 8); valid in lane 0
ZK_D Fe claim_eval(const ClaimJob &cj, const FieldParams &P) {
    const uint32_t lane = threadIdx.x & 63;
    const Fe r = fe_load(cj.prev_chal, 0);
    Fe term = fe_zero();
    if (lane < cj.ns) {
        Fe one;
#pragma unroll
        for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
        term = fe_mul(fe_load(cj.prev_rp, lane), fe_load(cj.w, lane), P);
        Fe node = fe_zero();   // Montgomery form of u
        for (uint32_t u = 0; u < cj.ns; ++u) {
            if (u != lane) term = fe_mul(term, fe_sub(r, node, P), P);
            node = fe_add(node, one, P);
        }
    }
    return fe_wave_sum(term, P, 8);
}

// ---- the variables a bulk launch leaves over, applied as a weight on its outputs (round 4) ------------------------------------
// A bulk launch (k_eval_low, k_eval_stream) writes out[g] = the table restricted in its LOW L variables; when H <= 12 variables
// remain, workgroup g also multiplies its output by eq(point_high, g) -- H selected factors, multiplied up by ONE otherwise idle wave
// as a four-level butterfly over a 16-lane row while the other waves build their tables -- so that what is left of evaluate is the
// plain SUM of the 2^H outputs (k_eval_sum: ~3 us) instead of a second bulk launch with its own tables, products and reduction
// (10-12 us).  Exact field arithmetic: sum_g eq(hi, g) * out[g] is the same canonical element.
constexpr int kEvalHighMax = 12;
struct EvalHighPoint {   // r for bit p of the OUTPUT index g (Montgomery form); n = 0: the outputs are not weighted
    uint32_t n;
    uint32_t r[kEvalHighMax][8];
};
ZK_D uint32_t eval_high_lane(uint32_t n) { return n > 8 ? 15u : n > 4 ? 7u : n > 2 ? 3u : n > 1 ? 1u : 0u; }
ZK_D Fe eval_high_weight(const EvalHighPoint &ph, uint64_t g, uint32_t lane, const FieldParams &P) {
    Fe one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.v[i] = P.r1[i];
    Fe f = one;
    for (uint32_t k = 0; k < ph.n; ++k) {   // uniform k: scalar loads from the argument segment, no per-lane indexing of it
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = ph.r[k][i];
        const Fe sel = (g >> k) & 1 ? r : fe_sub(one, r, P);
        if ((lane & 15) == k) f = sel;
    }
    // row_ror 8, 4, 2, 1: every lane of the row ends with the product of all sixteen.  A level whose partner lanes all hold `one`
    // is skipped (ph.n is uniform): 8 factors -- every table of up to 20 variables -- take three dependent multiplications, not four
    if (ph.n > 8) f = fe_mul(f, fe_dpp<0x128>(f), P);
    if (ph.n > 4) f = fe_mul(f, fe_dpp<0x124>(f), P);
    if (ph.n > 2) f = fe_mul(f, fe_dpp<0x122>(f), P);
    if (ph.n > 1) f = fe_mul(f, fe_dpp<0x121>(f), P);
    return f;   // complete in lane eval_high_lane(ph.n) (row_ror:m makes lane i take lane i - m: the last lane of the factors' power-of-two block has them all)
}
#endif

}  // namespace zk
