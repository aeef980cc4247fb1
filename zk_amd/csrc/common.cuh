// common.cuh -- constants and argument structs shared by the kernel translation units.
#pragma once
#include "field.cuh"

namespace zk {

constexpr int kBlock = 256;
constexpr int kMaxFactors = 8;
constexpr int kMaxLazy = 16;   // products accumulated unreduced between Montgomery reductions (see redc_wide)

struct FactorPtrs {
    const uint64_t *in[kMaxFactors];
    uint64_t *out[kMaxFactors];
};

// Challenge record in device memory (written by the on-device transcript): 8 words of r (Montgomery form) followed by
// the 9 words of its prepared multiplier form (Mul29 of r).  kChallengeBytes is the allocation size.
constexpr int kChallengeBytes = 96;
// A sum of products sum_i prod_{f in term i} T_f: the factors are listed flat (FactorPtrs), term after term.
constexpr int kMaxTerms = 4;
struct TermSpec {
    int n_terms;
    int term_k[kMaxTerms];
};
#if defined(__HIPCC__)
ZK_D Mul29 load_challenge29(const uint64_t *rptr) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(rptr) + 8;
    Mul29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = __builtin_amdgcn_readfirstlane(w[i]);   // wave-uniform -> SGPRs
    return r;
}
// Big fused rounds leave out the t = 1 sums; the tail derives S_i(1) = S_{i-1}(r_{i-1}) - S_i(0) (k_round_kd, SKIP1).
// prev_rp: the previous round polynomial (D + 1 elements, device); w[t] = 1 / prod_{u != t} (t - u), the Lagrange weights
// on the nodes 0..D (Montgomery form, computed by the host once per proof).
constexpr int kMaxSkipDegree = 4;
struct TailDerive {
    const uint64_t *prev_rp;   // null: nothing to derive
    Fe w[kMaxSkipDegree + 1];
};

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
#endif

}  // namespace zk
