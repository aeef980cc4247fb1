// rounds.hip -- instantiations + launch logic of the sumcheck round kernels.
#include <cstdlib>

#include "env.hpp"
#include "launch.hpp"
#include "round_kernels.cuh"

namespace zk {

static constexpr uint32_t kCapGrid = 2048;   // 8 workgroups per CU on 256 CUs

// Grid of a specialised round kernel.  Every thread pays a fixed epilogue (Montgomery-reduce the unreduced accumulators,
// then the workgroup's modular reduction tree: ~1600 instructions) on top of ~800-2200 instructions per pair index, so
// the cheapest grid is the one with the FEWEST waves that still fills the machine: up to kMaxLazy pairs per thread, but
// at least kMinBlocks workgroups (2 waves per SIMD on 256 CUs) while there is one pair per thread to give them.
static uint32_t min_blocks() {
    static const uint32_t v = (uint32_t)env_u64("ZK_ROUND_MIN_BLOCKS", 512, 1, 2048);   // tuning override
    return v;
}
// fused rounds with at most this many pairs (and at least 16) use the four-lanes-per-pair-index kernel (ZK_QUAD_MAX_PAIRS;
// 0 switches it off)
// ZK_ROUND0_DOT29: 1 (default) = round 0 of the two-table (2, 2) product on the carry-free kernel; 0 = the wide-accumulator kernel
// everywhere; 2 = the carry-free kernel for the product-plus-term shape as well (A/B)
static int round0_dot29_mode() {
    static const int v = (int)env_u64("ZK_ROUND0_DOT29", 1, 0, 2);
    return v;
}
static bool round0_dot29() { return round0_dot29_mode() != 0; }
static bool round0_dot29_extra() { return round0_dot29_mode() == 2; }
// ZK_ROUND_GLDS: 1 (default) = the big rounds take the LDS-DMA kernels (round_kernels.cuh: k_round0_glds<0 / 1> for round 0 over two
// tables (+ a term), k_round_fused_glds<3, 0> and <2, 1> for the fused SKIP1 + LEAD rounds over three tables); 0 = k_round0_dot29 / k_round_kd at
// every size (A/B).  Sizes: a multiple of 64 pairs and at least 2^21 (round 0, two tables), 2^20 (round 0, product plus term), 2^16 /
// 2^19 (fused rounds of the (3, 3) product / of the product-plus-term shape; the fused rounds of the two-table product gain nothing at
// any size and stay on k_round_kd) -- measured cross-overs, profiles/r06_glds_sizes.log; ZK_ROUND_GLDS_MIN_PAIRS replaces
// all of them (the parity sweeps force 64).  The half tables of a fused round bypass the caches from ZK_ROUND_GLDS_NT_MIN_PAIRS pairs up.
static bool glds_rounds() {
    static const bool v = env_u64("ZK_ROUND_GLDS", 1, 0, 1) != 0;
    return v;
}
static uint64_t glds_min_pairs(uint64_t unset) {
    static const uint64_t v = env_u64("ZK_ROUND_GLDS_MIN_PAIRS", 0, 0, (uint64_t)1 << 40);
    return v ? v : unset;
}
static uint64_t glds_nt_min_pairs() {
    static const uint64_t v = env_u64("ZK_ROUND_GLDS_NT_MIN_PAIRS", (uint64_t)1 << 20, 64, (uint64_t)1 << 40);
    return v;
}
static bool glds_takes(uint64_t q, uint64_t min_pairs) { return glds_rounds() && (q & 63) == 0 && q >= 64 && q >= glds_min_pairs(min_pairs); }
static uint64_t quad_max_pairs() {
    static const uint64_t v = env_u64("ZK_QUAD_MAX_PAIRS", (uint64_t)1 << 15, 0, (uint64_t)1 << 40);
    return v;
}
template <int K, int D, int EXTRA>
static uint32_t launch_quad(const RoundLaunchCtx &lc, const FactorPtrs &fp, uint64_t q, const uint64_t *d_r) {
    // 64 pair indices per workgroup pass, at most kMaxLazy products per lane
    uint64_t g = (q + 63) / 64;
    const uint64_t cap = (q + 64ull * kMaxLazy - 1) / (64ull * kMaxLazy);
    if (g > 2048) g = cap > 2048 ? cap : 2048;
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    uint64_t *part = lc.d_partials;
    const uint32_t grid = (uint32_t)g;
    auto single = [=]() {
        k_round_quad<K, D, EXTRA><<<grid, kBlock, 0, st>>>(fp, q, *P, d_r, part);
        return hipGetLastError();
    };
    if (!batch_record(BK_ROUND_QUAD, (uint32_t)K | ((uint32_t)D << 4) | ((uint32_t)EXTRA << 9), grid, kBlock, 0, q, 0, 0, 0,
                      RoundSlot{factor_ptrs4(fp), d_r, part, ClaimJob{}}, single))
        (void)single();
    return grid;
}
// the product-plus-term shape (a GKR layer: three tables, 2300-2700 instructions per pair index) is better off with ONE wave per SIMD
// and twice the pairs per thread below 2^17 pairs: 7.57 -> 7.50 ms on the depth-8 x 2^20 driver (profiles/r04_round_min_blocks_ab.log);
// ZK_ROUND_MIN_BLOCKS overrides both
static uint32_t min_blocks_plus1() {
    static const uint32_t v = (uint32_t)env_u64("ZK_ROUND_MIN_BLOCKS", 256, 1, 2048);
    return v;
}
static inline uint32_t round_grid(uint64_t q, uint32_t min_b = min_blocks()) {
    uint64_t b = (q + (uint64_t)kBlock * kMaxLazy - 1) / ((uint64_t)kBlock * kMaxLazy);   // kMaxLazy pairs per thread
    const uint64_t one_pair = (q + kBlock - 1) / kBlock;                                    // one pair per thread
    const uint64_t floor_b = one_pair < min_b ? one_pair : min_b;
    if (b < floor_b) b = floor_b;
    return (uint32_t)(b ? b : 1);
}
static inline uint32_t capped_grid(uint64_t q) {
    uint64_t b = (q + kBlock - 1) / kBlock;
    if (b > kCapGrid) b = kCapGrid;
    return (uint32_t)(b ? b : 1);
}

// a SKIP1 kernel launched with a claim job carries one workgroup more than its work grid (round_kernels.cuh)
static inline uint32_t claim_blocks(const RoundLaunchCtx &lc) { return lc.claim.out ? 1u : 0u; }

thread_local BatchRecorder *g_batch = nullptr;

// every launch of a round kernel goes through one of these: launch now, or -- inside zk_sumcheck_prove_batch -- hand the launch to the
// recorder (launch.hpp) to be merged with the other proofs' launches of the same step
constexpr uint32_t kd_shape(int K, int D, bool FUSED, int EXTRA, bool SKIP1, bool LEAD) {
    return (uint32_t)K | ((uint32_t)D << 4) | ((uint32_t)FUSED << 8) | ((uint32_t)EXTRA << 9) | ((uint32_t)SKIP1 << 10) | ((uint32_t)LEAD << 11);
}
template <int K, int D, bool FUSED, int EXTRA = 0, bool SKIP1 = false, bool LEAD = false>
static void go_kd(const RoundLaunchCtx &lc, const FactorPtrs &fp, uint64_t q, const uint64_t *d_r, uint32_t grid, const ClaimJob &cj = {}) {
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    uint64_t *part = lc.d_partials;
    auto single = [=]() {
        k_round_kd<K, D, FUSED, EXTRA, SKIP1, LEAD><<<grid, kBlock, 0, st>>>(fp, q, *P, d_r, part, cj);
        return hipGetLastError();
    };
    if (batch_record(BK_ROUND_KD, kd_shape(K, D, FUSED, EXTRA, SKIP1, LEAD), grid, kBlock, 0, q, 0, 0, 0, RoundSlot{factor_ptrs4(fp), d_r, part, cj}, single))
        return;
    (void)single();
}
template <int EXTRA>
static void go_round0_dot29(const RoundLaunchCtx &lc, const FactorPtrs &fp, uint64_t q, uint32_t grid) {
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    uint64_t *part = lc.d_partials;
    auto single = [=]() {
        k_round0_dot29<EXTRA><<<grid, kBlock, 0, st>>>(fp, q, *P, part);
        return hipGetLastError();
    };
    if (batch_record(BK_ROUND0_DOT29, (uint32_t)EXTRA, grid, kBlock, 0, q, 0, 0, 0, RoundSlot{factor_ptrs4(fp), nullptr, part, ClaimJob{}}, single)) return;
    (void)single();
}

// the LDS-DMA forms (round_kernels.cuh): shape bit 12 marks them in a batch record, bit 13 the nontemporal half tables
constexpr uint32_t kShapeGlds = 1u << 12, kShapeGldsNt = 1u << 13;
template <int EXTRA>
static void go_round0_glds(const RoundLaunchCtx &lc, const FactorPtrs &fp, uint64_t q, uint32_t grid) {
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    uint64_t *part = lc.d_partials;
    constexpr uint32_t lds = EXTRA ? kGldsRing3Bytes : kGldsRingBytes;
    auto single = [=]() {
        k_round0_glds<EXTRA><<<grid, kBlock, lds, st>>>(fp, q, *P, part);
        return hipGetLastError();
    };
    if (batch_record(BK_ROUND0_DOT29, kShapeGlds | (uint32_t)EXTRA, grid, kBlock, lds, q, 0, 0, 0, RoundSlot{factor_ptrs4(fp), nullptr, part, ClaimJob{}}, single)) return;
    (void)single();
}
template <int K, int EXTRA>
static void go_fused_glds(const RoundLaunchCtx &lc, const FactorPtrs &fp, uint64_t q, const uint64_t *d_r, uint32_t grid, const ClaimJob &cj) {
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    uint64_t *part = lc.d_partials;
    const bool nt = q >= glds_nt_min_pairs();
    auto single = [=]() {
        if (nt) k_round_fused_glds<K, EXTRA, true><<<grid, kBlock, kGldsRingBytes, st>>>(fp, q, *P, d_r, part, cj);
        else k_round_fused_glds<K, EXTRA, false><<<grid, kBlock, kGldsRingBytes, st>>>(fp, q, *P, d_r, part, cj);
        return hipGetLastError();
    };
    if (batch_record(BK_ROUND_KD, kd_shape(K, K, true, EXTRA, true, true) | kShapeGlds | (nt ? kShapeGldsNt : 0u), grid, kBlock, kGldsRingBytes, q, 0, 0, 0,
                     RoundSlot{factor_ptrs4(fp), d_r, part, cj}, single))
        return;
    (void)single();
}

template <int K, int D>
static void launch_kd(const RoundLaunchCtx &lc, const FactorPtrs &fp, uint64_t q, bool fused, const uint64_t *d_r, uint32_t g) {
    if (fused) go_kd<K, D, true>(lc, fp, q, d_r, g);
    else go_kd<K, D, false>(lc, fp, q, d_r, g);
}
template <int D>
static void launch_generic(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, bool fused, const uint64_t *d_r, uint32_t g) {
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    uint64_t *part = lc.d_partials;
    auto single = [=]() {
        if (fused) k_round<D, true><<<g, kBlock, 0, st>>>(fp, k, q, *P, d_r, part);
        else k_round<D, false><<<g, kBlock, 0, st>>>(fp, k, q, *P, d_r, part);
        return hipGetLastError();
    };
    if (!batch_record_other(single)) (void)single();   // (runtime-k shapes have no batched twin: a batch replays them proof by proof)
}

int launch_round(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, uint32_t D, bool fused,
                 const uint64_t *d_r, uint32_t *out_grid, bool *skip1, bool *lead) {
    if (D < 1 || D > 4 || k < 1 || k > kMaxFactors) return kLaunchUnsupported;
    uint32_t g = round_grid(q);
    bool no_lead = false;
    if (!lead) lead = &no_lead;
    // LEAD variants: (2,2) and (3,3), sums-only (round 0) or fused with SKIP1 (the big rounds after it)
    if (*lead && !((k == 2 && D == 2) || (k == 3 && D == 3))) *lead = false;
    if (*lead && (uint64_t)g * (D + 1) <= lc.capacity_elems) {
        const int shl = k * 10 + (int)D;
        bool done = true;
        if (!fused) {
            if (shl == 22 && round0_dot29() && glds_takes(q, (uint64_t)1 << 21)) {
                if (g > 512) g = 512;   // two workgroups per CU; the kernel reduces its columns every kMaxLazy pair indices itself
                go_round0_glds<0>(lc, fp, q, g);
            } else if (shl == 22 && round0_dot29()) go_round0_dot29<0>(lc, fp, q, g);
            else if (shl == 22) go_kd<2, 2, false, 0, false, true>(lc, fp, q, d_r, g);
            else go_kd<3, 3, false, 0, false, true>(lc, fp, q, d_r, g);
            if (skip1) *skip1 = false;
        } else if (skip1 && *skip1) {
            if (shl == 33 && glds_takes(q, (uint64_t)1 << 16)) go_fused_glds<3, 0>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
            else if (shl == 22) go_kd<2, 2, true, 0, true, true>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
            else go_kd<3, 3, true, 0, true, true>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
        } else {
            done = false;
        }
        if (done) {
            if (hipGetLastError() != hipSuccess) return kLaunchHipError;
            *out_grid = g;
            return kLaunchOk;
        }
    }
    *lead = false;
    if (fused && q >= 16 && q <= 4 * quad_max_pairs() && (uint64_t)2048 * (D + 1) <= lc.capacity_elems) {
        // measured cross-over (MI355X, BN254): 2^15 pairs for k = 2, 2^17 for k = 3 (its lane does 14 multiplies per pair index)
        const int shq = k * 10 + (int)D;
        uint32_t gq = 0;
        if (shq == 22 && q <= quad_max_pairs()) gq = launch_quad<2, 2, 0>(lc, fp, q, d_r);
        else if (shq == 33) gq = launch_quad<3, 3, 0>(lc, fp, q, d_r);
        if (gq) {
            if (skip1) *skip1 = false;
            if (hipGetLastError() != hipSuccess) return kLaunchHipError;
            *out_grid = gq;
            return kLaunchOk;
        }
    }
    if (skip1 && *skip1) {   // the variants without the t = 1 products exist for the GKR-style shapes, fused only
        const bool fits1 = (uint64_t)g * (D + 1) <= lc.capacity_elems;
        const int shape1 = (fits1 && fused) ? k * 10 + (int)D : 0;
        if (shape1 == 22) go_kd<2, 2, true, 0, true>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
        else if (shape1 == 33) go_kd<3, 3, true, 0, true>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
        else *skip1 = false;
        if (*skip1) {
            if (hipGetLastError() != hipSuccess) return kLaunchHipError;
            *out_grid = g;
            return kLaunchOk;
        }
    }
    const bool fits = (uint64_t)g * (D + 1) <= lc.capacity_elems;
    // specialised shapes (GKR-style products have D = k): everything else takes the runtime-k kernel
    const int shape = fits ? k * 10 + (int)D : 0;
    switch (shape) {
        case 11: launch_kd<1, 1>(lc, fp, q, fused, d_r, g); break;
        case 12: launch_kd<1, 2>(lc, fp, q, fused, d_r, g); break;
        case 21: launch_kd<2, 1>(lc, fp, q, fused, d_r, g); break;
        case 22: launch_kd<2, 2>(lc, fp, q, fused, d_r, g); break;
        case 23: launch_kd<2, 3>(lc, fp, q, fused, d_r, g); break;
        case 32: launch_kd<3, 2>(lc, fp, q, fused, d_r, g); break;
        case 33: launch_kd<3, 3>(lc, fp, q, fused, d_r, g); break;
        default:
            g = capped_grid(q);   // k_round flushes its lazy accumulators itself
            switch (D) {
                case 1: launch_generic<1>(lc, fp, k, q, fused, d_r, g); break;
                case 2: launch_generic<2>(lc, fp, k, q, fused, d_r, g); break;
                case 3: launch_generic<3>(lc, fp, k, q, fused, d_r, g); break;
                default: launch_generic<4>(lc, fp, k, q, fused, d_r, g); break;
            }
    }
    if (hipGetLastError() != hipSuccess) return kLaunchHipError;
    *out_grid = g;
    return kLaunchOk;
}

// Terms {k, 1}: the k-factor product plus one single-factor term in one pass (fp lists the k factors, then the extra one).
int launch_round_plus1(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, uint32_t D, bool fused,
                       const uint64_t *d_r, uint32_t *out_grid, bool *skip1, bool *lead) {
    uint32_t g = round_grid(q, min_blocks_plus1());
    if ((uint64_t)g * (D + 1) > lc.capacity_elems) return kLaunchUnsupported;
    const int shape = k * 10 + (int)D;
    bool no_lead = false;
    if (!lead) lead = &no_lead;
    if (*lead && shape == 22 && (!fused || (skip1 && *skip1))) {   // the GKR layer polynomial W*H + B: sums-only or fused + SKIP1
        if (!fused) {
            // (the carry-free round-0 kernel with a third table has no registers left for the second prefetch buffer and measures
            // 0.1-0.4 % SLOWER on the GKR driver: profiles/r04_round0_dot29_ab.log; ZK_ROUND0_DOT29=2 selects it for A/B runs)
            if (round0_dot29_extra()) go_round0_dot29<1>(lc, fp, q, g);
            else if (round0_dot29() && glds_takes(q, (uint64_t)1 << 20)) {
                if (g > 512) g = 512;
                go_round0_glds<1>(lc, fp, q, g);
            } else go_kd<2, 2, false, 1, false, true>(lc, fp, q, d_r, g);
            if (skip1) *skip1 = false;
        } else {
            if (glds_takes(q, (uint64_t)1 << 19)) go_fused_glds<2, 1>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
            else go_kd<2, 2, true, 1, true, true>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
        }
        if (hipGetLastError() != hipSuccess) return kLaunchHipError;
        *out_grid = g;
        return kLaunchOk;
    }
    *lead = false;
    if (fused && shape == 22 && q >= 16 && q <= quad_max_pairs() && (uint64_t)2048 * (D + 1) <= lc.capacity_elems) {
        const uint32_t gq = launch_quad<2, 2, 1>(lc, fp, q, d_r);
        if (skip1) *skip1 = false;
        if (hipGetLastError() != hipSuccess) return kLaunchHipError;
        *out_grid = gq;
        return kLaunchOk;
    }
    if (skip1 && *skip1 && !(shape == 22 && fused)) *skip1 = false;
    if (shape == 22) {
        if (fused && skip1 && *skip1) go_kd<2, 2, true, 1, true>(lc, fp, q, d_r, g + claim_blocks(lc), lc.claim);
        else if (fused) go_kd<2, 2, true, 1>(lc, fp, q, d_r, g);
        else go_kd<2, 2, false, 1>(lc, fp, q, d_r, g);
    } else if (shape == 33) {
        if (fused) go_kd<3, 3, true, 1>(lc, fp, q, d_r, g);
        else go_kd<3, 3, false, 1>(lc, fp, q, d_r, g);
    } else {
        return kLaunchUnsupported;
    }
    if (hipGetLastError() != hipSuccess) return kLaunchHipError;
    *out_grid = g;
    return kLaunchOk;
}

int launch_round_single_t(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, const Fe &tval, uint32_t *out_grid) {
    const uint32_t g = capped_grid(q);
    {
        const FieldParams *P = lc.P;
        hipStream_t st = lc.stream;
        uint64_t *part = lc.d_partials;
        auto single = [=]() {
            k_round_single_t<<<g, kBlock, 0, st>>>(fp, k, q, *P, tval, part);
            return hipGetLastError();
        };
        if (!batch_record_other(single)) (void)single();
    }
    if (hipGetLastError() != hipSuccess) return kLaunchHipError;
    *out_grid = g;
    return kLaunchOk;
}

// ---- batched twins (zk_sumcheck_prove_batch): launch idx of every proof of the batch as ONE launch, grid (x, proofs) ------------------
// Instantiated for the shapes the configs use -- products of two or three tables with D = K, no extra term -- on every kernel the
// shipped thresholds select for them; anything else returns kLaunchUnsupported and is replayed proof by proof.
int batch_launch_rounds(const BatchRecorder &r, size_t idx) {
    const BatchRecord &r0 = r.recs[0][idx];
    BatchOf<RoundSlot> slots;
    batch_gather(r, idx, slots);
    const dim3 grid(r0.grid, (uint32_t)r.n);
    const uint64_t q = r0.s[0];
    const FieldParams &P = *r.P;
#define ZK_KD_B(K, D, F, E, S, L)                                                                  \
    case kd_shape(K, D, F, E, S, L):                                                               \
        k_round_kd_b<K, D, F, E, S, L><<<grid, kBlock, 0, r.stream>>>(slots, q, P);               \
        break;
    if (r0.kernel == BK_ROUND_KD && (r0.shape & kShapeGlds)) {
        const uint32_t base = r0.shape & ~(kShapeGlds | kShapeGldsNt);
        const bool nt = (r0.shape & kShapeGldsNt) != 0;
        if (base == kd_shape(3, 3, true, 0, true, true)) {
            if (nt) k_round_fused_glds_b<3, 0, true><<<grid, kBlock, kGldsRingBytes, r.stream>>>(slots, q, P);
            else k_round_fused_glds_b<3, 0, false><<<grid, kBlock, kGldsRingBytes, r.stream>>>(slots, q, P);
        } else {
            return kLaunchUnsupported;
        }
    } else if (r0.kernel == BK_ROUND_KD) {
        switch (r0.shape) {
            ZK_KD_B(2, 2, true, 0, true, true)     // the big fused rounds (SKIP1 + LEAD)
            ZK_KD_B(3, 3, true, 0, true, true)
            ZK_KD_B(3, 3, false, 0, false, true)   // round 0 of three tables (LEAD)
            ZK_KD_B(2, 2, true, 0, false, false)   // plain fused / sums-only rounds (small sizes, thresholds moved by the tests)
            ZK_KD_B(3, 3, true, 0, false, false)
            ZK_KD_B(2, 2, false, 0, false, false)
            ZK_KD_B(3, 3, false, 0, false, false)
            default: return kLaunchUnsupported;
        }
    } else if (r0.kernel == BK_ROUND0_DOT29 && r0.shape == kShapeGlds) {
        k_round0_glds_b<0><<<grid, kBlock, kGldsRingBytes, r.stream>>>(slots, q, P);
    } else if (r0.kernel == BK_ROUND0_DOT29 && r0.shape == (kShapeGlds | 1u)) {
        k_round0_glds_b<1><<<grid, kBlock, kGldsRing3Bytes, r.stream>>>(slots, q, P);
    } else if (r0.kernel == BK_ROUND0_DOT29) {
        if (r0.shape != 0) return kLaunchUnsupported;
        k_round0_dot29_b<0><<<grid, kBlock, 0, r.stream>>>(slots, q, P);
    } else if (r0.kernel == BK_ROUND_QUAD) {
        if (r0.shape == (2u | (2u << 4))) k_round_quad_b<2, 2, 0><<<grid, kBlock, 0, r.stream>>>(slots, q, P);
        else if (r0.shape == (3u | (3u << 4))) k_round_quad_b<3, 3, 0><<<grid, kBlock, 0, r.stream>>>(slots, q, P);
        else return kLaunchUnsupported;
    } else {
        return kLaunchUnsupported;
    }
#undef ZK_KD_B
    return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchHipError;
}

}  // namespace zk
