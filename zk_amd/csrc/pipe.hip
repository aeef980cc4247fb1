// pipe.hip -- instantiations + launch logic of the pipelined rounds (pipe_kernels.cuh).
#include <cstdlib>

#include "launch.hpp"
#include "pipe_kernels.cuh"

namespace zk {

bool pipe_shape_ok(int k, uint32_t D, int extra) {
    const int shape = k * 100 + (int)D * 10 + extra;
    return shape == 110 || shape == 120 || shape == 220 || shape == 230 || shape == 330 || shape == 221;
}
uint32_t pipe_values_per_block(int k, uint32_t D) { return (D + 1) * (uint32_t)(k + 1); }
static uint32_t pipe_threads(int k, uint32_t D, int extra) {
    switch (k * 100 + (int)D * 10 + extra) {
        case 110: return pipe_block_threads<1, 1, 0>();
        case 120: return pipe_block_threads<1, 2, 0>();
        case 220: return pipe_block_threads<2, 2, 0>();
        case 230: return pipe_block_threads<2, 3, 0>();
        case 330: return pipe_block_threads<3, 3, 0>();
        case 221: return pipe_block_threads<2, 2, 1>();
        default: return 256;
    }
}
uint32_t pipe_rows_per_block(int k, uint32_t D, int extra) { return pipe_threads(k, D, extra) / 16; }
uint32_t pipe_work_blocks(int k, uint32_t D, int extra, uint64_t q) {
    // Four passes per block once there is work for more than 64 blocks: a launch ends when its LAST block has reduced every
    // block's partials (~1 us per 128 of them), so few blocks with a few passes each (~2 us per pass) beat many with one
    const uint32_t rows = pipe_rows_per_block(k, D, extra);
    uint64_t g = (q + rows - 1) / rows;
    if (g > 64) {
        g = (g + 3) / 4;
        if (g < 64) g = 64;
    }
    if (g > kPipeMaxWorkBlocks) g = kPipeMaxWorkBlocks;
    return (uint32_t)(g ? g : 1);
}

// The fence-free partial hand-off of k_round_pipe (measured -1.2 % GKR / 0 % prover) is an A/B BUILD option
// only (make CXXFLAGS+=-DZK_PIPE_SC1_HANDOFF=1): relaxed write-through stores + a relaxed counter carry no release / acquire
// edge in the HIP memory model, so the shipped library has no run-time switch that could select it.
#ifndef ZK_PIPE_SC1_HANDOFF
#define ZK_PIPE_SC1_HANDOFF 0
#endif
static constexpr int sc1_handoff() { return ZK_PIPE_SC1_HANDOFF; }
template <int K, int D, int EXTRA>
static void launch_shape(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t g) {
    constexpr int kThreads = pipe_block_threads<K, D, EXTRA>();
    if (pl.fold)
        k_round_pipe<K, D, EXTRA, true><<<g + 1, kThreads, 0, lc.stream>>>(fp, pl.q, pl.emit, *lc.P, pl.chal_fold, pl.e_partials, pl.done_counter, pl.tail, sc1_handoff());
    else
        k_round_pipe<K, D, EXTRA, false><<<g + 1, kThreads, 0, lc.stream>>>(fp, pl.q, pl.emit, *lc.P, pl.chal_fold, pl.e_partials, pl.done_counter, pl.tail, sc1_handoff());
}

// k_round_mid: one pair index per quad and pass.  As many blocks as one pass needs, up to the cap (the launch ends with ONE block
// adding all the block partials up: ~1 us per 128 of them), then more passes (at most kMaxLazy: unreduced products per lane).
uint64_t mid_max_pairs() { return (uint64_t)kMidMaxWorkBlocks * kMidQuads * kMaxLazy; }
static uint32_t mid_work_blocks(uint64_t q) {
    uint64_t g = (q + kMidQuads - 1) / kMidQuads;
    if (g > kMidMaxWorkBlocks) g = kMidMaxWorkBlocks;
    return (uint32_t)(g ? g : 1);
}
template <int K, int D, int EXTRA>
static void launch_mid_shape(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t g) {
    if (pl.fold)
        k_round_mid<K, D, EXTRA, true><<<g + 1, kMidThreads, 0, lc.stream>>>(fp, pl.q, pl.emit, *lc.P, pl.chal_fold, pl.e_partials, pl.done_counter, pl.tail, pl.mid_total ? 1 : 0);
    else
        k_round_mid<K, D, EXTRA, false><<<g + 1, kMidThreads, 0, lc.stream>>>(fp, pl.q, pl.emit, *lc.P, pl.chal_fold, pl.e_partials, pl.done_counter, pl.tail, pl.mid_total ? 1 : 0);
}
static int launch_round_mid(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t *out_work_blocks) {
    if (pl.q > mid_max_pairs()) return kLaunchUnsupported;
    const uint32_t g = mid_work_blocks(pl.q);
    switch (pl.k * 100 + (int)pl.D * 10 + pl.extra) {
        case 110: launch_mid_shape<1, 1, 0>(lc, fp, pl, g); break;
        case 120: launch_mid_shape<1, 2, 0>(lc, fp, pl, g); break;
        case 220: launch_mid_shape<2, 2, 0>(lc, fp, pl, g); break;
        case 230: launch_mid_shape<2, 3, 0>(lc, fp, pl, g); break;
        case 330: launch_mid_shape<3, 3, 0>(lc, fp, pl, g); break;
        case 221: launch_mid_shape<2, 2, 1>(lc, fp, pl, g); break;
        default: return kLaunchUnsupported;
    }
    if (hipGetLastError() != hipSuccess) return kLaunchHipError;
    *out_work_blocks = g;
    return kLaunchOk;
}

int launch_round_pipe(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t *out_work_blocks) {
    if (!pipe_shape_ok(pl.k, pl.D, pl.extra)) return kLaunchUnsupported;
    if (pl.mid) return launch_round_mid(lc, fp, pl, out_work_blocks);
    const uint32_t g = pipe_work_blocks(pl.k, pl.D, pl.extra, pl.q);
    switch (pl.k * 100 + (int)pl.D * 10 + pl.extra) {
        case 110: launch_shape<1, 1, 0>(lc, fp, pl, g); break;
        case 120: launch_shape<1, 2, 0>(lc, fp, pl, g); break;
        case 220: launch_shape<2, 2, 0>(lc, fp, pl, g); break;
        case 230: launch_shape<2, 3, 0>(lc, fp, pl, g); break;
        case 330: launch_shape<3, 3, 0>(lc, fp, pl, g); break;
        case 221: launch_shape<2, 2, 1>(lc, fp, pl, g); break;
        default: return kLaunchUnsupported;
    }
    if (hipGetLastError() != hipSuccess) return kLaunchHipError;
    *out_work_blocks = g;
    return kLaunchOk;
}


// ---- the pipelined finisher ----
size_t finish_pipe_lds_bytes(int k, int extra, uint32_t m_in) {
    return (size_t)(k + extra) * ((size_t)32 << (m_in - 1)) + kFinMiscBytes;
}
template <int K, int D, int EXTRA>
static hipError_t launch_fin_shape(const RoundLaunchCtx &lc, const FactorPtrs &fp, const FinishPipeLaunch &fl) {
    const size_t lds = finish_pipe_lds_bytes(K, EXTRA, fl.m_in);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_finish_pipe<K, D, EXTRA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_finish_pipe<K, D, EXTRA><<<1, kFinishPipeThreads, lds, lc.stream>>>(fp, fl.m_in, fl.entry, fl.e_partials, fl.e_blocks, *lc.P, fl.pc, fl.chal_in,
                                                                         fl.chal_last, fl.sponge, fl.out_rp, fl.out_ch, fl.out_final, fl.dbg, fl.pub);
    return hipGetLastError();
}
int launch_finish_pipe(const RoundLaunchCtx &lc, const FactorPtrs &fp, const FinishPipeLaunch &fl) {
    if (!pipe_shape_ok(fl.k, fl.D, fl.extra) || fl.m_in < 3 || fl.m_in > (uint32_t)finish_pipe_vars(fl.k + fl.extra) + 1) return kLaunchUnsupported;
    hipError_t e;
    switch (fl.k * 100 + (int)fl.D * 10 + fl.extra) {
        case 110: e = launch_fin_shape<1, 1, 0>(lc, fp, fl); break;
        case 120: e = launch_fin_shape<1, 2, 0>(lc, fp, fl); break;
        case 220: e = launch_fin_shape<2, 2, 0>(lc, fp, fl); break;
        case 230: e = launch_fin_shape<2, 3, 0>(lc, fp, fl); break;
        case 330: e = launch_fin_shape<3, 3, 0>(lc, fp, fl); break;
        case 221: e = launch_fin_shape<2, 2, 1>(lc, fp, fl); break;
        default: return kLaunchUnsupported;
    }
    return e == hipSuccess ? kLaunchOk : kLaunchHipError;
}
uint32_t finish_pipe_max_vars(int n_factors) { return (uint32_t)finish_pipe_vars(n_factors); }

}  // namespace zk
