// pipe.hip -- instantiations + launch logic of the pipelined rounds (pipe_kernels.cuh).
#include <cstddef>
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
constexpr uint32_t pipe_shape_id(int K, int D, int EXTRA, bool FOLD) { return (uint32_t)K | ((uint32_t)D << 4) | ((uint32_t)EXTRA << 8) | ((uint32_t)FOLD << 9); }
template <int K, int D, int EXTRA>
static void launch_shape(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t g) {
    constexpr int kThreads = pipe_block_threads<K, D, EXTRA>();
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    const PipeLaunch l = pl;
    auto single = [=]() {
        if (l.fold)
            k_round_pipe<K, D, EXTRA, true><<<g + 1, kThreads, 0, st>>>(fp, l.q, l.emit, *P, l.chal_fold, l.e_partials, l.done_counter, l.tail, sc1_handoff());
        else
            k_round_pipe<K, D, EXTRA, false><<<g + 1, kThreads, 0, st>>>(fp, l.q, l.emit, *P, l.chal_fold, l.e_partials, l.done_counter, l.tail, sc1_handoff());
        return hipGetLastError();
    };
    // (the shared arguments of the batched twin: pairs, emit, and what the transcript blocks do -- mode, block count, values per block)
    if (batch_record(BK_PIPE, pipe_shape_id(K, D, EXTRA, pl.fold), g + 1, kThreads, 0, pl.q, (uint64_t)pl.emit,
                     ((uint64_t)(uint32_t)pl.tail.mode << 32) | pl.tail.nblocks, pl.tail.n_in,
                     PipeSlot{factor_ptrs4(fp), pl.chal_fold, pl.e_partials, pl.done_counter, pl.tail}, single))
        return;
    (void)single();
}

int launch_spin_us(hipStream_t stream, uint32_t us) {
    k_spin_us<<<1, 1, 0, stream>>>(us);
    return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchHipError;
}

int launch_round_pipe(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t *out_work_blocks) {
    if (!pipe_shape_ok(pl.k, pl.D, pl.extra)) return kLaunchUnsupported;
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
struct FinishRec {   // what a finisher launch records for a batch: its slot, then the (field-wide) constants the batched twin takes once
    FinishSlot slot;
    PipeConsts pc;
};
template <int K, int D, int EXTRA>
static hipError_t launch_fin_shape(const RoundLaunchCtx &lc, const FactorPtrs &fp, const FinishPipeLaunch &fl) {
    const size_t lds = finish_pipe_lds_bytes(K, EXTRA, fl.m_in);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_finish_pipe<K, D, EXTRA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const FieldParams *P = lc.P;
    hipStream_t st = lc.stream;
    const FinishPipeLaunch l = fl;
    auto single = [=]() {
        k_finish_pipe<K, D, EXTRA><<<1, kFinishPipeThreads, lds, st>>>(fp, l.m_in, l.entry, l.e_partials, l.e_blocks, *P, l.pc, l.chal_in, l.chal_last, l.sponge,
                                                                      l.out_rp, l.out_ch, l.out_final, l.dbg, l.pub);
        return hipGetLastError();
    };
    if (g_batch) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_finish_pipe_b<K, D, EXTRA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (batch_record(BK_FINISH_PIPE, pipe_shape_id(K, D, EXTRA, false), 1, kFinishPipeThreads, lds, fl.m_in, (uint64_t)fl.entry, fl.e_blocks, 0,
                     FinishRec{FinishSlot{factor_ptrs4(fp), fl.e_partials, fl.chal_in, fl.chal_last, fl.sponge, fl.out_rp, fl.out_ch, fl.out_final, fl.pub}, fl.pc},
                     single))
        return hipSuccess;
    return single();
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

// ---- batched twins (zk_sumcheck_prove_batch): grid (x, proofs); shapes (2, 2) and (3, 3) without an extra term ------------------------
int batch_launch_pipe(const BatchRecorder &r, size_t idx) {
    const BatchRecord &r0 = r.recs[0][idx];
    const dim3 grid(r0.grid, (uint32_t)r.n);
    const FieldParams &P = *r.P;
    if (r0.kernel == BK_PIPE) {
        BatchOf<PipeSlot> slots;
        batch_gather(r, idx, slots);
        const uint64_t q = r0.s[0];
        const int emit = (int)r0.s[1];
        switch (r0.shape) {
            case pipe_shape_id(2, 2, 0, true): k_round_pipe_b<2, 2, 0, true><<<grid, r0.block, 0, r.stream>>>(slots, q, emit, P, sc1_handoff()); break;
            case pipe_shape_id(2, 2, 0, false): k_round_pipe_b<2, 2, 0, false><<<grid, r0.block, 0, r.stream>>>(slots, q, emit, P, sc1_handoff()); break;
            case pipe_shape_id(3, 3, 0, true): k_round_pipe_b<3, 3, 0, true><<<grid, r0.block, 0, r.stream>>>(slots, q, emit, P, sc1_handoff()); break;
            case pipe_shape_id(3, 3, 0, false): k_round_pipe_b<3, 3, 0, false><<<grid, r0.block, 0, r.stream>>>(slots, q, emit, P, sc1_handoff()); break;
            default: return kLaunchUnsupported;
        }
    } else if (r0.kernel == BK_FINISH_PIPE) {
        BatchOf<FinishSlot> slots;
        batch_gather(r, idx, slots);
        PipeConsts pc;   // every proof's launch carried the same constants (one field per batch): proof 0's copy
        std::memcpy(&pc, r0.slot + offsetof(FinishRec, pc), sizeof pc);
        const uint32_t m_in = (uint32_t)r0.s[0], e_blocks = (uint32_t)r0.s[2];
        const int entry = (int)r0.s[1];
        switch (r0.shape) {
            case pipe_shape_id(2, 2, 0, false): k_finish_pipe_b<2, 2, 0><<<grid, r0.block, r0.lds, r.stream>>>(slots, m_in, entry, e_blocks, P, pc); break;
            case pipe_shape_id(3, 3, 0, false): k_finish_pipe_b<3, 3, 0><<<grid, r0.block, r0.lds, r.stream>>>(slots, m_in, entry, e_blocks, P, pc); break;
            default: return kLaunchUnsupported;
        }
    } else {
        return kLaunchUnsupported;
    }
    return hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchHipError;
}

}  // namespace zk
