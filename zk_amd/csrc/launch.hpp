// launch.hpp -- host-side launcher of the round kernels (defined in rounds.hip, used by capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#include "common.cuh"

namespace zk {

// ---- batched proving (zk_sumcheck_prove_batch): record, then merge ------------------------------------------------------------------
// The host schedule of a proof (capi.hip: prover_step and everything under it) decides kernel, grid and arguments per launch.  A batch
// runs that SAME code once per proof with a recorder installed (thread-local): every launch site on the prover path hands its launch to
// batch_record() instead of launching.  Proofs of one shape and size produce identical launch sequences that differ in their pointers
// only, so launch i of all proofs becomes ONE launch of the kernel's batched twin (blockIdx.y = proof; the per-proof pointers travel as
// an array of slots in the kernel arguments).  A step with no batched twin -- or whose shared arguments differ between proofs -- is
// replayed proof by proof through the recorded single-proof closures: slower, never wrong.
enum BatchKernel : int { BK_STORE_SPONGE = 1, BK_ROUND_KD, BK_ROUND0_DOT29, BK_ROUND_QUAD, BK_TAIL, BK_PIPE, BK_FINISH_PIPE, BK_OTHER };
struct BatchRecord {
    int kernel;
    uint32_t shape;      // template selector of the kernel (its own encoding)
    uint32_t grid, block;
    size_t lds;
    uint64_t s[4];       // arguments shared by the proofs (sizes, counts, flags): the merge requires them equal
    std::function<hipError_t()> single;   // this proof's launch on its own
    alignas(16) unsigned char slot[384];  // this proof's slot of the batched twin's argument array
};
struct BatchRecorder {
    int n = 0;           // proofs in the batch
    int cur = 0;         // proof being recorded
    hipStream_t stream = nullptr;
    const FieldParams *P = nullptr;
    std::vector<BatchRecord> recs[kMaxBatch];
    uint64_t merged = 0, replayed = 0;   // launches issued batched / proof by proof (reported by zk_sumcheck_prove_batch)
};
extern thread_local BatchRecorder *g_batch;
template <class Slot, class F>
inline bool batch_record(int kernel, uint32_t shape, uint32_t grid, uint32_t block, size_t lds, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3,
                         const Slot &slot, F &&single) {
    BatchRecorder *r = g_batch;
    if (!r) return false;
    static_assert(sizeof(Slot) <= sizeof(BatchRecord::slot), "slot does not fit a batch record");
    static_assert(std::is_trivially_copyable<Slot>::value, "slots are plain data");
    r->recs[r->cur].emplace_back();
    BatchRecord &rec = r->recs[r->cur].back();
    rec.kernel = kernel;
    rec.shape = shape;
    rec.grid = grid;
    rec.block = block;
    rec.lds = lds;
    rec.s[0] = s0, rec.s[1] = s1, rec.s[2] = s2, rec.s[3] = s3;
    rec.single = std::function<hipError_t()>(std::forward<F>(single));
    std::memcpy(rec.slot, &slot, sizeof(Slot));
    return true;
}
// a launch with no batched twin: recorded for replay only
template <class F>
inline bool batch_record_other(F &&single) {
    struct Nothing {
        int x;
    } none = {0};
    return batch_record(BK_OTHER, 0, 0, 0, 0, 0, 0, 0, 0, none, std::forward<F>(single));
}
// the batched twins, one dispatcher per translation unit (kLaunchUnsupported: no twin for this shape -> replay)
int batch_launch_rounds(const BatchRecorder &r, size_t idx);   // rounds.hip: BK_ROUND_KD, BK_ROUND0_DOT29, BK_ROUND_QUAD
int batch_launch_pipe(const BatchRecorder &r, size_t idx);     // pipe.hip:   BK_PIPE, BK_FINISH_PIPE
template <class Slot>
inline void batch_gather(const BatchRecorder &r, size_t idx, BatchOf<Slot> &out) {
    std::memset(&out, 0, sizeof out);
    for (int b = 0; b < r.n; ++b) std::memcpy(&out.a[b], r.recs[b][idx].slot, sizeof(Slot));
}

struct RoundLaunchCtx {
    hipStream_t stream;
    const FieldParams *P;
    uint64_t *d_partials;          // block sums: capacity_elems field elements
    uint64_t capacity_elems;
    ClaimJob claim;                // out != null: a SKIP1 kernel, if one is launched, also evaluates the tail's claim (one extra workgroup)
};
enum { kLaunchOk = 0, kLaunchUnsupported = -1, kLaunchHipError = -2 };
// One round: (fold at *d_r when fused +) sums for t = 0..D over q pairs -> per-block partials; *out_grid blocks.
// skip1 (optional, in/out): request the variant that leaves out the t = 1 sums (S(1) is derived by the tail); set to false
// when the shape has no such variant and the full kernel was launched instead.
// lead (optional, in/out): request the variant whose slot D accumulates the leading coefficient instead of S(D) (k_round_kd
// LEAD; the tail rebuilds S(D)); set to false when the launched kernel has no such variant.
int launch_round(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, uint32_t D, bool fused,
                 const uint64_t *d_r, uint32_t *out_grid, bool *skip1 = nullptr, bool *lead = nullptr);
// Terms {k, 1} of a sum of products in one pass: fp = the k factors of the product, then the single-factor term
// ((k, D) = (2, 2) or (3, 3); anything else returns kLaunchUnsupported and the caller launches term by term).
int launch_round_plus1(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, uint32_t D, bool fused,
                       const uint64_t *d_r, uint32_t *out_grid, bool *skip1 = nullptr, bool *lead = nullptr);
// One evaluation point (any degree): sums of prod_f (lo - t*(lo-hi)) -> per-block partials (1 sum per block).
int launch_round_single_t(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, const Fe &tval, uint32_t *out_grid);


// ---- pipelined rounds (pipe_kernels.cuh / pipe.hip): sums of round s as a polynomial in the pending challenge -------------
constexpr uint32_t kPipeMaxWorkBlocks = 256;   // k_round_pipe (hex rows)
}  // namespace zk
#include "pipe_args.hpp"
namespace zk {
struct PipeLaunch {
    int k, extra;             // shape: k-factor product (+ one single-factor term)
    uint32_t D;
    bool fold;                // work blocks first fold fp.in (8q elements) -> fp.out (4q) at *chal_fold
    int emit;                 // 1: write the E partials of the round with q pairs; 0: fold only (leaving the pipeline)
    uint64_t q;               // pairs of the round whose E is prepared
    const uint64_t *chal_fold;
    uint64_t *e_partials;     // out: slot 0 = total, slot 1 + b = work block b; (D+1)*(k+1) elements each
    uint32_t *done_counter;   // device word, zero between launches (last-block-done reduction of the partials)
    PipeTailArgs tail;        // the transcript block's job (the round before)
};
bool pipe_shape_ok(int k, uint32_t D, int extra);
uint32_t pipe_values_per_block(int k, uint32_t D);
uint32_t pipe_rows_per_block(int k, uint32_t D, int extra);
uint32_t pipe_work_blocks(int k, uint32_t D, int extra, uint64_t q);
int launch_round_pipe(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t *out_work_blocks);
// ZK_SHARD_FAKE_ALLREDUCE_US: `us` microseconds of one spinning wave on `stream`
int launch_spin_us(hipStream_t stream, uint32_t us);
// the pipelined finisher (k_finish_pipe): every remaining round in one launch; entry = kFinEntry* of pipe_kernels.cuh
// (0 fresh tables, 1 tables with *chal_in pending, 2 the same with the E partials of the first round ready)
struct FinishPipeLaunch {
    int k, extra;
    uint32_t D;
    uint32_t m_in;            // variables of the tables in fp.in (3 .. finish_pipe_max_vars(k + extra) + 1)
    int entry;
    const uint64_t *e_partials;
    uint32_t e_blocks;
    PipeConsts pc;
    const uint64_t *chal_in;
    uint64_t *chal_last;      // record the last challenge is published in
    WordSponge *sponge;
    uint64_t *out_rp, *out_ch, *out_final;
    uint64_t *dbg;            // optional timestamps (ZK_PIPE_DEBUG)
    FinishPublish pub;        // flag != null: this launch ends the call -- publish the proof block and the completion word
};
int launch_finish_pipe(const RoundLaunchCtx &lc, const FactorPtrs &fp, const FinishPipeLaunch &fl);
uint32_t finish_pipe_max_vars(int n_factors);   // largest after-fold table (variables) the pipelined finisher keeps in LDS

}  // namespace zk
