// launch.hpp -- host-side launcher of the round kernels (defined in rounds.hip, used by capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.cuh"

namespace zk {

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
constexpr uint32_t kMidMaxWorkBlocks = 512;    // k_round_mid (quads): also the capacity of an E-partial buffer
}  // namespace zk
#include "pipe_args.hpp"
namespace zk {
struct PipeLaunch {
    int k, extra;             // shape: k-factor product (+ one single-factor term)
    uint32_t D;
    bool fold;                // work blocks first fold fp.in (8q elements) -> fp.out (4q) at *chal_fold
    bool mid;                 // k_round_mid (four lanes per pair index) instead of k_round_pipe's sixteen-lane rows
    bool mid_total;           // ... whose last block adds the block partials up (slot 0); false: the next launch's transcript block does
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
uint64_t mid_max_pairs();   // pair indices one k_round_mid launch can take (grid cap x unreduced products per lane)
int launch_round_pipe(const RoundLaunchCtx &lc, const FactorPtrs &fp, const PipeLaunch &pl, uint32_t *out_work_blocks);
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
