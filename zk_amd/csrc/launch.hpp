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
};
enum { kLaunchOk = 0, kLaunchUnsupported = -1, kLaunchHipError = -2 };
// One round: (fold at *d_r when fused +) sums for t = 0..D over q pairs -> per-block partials; *out_grid blocks.
// skip1 (optional, in/out): request the variant that leaves out the t = 1 sums (S(1) is derived by the tail); set to false
// when the shape has no such variant and the full kernel was launched instead.
int launch_round(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, uint32_t D, bool fused,
                 const uint64_t *d_r, uint32_t *out_grid, bool *skip1 = nullptr);
// Terms {k, 1} of a sum of products in one pass: fp = the k factors of the product, then the single-factor term
// ((k, D) = (2, 2) or (3, 3); anything else returns kLaunchUnsupported and the caller launches term by term).
int launch_round_plus1(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, uint32_t D, bool fused,
                       const uint64_t *d_r, uint32_t *out_grid, bool *skip1 = nullptr);
// One evaluation point (any degree): sums of prod_f (lo - t*(lo-hi)) -> per-block partials (1 sum per block).
int launch_round_single_t(const RoundLaunchCtx &lc, const FactorPtrs &fp, int k, uint64_t q, const Fe &tval, uint32_t *out_grid);

}  // namespace zk
