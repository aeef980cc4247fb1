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
#if defined(__HIPCC__)
ZK_D Mul29 load_challenge29(const uint64_t *rptr) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(rptr) + 8;
    Mul29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = __builtin_amdgcn_readfirstlane(w[i]);   // wave-uniform -> SGPRs
    return r;
}
#endif

}  // namespace zk
