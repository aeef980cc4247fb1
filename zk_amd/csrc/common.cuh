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

}  // namespace zk
