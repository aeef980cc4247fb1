// quad.cuh -- DPP helpers for kernels that give the four lanes of a quad one pair index (k_round_quad, the pipelined rounds).
#pragma once
#include "common.cuh"

namespace zk {

template <int SRC>
ZK_D Fe quad_bcast(const Fe &x) {   // every lane of a quad reads lane SRC of the quad
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = __builtin_amdgcn_mov_dpp(x.v[i], SRC | (SRC << 2) | (SRC << 4) | (SRC << 6), 0xF, 0xF, true);
    return o;
}
template <int G, int NF, int NS>
ZK_D void quad_transpose(const Fe (&v)[NS], Fe (&w)[NF], uint32_t l4) {   // w[g] of lane t <- v[t] of lane g
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const Fe b = quad_bcast<G>(v[t]);
        if (l4 == (uint32_t)t) w[G] = b;
    }
    if constexpr (G + 1 < NF) quad_transpose<G + 1, NF, NS>(v, w, l4);
}
}  // namespace zk
