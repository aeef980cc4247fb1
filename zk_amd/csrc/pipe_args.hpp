// pipe_args.hpp -- argument block of the transcript block of a pipelined round kernel (shared by host and device code).
#pragma once
#include "common.cuh"
#include "keccak.hpp"

namespace zk {

// Constants of the transcript wave's evaluation (pipe_kernels.cuh pipe_eval_canon), prepared once per context on the host.
struct PipeConsts {
    Mul29 inv2p;   // prepared multiplier of 1/2 in Montgomery form (the K = 3 nodes 1 and -1)
    Mul29 k266;    // limbs of 2^266 mod p: fe_mul29(x, k266) = x * 2^5 mod p, whose 29-bit split is the prepared CANONICAL challenge
};

// The finisher as the LAST launch of a call: wave 0 copies the finished proof block into pinned host memory and stores the completion
// word behind a system-scope fence (what k_publish_host does as a launch of its own, ~4 us + a launch boundary).  flag == null: no.
struct FinishPublish {
    const uint64_t *src;        // device: [round polys | challenges | factor values], n_u64 words (even)
    uint64_t *dst_host;         // pinned, mapped
    uint32_t n_u64;
    volatile uint32_t *flag;    // pinned, coherent
    uint32_t seq;
};

struct PipeTailArgs {
    const uint64_t *partials;   // per-block partials of the round being closed: [block][n_in] elements
    uint32_t nblocks;
    uint32_t n_in;              // values per block: NS (mode 0; SKIP1 leaves slot 1 unwritten) or NS * NR (mode 1)
    int mode;                   // 0: plain sums   1: E(t; rho), evaluated at *chal_in
    const uint64_t *chal_in;    // challenge record of the previous round (mode 1, and SKIP1's derive)
    uint64_t *chal_out;         // challenge record this round's challenge is published in
    WordSponge *sponge;
    uint64_t *out_rp, *out_ch;
    PipeConsts pc;
    TailDerive dv;              // mode 0 after a SKIP1 round kernel
    uint64_t *dbg;              // optional: 100 MHz timestamps of the phases (ZK_PIPE_DEBUG), 32 slots per launch
};

}  // namespace zk
