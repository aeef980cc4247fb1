// zeta_kernels.cuh -- CoeffMultilinearPolynomial::to_evaluation_form (coefficient_form.rs:340-347) as LDS-tiled passes.
//
// eval[idx] = sum of the coefficients whose key is a subset of the point's variable set.  With key bit v <-> variable v and
// table index bit (n-1-v) <-> variable v that is the zeta (subset-sum) transform of the coefficient vector placed at
// bit-reversed positions: one butterfly level per index bit, T[x | b] += T[x] -- additions only, so the transform is bound by
// HBM and what matters is how many times the table crosses it.  Until round 5 it crossed eight times at 2^24 (three index
// bits per launch, k_zeta_multi<3> in kernels.cuh, kept for reference timing); here:
//
//   pass 1  (k_zeta_first):  a workgroup owns 2^11 CONSECUTIVE table entries (64 KiB) and all 11 low index bits.  It never
//            reads the table: its LDS tile starts at zero, the terms that fall into it are scattered in (the host sorts the
//            terms by table index, the workgroup finds its range by binary search), 11 butterfly levels run in LDS, the tile
//            is written once.  No memset, no scatter kernel.  A tile without terms is written as zeros straight away.
//   pass 2+ (k_zeta_tile):   the remaining index bits, at most 8 per pass: a tile is 2^L rows (the L bits of the pass, stride
//            2^pos) by C = 2^(11-L) consecutive entries (runs of 32 C bytes: 512 B at L = 7, 1 KiB at L = 6), read once and
//            written once, in place (tiles are disjoint and a tile is complete in LDS before anything is stored).
//
//   2^24: 11 + 7 + 6 bits = 0.5 + 1 + 1 GiB of traffic instead of 8 GiB.
//
// Inside a tile the levels run in register groups of up to three bits (a thread holds the 2^G entries that differ in the
// group's bits: G levels of additions without touching LDS), the tile crossing LDS between groups.  LDS layout: two planes
// of 16-byte halves (low / high 128 bits of an entry), entry i of the tile in slot i ^ ((i >> 3) & 15) of each plane: with
// that swizzle every ds_read_b128 / ds_write_b128 lane group of every access pattern used here (consecutive entries,
// stride-8 entries of the lowest group, the transfers to and from HBM) hits distinct banks -- checked by enumeration over
// the lane groups MI355X_MICROARCH.md lists (the one exception, a two-bit group at bit 3, is two-way on reads).  Global
// accesses are one dwordx4 per lane over consecutive 16-byte pieces of a run (the fold's data movement, kernels.cuh).
#pragma once
#include "common.cuh"

namespace zk {

constexpr uint32_t kZetaTileLog = 11;
constexpr uint32_t kZetaTile = 1u << kZetaTileLog;                 // entries per tile: 64 KiB
constexpr uint32_t kZetaPlaneBytes = kZetaTile * 16 + 64;          // the high plane starts 64 B off a 128-B boundary: the lane pairs of the
                                                                   // HBM -> LDS transfer (low half, high half of one entry) write distinct banks
constexpr uint32_t kZetaLdsBytes = 2 * kZetaPlaneBytes;            // 131,200 B for two workgroups: two per CU

ZK_D uint32_t zeta_slot(uint32_t i) { return i ^ ((i >> 3) & 15u); }

// the G levels of bits [s, s + G) of the tile-local index, for every entry of a tile of 2^tile_log entries
template <int G>
ZK_D void zeta_group(unsigned char *smem, uint32_t s, uint32_t tile_log, const FieldParams &P) {
    uint4 *plo = reinterpret_cast<uint4 *>(smem), *phi = reinterpret_cast<uint4 *>(smem + kZetaPlaneBytes);
    const uint32_t items = 1u << (tile_log - G);
    for (uint32_t w = threadIdx.x; w < items; w += kBlock) {
        const uint32_t low = w & ((1u << s) - 1u), high = w >> s;
        const uint32_t i0 = (high << (s + G)) | low;
        Fe x[1 << G];
#pragma unroll
        for (int u = 0; u < (1 << G); ++u) {
            const uint32_t sl = zeta_slot(i0 | ((uint32_t)u << s));
            const uint4 a = plo[sl], b = phi[sl];
            x[u] = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
        }
#pragma unroll
        for (int b = 0; b < G; ++b)
#pragma unroll
            for (int c = 0; c < (1 << G); ++c)
                if (c & (1 << b)) x[c] = fe_add(x[c], x[c ^ (1 << b)], P);
#pragma unroll
        for (int u = 1; u < (1 << G); ++u) {   // entry 0 of a group never changes
            const uint32_t sl = zeta_slot(i0 | ((uint32_t)u << s));
            plo[sl] = make_uint4(x[u].v[0], x[u].v[1], x[u].v[2], x[u].v[3]);
            phi[sl] = make_uint4(x[u].v[4], x[u].v[5], x[u].v[6], x[u].v[7]);
        }
    }
}
// levels of bits [lb, lb + L) of the tile-local index; ends with the tile complete in LDS (barrier included)
ZK_D void zeta_levels(unsigned char *smem, uint32_t lb, uint32_t L, uint32_t tile_log, const FieldParams &P) {
    for (uint32_t s = lb; s < lb + L;) {
        const uint32_t g = lb + L - s >= 3 ? 3u : lb + L - s;
        __syncthreads();
        if (g == 3) zeta_group<3>(smem, s, tile_log, P);
        else if (g == 2) zeta_group<2>(smem, s, tile_log, P);
        else zeta_group<1>(smem, s, tile_log, P);
        s += g;
    }
    __syncthreads();
}

// pass 1: the 2^tile_log consecutive entries [tile * 2^tile_log, ...) from the sorted term list, all tile_log low index bits
// (tile_log = min(n_vars, 11)).  idx: table indices of the terms, ascending; coeffs: their coefficients -- in the same order (perm == null)
// or, for a list ordered on the device (zeta_sort.hip), in the CALLER's order with perm[t] = position of sorted term t.  Equal indices
// may repeat: a run is added up by the thread of its first term (BTreeMap semantics: duplicate terms are summed, coefficient_form.rs:164-171;
// the host path merges them beforehand, so its lists are unique).
__global__ __launch_bounds__(kBlock) void k_zeta_first(uint64_t *__restrict__ table, const uint64_t *__restrict__ idx,
                                                       const uint64_t *__restrict__ coeffs, uint64_t n_terms, uint32_t tile_log,
                                                       FieldParams P, const uint32_t *__restrict__ perm = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint64_t range[2];
    uint4 *plo = reinterpret_cast<uint4 *>(smem), *phi = reinterpret_cast<uint4 *>(smem + kZetaPlaneBytes);
    const uint32_t tid = threadIdx.x, entries = 1u << tile_log;
    const uint64_t first = (uint64_t)blockIdx.x << tile_log;
    if (tid < 2) {   // lower_bound of the tile's first index (tid 0) and of the next tile's (tid 1)
        const uint64_t key = first + ((uint64_t)tid << tile_log);
        uint64_t lo = 0, hi = n_terms;
        while (lo < hi) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (idx[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        range[tid] = lo;
    }
    __syncthreads();
    const uint64_t t0 = range[0], t1 = range[1];
    uint4 *g4 = reinterpret_cast<uint4 *>(table) + 2 * first;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (t0 == t1) {   // no term in this tile (workgroup-uniform): zeros, no LDS
        for (uint32_t q = tid; q < 2 * entries; q += kBlock) g4[q] = zero;
        return;
    }
    for (uint32_t i = tid; i < entries; i += kBlock) {
        plo[i] = zero;
        phi[i] = zero;
    }
    __syncthreads();
    for (uint64_t t = t0 + tid; t < t1; t += kBlock) {
        const uint64_t my = idx[t];
        if (t > t0 && idx[t - 1] == my) continue;   // not the first of its run: the run's first thread adds it
        const uint32_t sl = zeta_slot((uint32_t)(my - first));
        Fe v = fe_load(coeffs, perm ? perm[t] : t);
        for (uint64_t u = t + 1; u < t1 && idx[u] == my; ++u) v = fe_add(v, fe_load(coeffs, perm ? perm[u] : u), P);
        plo[sl] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
        phi[sl] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
    }
    zeta_levels(smem, 0, tile_log, tile_log, P);
    for (uint32_t q = tid; q < 2 * entries; q += kBlock)   // piece q of the run: half (q & 1) of entry q >> 1
        g4[q] = *reinterpret_cast<const uint4 *>(smem + (q & 1) * kZetaPlaneBytes + zeta_slot(q >> 1) * 16);
}

// passes 2+: index bits [pos, pos + L), 1 <= L <= 8, pos >= 11.  Tile = 2^L rows x C = 2^(11-L) consecutive entries; tile-local
// index i = (row << log2 C) | col.  One workgroup per tile, gridDim.x = 2^(n_vars - 11).
__global__ __launch_bounds__(kBlock) void k_zeta_tile(uint64_t *table, uint32_t pos, uint32_t L, FieldParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, log_c = kZetaTileLog - L;
    const uint64_t outer = (uint64_t)blockIdx.x >> (pos - log_c), cg = (uint64_t)blockIdx.x & ((1ull << (pos - log_c)) - 1);
    const uint64_t base = (outer << (pos + L)) + (cg << log_c);
    uint4 *g4 = reinterpret_cast<uint4 *>(table);
    const uint32_t pieces_log = log_c + 1;   // 16-byte pieces per run
#pragma unroll 4
    for (uint32_t q = tid; q < 2 * kZetaTile; q += kBlock) {
        const uint32_t row = q >> pieces_log, piece = q & ((1u << pieces_log) - 1u);
        const uint4 v = g4[2 * (base + ((uint64_t)row << pos)) + piece];
        const uint32_t i = (row << log_c) | (piece >> 1);
        *reinterpret_cast<uint4 *>(smem + (piece & 1) * kZetaPlaneBytes + zeta_slot(i) * 16) = v;
    }
    zeta_levels(smem, log_c, L, kZetaTileLog, P);
#pragma unroll 4
    for (uint32_t q = tid; q < 2 * kZetaTile; q += kBlock) {
        const uint32_t row = q >> pieces_log, piece = q & ((1u << pieces_log) - 1u);
        if (row == 0) continue;   // row 0 of a tile never changes
        const uint32_t i = (row << log_c) | (piece >> 1);
        g4[2 * (base + ((uint64_t)row << pos)) + piece] =
            *reinterpret_cast<const uint4 *>(smem + (piece & 1) * kZetaPlaneBytes + zeta_slot(i) * 16);
    }
}

}  // namespace zk
