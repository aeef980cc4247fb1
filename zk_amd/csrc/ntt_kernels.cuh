// ntt_kernels.cuh -- LDS-staged radix-2 NTT for gfx950 (fft crate: fft/src/lib.rs:4-46 computes the same DFT
// out[i] = sum_j in[j] * omega^(i*j), natural order in and out, by recursion with a `pow` per butterfly).
//
// Decomposition (decimation in time, "four-step" generalised to P passes): N = R_1 * R_2 * ... * R_P, R_p = 2^(l_p),
// 4 <= l_p <= 8.  With O_p = R_1..R_(p-1) (digits already transformed) and I_p = R_(p+1)..R_P (digits still in time
// order) the data is viewed as [O_p][R_p][I_p]:
//   pass p < P : for every (o, i): R_p-point DFT along the middle axis (stride I_p), then multiply element (k, i) by
//                the inter-pass twiddle omega_N^(O_p * i * k)  (= omega_M^(i*k), M = R_p * I_p); in-place addresses.
//   pass P     : R_P-point DFT along the contiguous axis; the result digit string (k_1, .., k_P) is written to
//                k_1 + R_1*(k_2 + R_2*(k_3 + ...)) -- the transposition that makes the output natural order.
// One workgroup transforms a tile of R x 16 elements held in LDS: the 16 "columns" are 16 consecutive values of the
// contiguous index (512 B runs in HBM), so every global access is a 512 B - 2 KiB run, and each pass moves the vector
// through HBM exactly once: P = 3 passes for 2^24 (the algorithmic 2*N*32 B is one pass; attainable fraction <= 1/3).
//
// LDS layout: two planes of 16-byte halves (low / high 128 bits of each element), row r = element index along the DFT
// axis, 16 column slots of 16 B per row, rows padded to 272 B.  A 16-lane group reading one row touches 64 distinct
// banks (conflict-free ds_read_b128); lanes walking down a column advance 4 banks per row (conflict-free as well),
// which is what the transposing load of the last pass does.
//
// The butterflies are ALU-bound on 256-bit modular multiplication (one per butterfly + one per inter-pass twiddle),
// not HBM-bound: see DESIGN.md for both rooflines.
#pragma once
#include "common.cuh"

namespace zk {

constexpr int kNttCols = 16;          // tile columns (consecutive contiguous-axis indices)
constexpr int kNttRowBytes = 272;     // 16 slots * 16 B + 16 B pad
constexpr int kNttThreads = 512;
constexpr int kNttMaxLog = 8;         // R <= 256

struct NttPlan {
    uint32_t log_n;
    uint32_t n_pass;
    uint32_t l[4];          // log2 radix of each pass
    uint32_t lo_bits;       // two-level twiddle table: w_lo[i] = omega^i (i < 2^lo_bits), w_hi[i] = omega^(i << lo_bits)
    const uint64_t *w_lo;
    const uint64_t *w_hi;
};

// omega^e from the two-level table (e < n)
ZK_D Fe ntt_twiddle(const NttPlan &pl, uint64_t e, const FieldParams &P) {
    const uint64_t lo = e & ((1ull << pl.lo_bits) - 1), hi = e >> pl.lo_bits;
    if (hi == 0) return fe_load(pl.w_lo, lo);
    if (lo == 0) return fe_load(pl.w_hi, hi);
    return fe_mul(fe_load(pl.w_hi, hi), fe_load(pl.w_lo, lo), P);
}

ZK_D uint32_t lds_off(uint32_t row, uint32_t col) { return row * kNttRowBytes + col * 16; }
ZK_D Fe lds_get(const unsigned char *lo_plane, const unsigned char *hi_plane, uint32_t row, uint32_t col) {
    const uint4 a = *reinterpret_cast<const uint4 *>(lo_plane + lds_off(row, col));
    const uint4 b = *reinterpret_cast<const uint4 *>(hi_plane + lds_off(row, col));
    Fe r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
ZK_D void lds_put(unsigned char *lo_plane, unsigned char *hi_plane, uint32_t row, uint32_t col, const Fe &v) {
    *reinterpret_cast<uint4 *>(lo_plane + lds_off(row, col)) = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    *reinterpret_cast<uint4 *>(hi_plane + lds_off(row, col)) = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
}

// LAST = false: pass p < P (strided axis, inter-pass twiddle, same addresses in and out)
// LAST = true : pass P (contiguous axis, transposing store, optional scaling by n^-1 for the inverse transform)
template <bool LAST>
__global__ __launch_bounds__(kNttThreads) void k_ntt_pass(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                          NttPlan pl, uint32_t pass, FieldParams P, Fe scale, int do_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t l = pl.l[pass], R = 1u << l;
    unsigned char *lo_plane = smem;
    unsigned char *hi_plane = smem + (size_t)R * kNttRowBytes;
    uint64_t *tws = reinterpret_cast<uint64_t *>(smem + 2 * (size_t)R * kNttRowBytes);   // omega_R^j, j < R/2

    uint32_t lo_sum = 0;   // log2 O_p
    for (uint32_t p = 0; p < pass; ++p) lo_sum += pl.l[p];
    const uint32_t log_inner = pl.log_n - lo_sum - l;           // log2 I_p
    const uint64_t inner = 1ull << log_inner;
    const uint32_t tid = threadIdx.x;

    // sub-DFT twiddles omega_R^j = omega_N^(j * N/R)
    for (uint32_t j = tid; j < R / 2; j += kNttThreads) fe_store(tws, j, ntt_twiddle(pl, (uint64_t)j << (pl.log_n - l), P));

    // ---- tile coordinates ----
    uint64_t base_in = 0, base_out = 0, tw_i0 = 0;
    uint64_t out_stride_a = 0;
    if (!LAST) {
        const uint64_t tiles_per_outer = inner / kNttCols;
        const uint64_t o = blockIdx.x / tiles_per_outer, i0 = (blockIdx.x % tiles_per_outer) * kNttCols;
        base_in = base_out = ((o << l) << log_inner) + i0;        // + a * inner + t
        tw_i0 = i0;
    } else {
        // outer index o = (k_1, rest) with k_1 the most significant digit; tile = 16 consecutive k_1 at fixed rest
        const uint32_t l1 = pl.n_pass > 1 ? pl.l[0] : 0;
        const uint32_t log_q = lo_sum - l1;                         // log2 of the `rest` range
        const uint64_t rest = blockIdx.x & ((1ull << log_q) - 1), k1_0 = (blockIdx.x >> log_q) * kNttCols;
        base_in = ((k1_0 << log_q) + rest) << l;                    // + (t << (log_q + l)) + a
        // output digit reversal of `rest` = (k_2, .., k_(P-1)), k_2 most significant: sum_j k_j * (R_1 .. R_(j-1))
        uint64_t rev = 0, weight = 1ull << l1;
        uint32_t shift = log_q;
        for (uint32_t p = 1; p + 1 < pl.n_pass; ++p) {
            shift -= pl.l[p];
            const uint64_t digit = (rest >> shift) & ((1ull << pl.l[p]) - 1);
            rev += digit * weight;
            weight <<= pl.l[p];
        }
        base_out = k1_0 + rev;                                      // + t + a' * O_P
        out_stride_a = 1ull << lo_sum;
    }

    // ---- load tile into LDS in natural row order (the stages below are decimation in frequency) ----
    if (!LAST) {
        for (uint32_t e = tid; e < R * kNttCols; e += kNttThreads) {
            const uint32_t t = e & (kNttCols - 1), a = e >> 4;
            const Fe v = fe_load(in, base_in + ((uint64_t)a << log_inner) + t);
            lds_put(lo_plane, hi_plane, a, t, v);
        }
    } else {
        const uint32_t l1 = pl.n_pass > 1 ? pl.l[0] : 0;
        const uint32_t t_shift = (lo_sum - l1) + l;
        for (uint32_t e = tid; e < R * kNttCols; e += kNttThreads) {
            const uint32_t a = e & (R - 1), t = e >> l;
            const Fe v = fe_load(in, base_in + ((uint64_t)t << t_shift) + a);
            lds_put(lo_plane, hi_plane, a, t, v);
        }
    }
    __syncthreads();

    // ---- l radix-2 DIF stages, h = R/2 .. 1: (x_j, x_(j+h)) <- (x_j + x_(j+h), (x_j - x_(j+h)) * omega_(2h)^j) ----
    // natural-order rows in, bit-reversed rows out: frequency k ends up in row bitrev_l(k).  16 lanes share a butterfly
    // row pair, so every LDS access is one conflict-free 256-B row segment per 16-lane group.
    for (int s = (int)l - 1; s >= 0; --s) {
        const uint32_t h = 1u << s;
        for (uint32_t e = tid; e < (R / 2) * kNttCols; e += kNttThreads) {
            const uint32_t t = e & (kNttCols - 1), b = e >> 4;
            const uint32_t j = b & (h - 1), r0 = ((b >> s) << (s + 1)) + j, r1 = r0 + h;
            const Fe u = lds_get(lo_plane, hi_plane, r0, t);
            const Fe v = lds_get(lo_plane, hi_plane, r1, t);
            Fe d = fe_sub(u, v, P);
            if (s) d = fe_mul(d, fe_load(tws, (uint64_t)j << (l - 1 - s)), P);   // the h = 1 stage has unit twiddles
            lds_put(lo_plane, hi_plane, r0, t, fe_add(u, v, P));
            lds_put(lo_plane, hi_plane, r1, t, d);
        }
        __syncthreads();
    }

    // ---- store ----
    if (!LAST) {
        const uint64_t tw_scale_log = lo_sum;   // exponent = O_p * i * k
        for (uint32_t e = tid; e < R * kNttCols; e += kNttThreads) {
            const uint32_t t = e & (kNttCols - 1), k = e >> 4;
            Fe v = lds_get(lo_plane, hi_plane, __brev(k) >> (32 - l), t);
            if (k) v = fe_mul(v, ntt_twiddle(pl, ((tw_i0 + t) * k) << tw_scale_log, P), P);
            fe_store(out, base_out + ((uint64_t)k << log_inner) + t, v);
        }
    } else {
        for (uint32_t e = tid; e < R * kNttCols; e += kNttThreads) {
            const uint32_t t = e & (kNttCols - 1), k = e >> 4;
            Fe v = lds_get(lo_plane, hi_plane, __brev(k) >> (32 - l), t);
            if (do_scale) v = fe_mul(v, scale, P);
            fe_store(out, base_out + t + (uint64_t)k * out_stride_a, v);
        }
    }
}

// two-level twiddle table: w_lo[i] = omega^i for i < 2^lo_bits; w_hi[i] = (omega^(2^lo_bits))^i for i < 2^hi_bits
__global__ __launch_bounds__(kBlock) void k_ntt_tables(uint64_t *__restrict__ w_lo, uint64_t *__restrict__ w_hi, uint32_t lo_bits,
                                                       uint32_t hi_bits, Fe omega, FieldParams P) {
    const uint64_t n_lo = 1ull << lo_bits, n_hi = 1ull << hi_bits, total = n_lo + n_hi;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    Fe omega_hi = omega;   // omega^(2^lo_bits)
    for (uint32_t i = 0; i < lo_bits; ++i) omega_hi = fe_sqr(omega_hi, P);
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += stride) {
        const bool is_hi = idx >= n_lo;
        uint64_t e = is_hi ? idx - n_lo : idx;
        Fe base = is_hi ? omega_hi : omega, acc = fe_one(P);
        while (e) {
            if (e & 1) acc = fe_mul(acc, base, P);
            base = fe_sqr(base, P);
            e >>= 1;
        }
        fe_store(is_hi ? w_hi : w_lo, is_hi ? idx - n_lo : idx, acc);
    }
}

}  // namespace zk
