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

constexpr int kNttColsLog = 3;
constexpr int kNttCols = 1 << kNttColsLog;   // tile columns (consecutive contiguous-axis indices)
constexpr int kNttRowBytes = kNttCols * 16 + 16;   // one 16-B slot per column + 16 B pad; ONE plane of 256 rows = 36 KiB (+ 8 KiB of twiddles): three workgroups per CU
constexpr int kNttThreads = 32 * kNttCols;
constexpr int kNttMaxLog = 8;         // R <= 256

// Twiddles are stored ready for the carry-free multiplier (field.cuh fe_mul29): every multiplication of the transform
// has a table value as one operand, so the tables hold omega^e * 2^5 mod p -- the low table already split into nine
// 29-bit limbs (Mul29, 64-byte records), the high table as a plain element so that two levels compose with one more
// fe_mul29:  hi' (x) lo' = (w_hi 2^5)(w_lo 2^5) 2^-261 = (w_hi w_lo) 2^5, again a prepared value.
constexpr int kTw29Words = 16;   // record stride of a stored Mul29 (9 words used)
struct NttPlan {
    uint32_t log_n;
    uint32_t n_pass;
    uint32_t l[4];          // log2 radix of each pass
    uint32_t lo_bits;       // two-level table: w_lo[i] ~ omega^i (i < 2^lo_bits), w_hi[i] ~ omega^(i << lo_bits)
    const uint32_t *w_lo;   // Mul29 records of omega^i
    const uint64_t *w_hi;   // elements omega^(i << lo_bits) * 2^5 mod p
    // optional full inter-pass twiddle table of pass p < P: element (k, i) at [k * I_p + i] = omega^(O_p * i * k) * 2^5 mod p,
    // R_p * I_p = n / O_p entries (the whole vector for pass 0, n / R_1 for pass 1, ...).  Trades the compose multiply
    // for a 32-byte read in a pass that is bound by VALU issue, not HBM.  Null: compose from the two-level table.
    const uint64_t *w_full[4];
};
ZK_D Mul29 load_mul29(const uint32_t *rec) {
    const uint4 a = *reinterpret_cast<const uint4 *>(rec), b = *reinterpret_cast<const uint4 *>(rec + 4);
    Mul29 m = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, rec[8]}};
    return m;
}
ZK_D void store_mul29(uint32_t *rec, const Mul29 &m) {
    *reinterpret_cast<uint4 *>(rec) = make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]);
    *reinterpret_cast<uint4 *>(rec + 4) = make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]);
    rec[8] = m.l[8];
}
// prepared multiplier for omega^e (e < n)
ZK_D Mul29 ntt_twiddle(const NttPlan &pl, uint64_t e, const FieldParams &P) {
    const uint64_t lo = e & ((1ull << pl.lo_bits) - 1), hi = e >> pl.lo_bits;
    if (hi == 0) return load_mul29(pl.w_lo + lo * kTw29Words);
    const Fe h = fe_load(pl.w_hi, hi);
    Mul29 m;
    if (lo == 0) {
        split29(h.v, m.l);
        return m;
    }
    const Fe c = fe_mul29(h, load_mul29(pl.w_lo + lo * kTw29Words), P);
    split29(c.v, m.l);
    return m;
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

// Half-element forms: the tile crosses LDS one 16-byte half at a time (low 128 bits of every element, then the high ones)
// through ONE plane, so a workgroup holds 36 KiB + twiddles instead of 72 KiB and three workgroups share a CU (the passes
// are bound by VALU issue: rocprofv3 r03 shows 2 waves per SIMD parked 23 % and issue-stalled 29 % of their cycles).
ZK_D void lds_put_half(unsigned char *plane, uint32_t row, uint32_t col, const Fe &v, int half) {
    *reinterpret_cast<uint4 *>(plane + lds_off(row, col)) =
        half ? make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]) : make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
}
ZK_D void lds_get_half(const unsigned char *plane, uint32_t row, uint32_t col, Fe &v, int half) {
    const uint4 a = *reinterpret_cast<const uint4 *>(plane + lds_off(row, col));
    if (half) {
        v.v[4] = a.x; v.v[5] = a.y; v.v[6] = a.z; v.v[7] = a.w;
    } else {
        v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w;
    }
}

// ---- lazy arithmetic inside a transform ------------------------------------------------------------------------------------
// Values stay in [0, 2p) from the load of the first pass to the store of the last one (2p < 2^256 for every supported field):
// add / sub reduce modulo 2p (same instruction count as modulo p), the multiplier skips its final conditional subtraction
// (fe_mul29_t<true>: 16 instructions per butterfly), and the last pass subtracts p once.  Same values modulo p, so the canonical
// outputs are bit-identical.
struct Mod2p {
    uint32_t p[8];
};
ZK_D Mod2p mod2p_of(const FieldParams &P) {
    Mod2p m;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        m.p[i] = (P.p[i] << 1) | c;
        c = P.p[i] >> 31;
    }
    return m;
}
ZK_D Fe fe_add2(const Fe &a, const Fe &b, const Mod2p &M) {
    Fe s, d, r;
    const uint32_t carry = add8(s.v, a.v, b.v);
    const uint32_t borrow = sub8(d.v, s.v, M.p);
    const bool use_d = carry | (borrow ^ 1u);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = use_d ? d.v[i] : s.v[i];
    return r;
}
ZK_D Fe fe_sub2(const Fe &a, const Fe &b, const Mod2p &M) {
    Fe d, e, r;
    const uint32_t borrow = sub8(d.v, a.v, b.v);
    add8(e.v, d.v, M.p);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = borrow ? e.v[i] : d.v[i];
    return r;
}
ZK_D Fe fe_canon2(const Fe &a, const FieldParams &P) {   // [0, 2p) -> [0, p)
    Fe d, r;
    const uint32_t borrow = sub8(d.v, a.v, P.p);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}

// ---- stage groups -----------------------------------------------------------------------------------------------------
// The l radix-2 DIF stages of a tile are run in groups of up to 3 stages: a thread holds the 2^g rows (one column) that
// interact inside the group in REGISTERS and does the g stages there; tiles cross LDS only between groups (l = 8: groups
// 3+3+2 -> two exchanges instead of eight), the first group is fed straight from HBM and the last one stores straight to
// HBM.  Rows of a group covering stages s_hi..s_lo: r = (pre << (s_hi+1)) | (u << s_lo) | post, u = 0..2^g-1.
// DIF butterfly at stage s (half h = 2^s): (x_r, x_(r+h)) <- (x_r + x_(r+h), (x_r - x_(r+h)) * omega_R^((r mod h) * R/(2h))).
template <int L>
struct NttGroups {   // group sizes from the top stage down
    static constexpr int n = L <= 3 ? 1 : (L <= 6 ? 2 : 3);
    static constexpr int g0 = (L + n - 1) / n;
    static constexpr int g1 = n > 1 ? (L - g0 + (n - 1) - 1) / (n - 1) : 0;
    static constexpr int g2 = n > 2 ? L - g0 - g1 : 0;
};
// compile-time recursion over (stage within the group SU, butterfly B): every x[] index is a constant, so the group
// stays in registers (a runtime-indexed array would go to scratch)
template <int L, int S_LO, int G, int SU, int B>
ZK_D void ntt_bfly(Fe (&x)[1 << G], uint32_t post, const uint32_t *tws, const FieldParams &P, const Mod2p &M2) {
    constexpr uint32_t hu = 1u << SU;
    constexpr uint32_t ua = (uint32_t)((B >> SU) << (SU + 1)) | (B & (hu - 1)), ub = ua + hu;
    constexpr int s = S_LO + SU;             // global stage
    const Fe a0 = x[ua], a1 = x[ub];
    Fe d = fe_sub2(a0, a1, M2);
    // unit twiddles are skipped when known at compile time: the s = 0 stage, and j = 0 butterflies of the last group
    if constexpr (s > 0 && !(S_LO == 0 && (ua & (hu - 1)) == 0)) {
        const uint32_t j = ((ua & (hu - 1)) << S_LO) | post;              // r mod 2^s
        d = fe_mul29_t<true>(d, load_mul29(tws + ((size_t)j << (L - 1 - s)) * kTw29Words), P);
    }
    x[ua] = fe_add2(a0, a1, M2);
    x[ub] = d;
    if constexpr (B + 1 < (1 << (G - 1))) ntt_bfly<L, S_LO, G, SU, B + 1>(x, post, tws, P, M2);
    else if constexpr (SU > 0) ntt_bfly<L, S_LO, G, SU - 1, 0>(x, post, tws, P, M2);
}
template <int L, int S_HI, int S_LO>
ZK_D void ntt_group_stages(Fe (&x)[1 << (S_HI - S_LO + 1)], uint32_t post, const uint32_t *tws, const FieldParams &P, const Mod2p &M2) {
    constexpr int G = S_HI - S_LO + 1;
    ntt_bfly<L, S_LO, G, G - 1, 0>(x, post, tws, P, M2);
}

// One pass over one tile.  LAST = false: pass p < P (strided axis, inter-pass twiddle, same addresses in and out);
// LAST = true: pass P (contiguous axis, transposing store, optional scaling by n^-1 for the inverse transform).
// (measured on one box, 2^24 forward: two planes / 2 workgroups per CU 1.885 ms; this form, 3 per CU, 1.807; an unpadded swizzled
// plane with 48-byte twiddle records capped at 128 VGPRs for 4 per CU spills 27 dwords and is back at 1.889: profiles/r03_ntt_*)
template <int L, bool LAST>
__global__ __launch_bounds__(kNttThreads) void k_ntt_pass(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                          NttPlan pl, uint32_t pass, FieldParams P, Mul29 scale, int do_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t l = L, R = 1u << L;
    using GR = NttGroups<L>;
    constexpr int G0 = GR::g0, G1 = GR::g1, G2 = GR::g2;
    unsigned char *plane = smem;                                                          // ONE plane of R rows (halves take turns)
    uint32_t *tws = reinterpret_cast<uint32_t *>(smem + (size_t)R * kNttRowBytes);       // prepared omega_R^j, j < R/2

    uint32_t lo_sum = 0;   // log2 O_p
    for (uint32_t p = 0; p < pass; ++p) lo_sum += pl.l[p];
    const uint32_t log_inner = pl.log_n - lo_sum - l;           // log2 I_p
    const uint64_t inner = 1ull << log_inner;
    const uint32_t tid = threadIdx.x;

    // sub-DFT twiddles omega_R^j = omega_N^(j * N/R), prepared form
    for (uint32_t j = tid; j < R / 2; j += kNttThreads) store_mul29(tws + j * kTw29Words, ntt_twiddle(pl, (uint64_t)j << (pl.log_n - l), P));

    // ---- tile coordinates ----
    uint64_t base_in = 0, base_out = 0, tw_i0 = 0, out_stride_a = 0;
    uint32_t t_shift = 0;
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
        t_shift = log_q + l;
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
    const uint64_t *w_full = LAST ? nullptr : pl.w_full[pass];
    const Mod2p M2 = mod2p_of(P);

    // Every thread owns at most ONE work item per group ((R >> G) * kNttCols <= kNttThreads for R <= 2^8), so a group's 2^G
    // elements stay in its registers across the exchange: rows out / rows in are written and read one 16-byte half at a time.
    // ---- group 0: stages l-1 .. l-G0, rows (u << (l-G0)) | post, straight from HBM ----
    // thread -> (post, t).  MID: t fastest (512-B runs along the contiguous axis); LAST: post fastest (runs along the DFT axis)
    constexpr uint32_t items0 = (R >> G0) * kNttCols;
    static_assert(items0 <= (uint32_t)kNttThreads, "one item per thread and group");
    constexpr int S_LO0 = L - G0;
    const bool has0 = tid < items0;
    uint32_t t0, post0;
    if (!LAST) {
        t0 = tid & (kNttCols - 1);
        post0 = tid >> kNttColsLog;
    } else {
        post0 = tid & ((1u << S_LO0) - 1);
        t0 = tid >> S_LO0;
    }
    Fe x0[1 << G0];
    if (has0) {
#pragma unroll
        for (int u = 0; u < (1 << G0); ++u) {
            const uint32_t a = ((uint32_t)u << S_LO0) | post0;
            x0[u] = LAST ? fe_load(in, base_in + ((uint64_t)t0 << t_shift) + a) : fe_load(in, base_in + ((uint64_t)a << log_inner) + t0);
        }
    }
    __syncthreads();   // tws ready
    if (has0) ntt_group_stages<L, L - 1, S_LO0>(x0, post0, tws, P, M2);

    auto store_out = [&](uint32_t row, uint32_t t, Fe v) {
        const uint32_t k = __brev(row) >> (32 - l);   // DIF leaves frequency k in row bitrev(k)
        if (!LAST) {
            if (k) {                                                                       // omega_N^(O_p * i * k)
                Mul29 tw;
                if (w_full) {
                    const Fe f = fe_load(w_full, ((uint64_t)k << log_inner) + tw_i0 + t);
                    split29(f.v, tw.l);
                } else {
                    tw = ntt_twiddle(pl, ((tw_i0 + t) * k) << lo_sum, P);
                }
                v = fe_mul29_t<true>(v, tw, P);   // stays in [0, 2p): the next pass is lazy too
            }
            fe_store(out, base_out + ((uint64_t)k << log_inner) + t, v);
        } else {
            if (do_scale) v = fe_mul29(v, scale, P);   // (reduces fully)
            else v = fe_canon2(v, P);                  // the one reduction [0, 2p) -> [0, p) of the transform
            fe_store(out, base_out + t + (uint64_t)k * out_stride_a, v);
        }
    };
    if constexpr (GR::n == 1) {
        // (L <= 3 is never planned; kept for completeness)
    } else if constexpr (GR::n == 2) {
        constexpr int S_HI = L - G0 - 1;   // group 1 covers S_HI .. 0
        constexpr uint32_t items1 = (R >> G1) * kNttCols;
        static_assert(items1 <= (uint32_t)kNttThreads, "one item per thread and group");
        const bool has1 = tid < items1;
        const uint32_t t1 = tid & (kNttCols - 1), pre1 = tid >> kNttColsLog;
        Fe x1[1 << G1];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();   // everybody has read the low halves
            if (has0) {
#pragma unroll
                for (int u = 0; u < (1 << G0); ++u) lds_put_half(plane, ((uint32_t)u << S_LO0) | post0, t0, x0[u], half);
            }
            __syncthreads();
            if (has1) {
#pragma unroll
                for (int u = 0; u < (1 << G1); ++u) lds_get_half(plane, (pre1 << G1) | u, t1, x1[u], half);
            }
        }
        if (has1) {
            ntt_group_stages<L, S_HI, 0>(x1, 0, tws, P, M2);
#pragma unroll
            for (int u = 0; u < (1 << G1); ++u) store_out((pre1 << G1) | u, t1, x1[u]);
        }
    } else {
        constexpr int S_HI1 = L - G0 - 1, S_LO1 = S_HI1 - G1 + 1;   // group 1: S_HI1 .. S_LO1, group 2: S_LO1-1 .. 0
        constexpr uint32_t items1 = (R >> G1) * kNttCols, items2 = (R >> G2) * kNttCols;
        static_assert(items1 <= (uint32_t)kNttThreads && items2 <= 2 * (uint32_t)kNttThreads, "one item per thread in group 1, at most two in group 2");
        constexpr uint32_t IPT2 = items2 > (uint32_t)kNttThreads ? 2 : 1;   // (L = 8: 3 + 3 + 2 stages -> two 4-row items per thread)
        const bool has1 = tid < items1;
        const uint32_t t1 = tid & (kNttCols - 1), rr1 = tid >> kNttColsLog;
        const uint32_t post1 = rr1 & ((1u << S_LO1) - 1), pre1 = rr1 >> S_LO1;
        Fe x1[1 << G1];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            if (has0) {
#pragma unroll
                for (int u = 0; u < (1 << G0); ++u) lds_put_half(plane, ((uint32_t)u << S_LO0) | post0, t0, x0[u], half);
            }
            __syncthreads();
            if (has1) {
#pragma unroll
                for (int u = 0; u < (1 << G1); ++u) lds_get_half(plane, (pre1 << (S_HI1 + 1)) | ((uint32_t)u << S_LO1) | post1, t1, x1[u], half);
            }
        }
        if (has1) ntt_group_stages<L, S_HI1, S_LO1>(x1, post1, tws, P, M2);
        Fe x2[IPT2][1 << G2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();   // the plane is free again (group 1's reads of the high halves / group 2's of the low halves are done)
            if (has1) {
#pragma unroll
                for (int u = 0; u < (1 << G1); ++u) lds_put_half(plane, (pre1 << (S_HI1 + 1)) | ((uint32_t)u << S_LO1) | post1, t1, x1[u], half);
            }
            __syncthreads();
#pragma unroll
            for (uint32_t w = 0; w < IPT2; ++w) {
                const uint32_t it = tid + w * kNttThreads, t2 = it & (kNttCols - 1), pre2 = it >> kNttColsLog;
                if (it < items2) {
#pragma unroll
                    for (int u = 0; u < (1 << G2); ++u) lds_get_half(plane, (pre2 << G2) | u, t2, x2[w][u], half);
                }
            }
        }
#pragma unroll
        for (uint32_t w = 0; w < IPT2; ++w) {
            const uint32_t it = tid + w * kNttThreads, t2 = it & (kNttCols - 1), pre2 = it >> kNttColsLog;
            if (it < items2) {
                ntt_group_stages<L, G2 - 1, 0>(x2[w], 0, tws, P, M2);
#pragma unroll
                for (int u = 0; u < (1 << G2); ++u) store_out((pre2 << G2) | u, t2, x2[w][u]);
            }
        }
    }
}

// full inter-pass table of one middle pass: out[k * inner + i] = omega^(scale * i * k) * 2^5 mod p
__global__ __launch_bounds__(kBlock) void k_ntt_full_table(uint64_t *__restrict__ out, NttPlan pl, uint32_t log_inner, uint32_t log_r,
                                                           uint32_t log_scale, FieldParams P) {
    const uint64_t total = 1ull << (log_inner + log_r), stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += stride) {
        const uint64_t k = idx >> log_inner, i = idx & ((1ull << log_inner) - 1);
        const uint64_t e = (i * k) << log_scale;
        const uint64_t lo = e & ((1ull << pl.lo_bits) - 1), hi = e >> pl.lo_bits;
        // omega^e * 2^5 as a plain element: hi-table entries carry the factor already; a lo-only value is rebuilt from its limbs
        Fe v;
        if (lo == 0) {
            v = fe_load(pl.w_hi, hi);
        } else {
            const Mul29 m = load_mul29(pl.w_lo + lo * kTw29Words);
            if (hi == 0) {
                // merge the 29-bit limbs of the prepared value back into an element (value < p)
                uint32_t w[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const int bit = 32 * x, li = bit / 29, sh = bit - 29 * li;
                    uint32_t t = m.l[li] >> sh;
                    t |= m.l[li + 1] << (29 - sh);
                    if (29 - sh + 29 < 32 && li + 2 < 9) t |= m.l[li + 2] << (58 - sh);
                    w[x] = t;
                }
#pragma unroll
                for (int x = 0; x < 8; ++x) v.v[x] = w[x];
            } else {
                v = fe_mul29(fe_load(pl.w_hi, hi), m, P);
            }
        }
        fe_store(out, idx, v);
    }
}

// two-level prepared twiddle table: w_lo[i] = Mul29(omega^i), i < 2^lo_bits; w_hi[i] = omega^(i << lo_bits) * 2^5 mod p
__global__ __launch_bounds__(kBlock) void k_ntt_tables(uint32_t *__restrict__ w_lo, uint64_t *__restrict__ w_hi, uint32_t lo_bits,
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
        if (is_hi) {
#pragma unroll
            for (int d = 0; d < 5; ++d) acc = fe_add(acc, acc, P);
            fe_store(w_hi, idx - n_lo, acc);
        } else {
            store_mul29(w_lo + idx * kTw29Words, mul29_prepare(acc, P));
        }
    }
}

}  // namespace zk
