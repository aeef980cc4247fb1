// eval_kernels.cuh -- MultiLinearPolynomial::evaluate (polynomial/src/multilinear/evaluation_form.rs:83-89) on big tables: the
// streaming form of k_eval_low (kernels.cuh), which the reference's own criterion bench times (polynomial/benches/
// polynomial_evaluation.rs:85-105).
//
// Same contract as k_eval_low: out[g] = sum_{x < 2^L} in[g * 2^L + x] * eq(point_low, x), the table after the LOW L variables
// have been assigned (restriction commutes, exact field arithmetic: the same canonical elements as the reference's fold order),
// one workgroup per out[g], L = 8..15.  What is different is how a workgroup spends its instructions and its loads:
//
//  * HALF an element per lane.  A wave reads a row of its table as ONE fully coalesced 1-KiB dwordx4 access (nontemporal): lane l
//    holds the 128-bit half (l & 1) of element (l >> 1).  evaluate is a SUM, so the halves never have to meet: an element is
//    lo + 2^128 hi, the even lane accumulates lo * w, the odd lane hi * w, and the factor 2^128 is folded into the odd lane's
//    final weight.  No DPP half swap (k_fold_msb needs one because it must write whole elements back).
//  * Carry-free accumulation.  The 128-bit half is split into five 29-bit limbs, the row weight comes as nine 29-bit limbs, and
//    the 45 limb products go straight into thirteen 64-bit column sums with v_mad_u64_u32 -- no v_addc at all (field.cuh: on this
//    part a carry add costs what a multiply costs).  A column takes 8 rows (40 products < 2^58) between carry normalisations.
//    ~62 VALU instructions per half element (124 per element) against ~195 + 100 for k_eval_low's wide_mac + per-thread epilogue.
//  * The row weight is WAVE-UNIFORM (row m = index bits 7..L-1, the lane only chooses bits 0..6 and the half), so the workgroup
//    builds the <= 256 row weights once (four 16-entry eq tables -> one multiplication per entry), parks them in LDS as limbs and
//    every lane reads the same 36 bytes per row (broadcast reads).
//  * One Montgomery reduction, one multiplication by the lane's weight eq(bits 0..6) (* 2^128 for the odd lanes) and one workgroup
//    sum per 256 rows instead of per 16 elements.
//
// Bit-exactness: every quantity is an exact integer until the one reduction; sum_m half_m * A_m < 2^8 * 2^128 * p < p * R, so
// redc returns the canonical representative of (sum) * R^-1; with A_m = a_m R (Montgomery) that is sum half_m a_m mod p, and
// lo-part + 2^128 * hi-part = sum T_m a_m = the Montgomery form of the restricted value.
#pragma once
#include "common.cuh"

namespace zk {

constexpr int kEvalStreamMax = 15, kEvalStreamMin = 10;   // L: 7 lane bits + 3..8 row bits (a multiple of the prefetch depth in rows)
constexpr int kEvalStreamPrefetch = 8;                   // rows in flight per lane (8 x 16 B), also the normalisation period
struct EvalStreamPoint {   // r for index bit p (Montgomery form), p = 0 the least significant bit = the LAST variable
    uint32_t r[kEvalStreamMax][8];
    uint32_t c128[8];      // 2^128 in Montgomery form (2^384 mod p): what an odd lane's sum is worth more than an even lane's
};

// 128-bit half -> five 29-bit limbs (29, 29, 29, 29, 12 bits)
ZK_D void split29_half(const uint4 &x, uint32_t (&h)[5]) {
    constexpr uint32_t M = (1u << 29) - 1;
    h[0] = x.x & M;
    h[1] = __builtin_amdgcn_alignbit(x.y, x.x, 29) & M;   // bits 29..57
    h[2] = __builtin_amdgcn_alignbit(x.z, x.y, 26) & M;   // bits 58..86
    h[3] = __builtin_amdgcn_alignbit(x.w, x.z, 23) & M;   // bits 87..115
    h[4] = x.w >> 20;                                     // bits 116..127
}

struct EvalCols {
    uint64_t c[14];   // column k has weight 2^(29 k); 13 product columns + one for the carries
};
// cols += h * w  (5 x 9 limb products).  ONE asm statement: every product is a v_mad_u64_u32 straight into its 64-bit column (the
// compiler, left alone, sums a row's products from zero and adds the row sum with a v_lshl_add_u64 per column), and nothing is
// padded between statements (field.cuh mac_col).  i-major order: nine consecutive instructions write nine different columns.
ZK_D void eval_mac(EvalCols &a, const uint32_t (&h)[5], const uint32_t (&w)[9]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    asm("v_mad_u64_u32 %0, vcc, %13, %18, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %13, %19, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %13, %20, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %13, %21, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %13, %22, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %13, %23, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %13, %24, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %13, %25, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %13, %26, %8\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %18, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %14, %19, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %14, %20, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %14, %21, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %14, %22, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %14, %23, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %14, %24, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %14, %25, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %14, %26, %9\n\t"
        "v_mad_u64_u32 %2, vcc, %15, %18, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %15, %19, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %15, %20, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %15, %21, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %15, %22, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %15, %23, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %15, %24, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %15, %25, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %15, %26, %10\n\t"
        "v_mad_u64_u32 %3, vcc, %16, %18, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %16, %19, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %16, %20, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %16, %21, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %16, %22, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %16, %23, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %16, %24, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %16, %25, %10\n\t"
        "v_mad_u64_u32 %11, vcc, %16, %26, %11\n\t"
        "v_mad_u64_u32 %4, vcc, %17, %18, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %17, %19, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %17, %20, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %17, %21, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %17, %22, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %17, %23, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %17, %24, %10\n\t"
        "v_mad_u64_u32 %11, vcc, %17, %25, %11\n\t"
        "v_mad_u64_u32 %12, vcc, %17, %26, %12"
        : "+v"(a.c[0]), "+v"(a.c[1]), "+v"(a.c[2]), "+v"(a.c[3]), "+v"(a.c[4]), "+v"(a.c[5]), "+v"(a.c[6]), "+v"(a.c[7]), "+v"(a.c[8]),
          "+v"(a.c[9]), "+v"(a.c[10]), "+v"(a.c[11]), "+v"(a.c[12])
        : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]),
          "v"(w[6]), "v"(w[7]), "v"(w[8])
        : "vcc");
#else
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) a.c[i + j] += (uint64_t)h[i] * w[j];
#endif
}
ZK_D void eval_normalise(EvalCols &a) {
    constexpr uint64_t M = (1ull << 29) - 1;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        a.c[k + 1] += a.c[k] >> 29;
        a.c[k] &= M;
    }
}

// 320 threads: waves 0-3 are the workers described above; wave 4 only multiplies up eq(point_high, blockIdx.x) when the outputs
// are weighted (ph.n > 0, kernels.cuh eval_high_weight) -- all four worker waves have a table to build -- and then just keeps the
// barriers company.
constexpr int kEvalStreamThreads = kBlock + 64;
__global__ __launch_bounds__(kEvalStreamThreads) void k_eval_stream(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t L,
                                                                      EvalStreamPoint pt, FieldParams P, EvalHighPoint ph) {
    __shared__ Fe eq[4][16];                       // bits 0-3, 4-6, 7-10, 11-14
    __shared__ Fe wg;
    __shared__ __attribute__((aligned(16))) uint32_t A[256][12];   // row weights as 9 limbs (+ padding to 48 bytes)
    __shared__ uint32_t red[kBlock / 64][8];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t MB = L - 7, rows = 1u << MB;
    if (wave == 4) {   // the weight wave: no loads, no rows
        if (ph.n) {
            const Fe f = eval_high_weight(ph, blockIdx.x, lane, P);
            if (lane == eval_high_lane(ph.n)) wg = f;
        }
        __syncthreads();
        __syncthreads();
        __syncthreads();
        return;
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(in) + (((uint64_t)blockIdx.x << L) << 1) + tid;   // row m: + 256 * m
    constexpr int PF = kEvalStreamPrefetch;
    uint4 x[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) x[i] = nt_load16(src + (uint64_t)i * 256);
    // ---- the four small eq tables: wave w builds table w, lane j < 16 entry j (two levels of multiplications, under the loads)
    {
        const uint32_t first = wave == 0 ? 0u : wave == 1 ? 4u : wave == 2 ? 7u : 11u;
        const uint32_t want = wave == 1 ? 3u : 4u;
        const uint32_t nb = first >= L ? 0u : (L - first < want ? L - first : want);
        const Fe acc = eq_entry_depth2(pt, nb ? first : 0u, nb, lane, P);   // two levels of multiplications (kernels.cuh)
        if (lane < 16) eq[wave][lane] = acc;
    }
    __syncthreads();
    if (tid < rows) {   // row weight A[m] = eq(bits 11..14)[m >> 4] * eq(bits 7..10)[m & 15], split into limbs
        Fe a = eq[2][tid & 15];
        if (MB > 4) a = fe_mul(eq[3][tid >> 4], a, P);
        uint32_t l[9];
        split29(a.v, l);
#pragma unroll
        for (int i = 0; i < 9; ++i) A[tid][i] = l[i];
    }
    // the lane's own weight: eq(bits 0..6) of its element, times 2^128 for the lanes that hold high halves
    const uint32_t e = (wave << 5) | (lane >> 1);
    Fe wl = fe_mul(eq[1][e >> 4], eq[0][e & 15], P);
    {
        Fe c128;
#pragma unroll
        for (int i = 0; i < 8; ++i) c128.v[i] = pt.c128[i];
        const Fe wh = fe_mul(wl, c128, P);
        if (lane & 1) wl = wh;
    }
    __syncthreads();
    // ---- the rows
    EvalCols acc;
#pragma unroll
    for (int k = 0; k < 14; ++k) acc.c[k] = 0;
    // Four rows at a time: split the four loaded halves into limbs, hand their registers straight back to the loads of the rows
    // PF ahead, then do the four products.  A buffer's loads are issued at the start of its own phase and next needed at the start
    // of its next one -- eight rows of arithmetic later.  (As compiled today the allocator rotates the eight buffers at the loop
    // head and waits for all of them there, so the second set's lead is four rows, not eight; four waves per SIMD cover it --
    // kernel 97 us at 2^24 against 85 us for the bare stream.)
    auto phase = [&](uint32_t m, int base, bool reload) {
        __builtin_amdgcn_sched_barrier(0);   // keep the phases apart: the scheduler otherwise merges both splits and all eight loads
        uint32_t h[4][5];
#pragma unroll
        for (int i = 0; i < 4; ++i) split29_half(x[base + i], h[i]);
        if (reload) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[base + i] = nt_load16(src + (uint64_t)(m + i + PF) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t w[9];
            const uint4 w0 = *reinterpret_cast<const uint4 *>(&A[m + i][0]), w1 = *reinterpret_cast<const uint4 *>(&A[m + i][4]);
            w[0] = w0.x, w[1] = w0.y, w[2] = w0.z, w[3] = w0.w;
            w[4] = w1.x, w[5] = w1.y, w[6] = w1.z, w[7] = w1.w;
            w[8] = A[m + i][8];
            eval_mac(acc, h[i], w);
        }
    };
    static_assert(PF == 8, "two sets of four rows");
    uint32_t m0 = 0;
    for (; m0 + PF < rows; m0 += PF) {   // rows is a multiple of PF (L >= 10)
        phase(m0, 0, true);
        phase(m0 + 4, 4, true);
        eval_normalise(acc);   // 8 rows x 5 products < 2^58 each, on top of < 2^36 left by the last normalisation: < 2^64
    }
    phase(m0, 0, false);
    phase(m0 + 4, 4, false);
    eval_normalise(acc);
    // ---- one reduction per lane: columns (29-bit limbs now, the top one < 2^35) -> 32-bit words -> redc
    uint32_t t[16];
#pragma unroll
    for (int wd = 0; wd < 16; ++wd) {
        const int bit = 32 * wd, i = bit / 29, sh = bit - 29 * i;   // word wd starts inside limb i at offset sh
        uint32_t v = 0;
        if (i < 14) v = (uint32_t)(acc.c[i] >> sh);
        if (i + 1 < 14) v |= (uint32_t)(acc.c[i + 1] << (29 - sh));
        if (29 - sh + 29 < 32 && i + 2 < 14) v |= (uint32_t)(acc.c[i + 2] << (58 - sh));
        t[wd] = v;
    }
    Fe s = fe_mul(redc(t, P), wl, P);
    s = fe_wave_sum(s, P);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid >> 6][i] = s.v[i];
    }
    __syncthreads();
    if (tid == 0) {
        Fe tot = s;
        for (int wv = 1; wv < kBlock / 64; ++wv) {
            Fe o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = red[wv][i];
            tot = fe_add(tot, o, P);
        }
        if (ph.n) tot = fe_mul(tot, wg, P);
        fe_store(out, blockIdx.x, tot);
    }
}

}  // namespace zk
