// transcript.cuh -- the device-side Fiat-Shamir step (lane-parallel Keccak-256 sponge on one wave), shared by the
// translation units that run it (capi.hip: k_round_tail / k_finish; pipe.hip: the pipelined rounds).
#pragma once
#include "common.cuh"
#include "keccak.hpp"

namespace zk {

// ---- device-side Fiat-Shamir step (sumcheck/src/prover.rs:59-62 on one GPU lane) --------------------------------------
// absorb the round polynomial (32-byte BE canonical elements, sumcheck/src/lib.rs:23-29), squeeze the challenge
// (transcript/src/lib.rs:20-30) and publish it in Montgomery form for the next round's fused fold.
// Lane-parallel sponge.  State word A[x][y] lives on lane 8y + x + 1 of one wave: each plane (fixed y) occupies an
// 8-lane group whose slots 1..5 are the primaries x = 0..4, slot 0 mirrors x = 4 and slots 6, 7 mirror x = 0, 1, so the
// row neighbours that theta and chi need are DPP row shifts.  Planes (0,1), (2,3) and (4, zeros) share 16-lane rows, so
// theta's column parity is a same-slot XOR over half-rows, rows and wave halves: row_ror:8, then v_permlane16_swap /
// v_permlane32_swap (gfx950) -- all VALU, with the lo and hi words sharing the swaps (the first swap leaves the lo parity
// in the even rows and the hi parity in the odd rows of ONE register).  Only pi crosses lanes through ds_bpermute, always
// from PRIMARY lanes, so the mirrors are rebuilt every round (slot 0 stays valid through chi; slots 6, 7 are only read by
// chi right after pi).  Lanes 40..63 hold zeros and keep them.  The 24-round permutation is the latency floor of every
// sumcheck round (prover.rs:59-62 is inherently serial), so it is built for latency: ~40 VALU + 2 ds_bpermute per round,
// 2.7 us per permutation (the all-ds_bpermute version it replaced: 4.9 us; tools/mb/mb_tail.hip).
struct LaneKeccak {
    int lane, index;      // index = x + 5y for primary lanes, -1 otherwise
    int src_pi;           // lane whose rotated word lands here (pi)
    uint32_t rot;         // rho as a right-rotate by 32*swap + rot
    uint32_t rc_lo, rc_hi;   // lane r < 24 holds the round constant RC[r]
    bool swap, s0, s5, act;
};
ZK_D uint64_t shfl64(uint64_t v, int src) {
    const uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
template <int CTRL>
ZK_D uint32_t dpp32(uint32_t v) {   // row_shl:n = 0x100 + n (lane i reads lane i+n), row_shr:n = 0x110 + n, row_ror:n = 0x120 + n
    return __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xF, 0xF, true);
}
ZK_D int keccak_lane_of(int x, int y) { return 8 * y + x + 1; }
ZK_D LaneKeccak lane_keccak_init() {
    // rho offsets indexed by x + 5y
    constexpr uint8_t kRho[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    constexpr uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    LaneKeccak L;
    const int lane = threadIdx.x & 63;
    L.lane = lane;
    const uint64_t rc = RC[lane < 24 ? lane : 0];
    L.rc_lo = (uint32_t)rc;
    L.rc_hi = (uint32_t)(rc >> 32);
    const int slot = lane & 7, y = lane >> 3;
    L.act = y < 5;
    L.s0 = slot == 0;
    L.s5 = slot == 5 && y < 5;
    if (y < 5) {
        const int x = slot == 0 ? 4 : (slot >= 6 ? slot - 6 : slot - 1);   // mirrors carry their primary's x
        L.index = (slot >= 1 && slot <= 5) ? x + 5 * y : -1;
        // pi: B[y'][2x'+3y'] = A[x'][y'], i.e. destination (X, Y) = (y', 2x'+3y'); for destination (x, y) the source is
        // x' = (x + 3y) mod 5, y' = x
        const int sx = (x + 3 * y) % 5, sy = x;
        L.src_pi = keccak_lane_of(sx, sy);
        uint32_t rot = 0;
        const int idx = x + 5 * y;
#pragma unroll
        for (int i = 0; i < 25; ++i)
            if (i == idx) rot = kRho[i];
        const uint32_t q = (64 - rot) & 63;   // rotl by rot == rotr by q
        L.swap = q >= 32;
        L.rot = q & 31;
    } else {
        L.index = -1;
        L.src_pi = lane;
        L.rot = 0;
        L.swap = false;
    }
    return L;
}
ZK_D uint64_t lane_keccak_f1600(uint64_t a, const LaneKeccak &L) {
    uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
    {   // callers maintain the primaries only: make slot 0 (mirror of x = 4, slot 5) valid on entry
        const uint32_t ml = dpp32<0x105>(lo), mh = dpp32<0x105>(hi);
        lo = L.s0 ? ml : lo;
        hi = L.s0 ? mh : hi;
    }
    for (int round = 0; round < 24; ++round) {
        // theta: same-slot XOR over the 5 planes ...
        uint32_t cl = lo ^ dpp32<0x128>(lo), ch = hi ^ dpp32<0x128>(hi);   // planes sharing a row (row_ror:8)
        {
            const auto r = __builtin_amdgcn_permlane16_swap(cl, ch, false, false);
            const uint32_t z = r[0] ^ r[1];                                 // rows: [lo01, hi01, lo23, hi23]
            const auto s = __builtin_amdgcn_permlane32_swap(z, z, false, false);
            const uint32_t w = s[0] ^ s[1];                                 // rows: [lo, hi, lo, hi]
            const auto e = __builtin_amdgcn_permlane16_swap(w, w, false, false);
            cl = e[0];                                                      // every row: C lo / C hi
            ch = e[1];
        }
        // ... then D = C[x-1] ^ rotl(C[x+1], 1): slot s reads slots s-1 and s+1, except x = 4 (slot 5), whose x+1 = 0 is slot 1
        const uint32_t ml = dpp32<0x111>(cl), mh = dpp32<0x111>(ch);
        uint32_t pl = dpp32<0x101>(cl), ph = dpp32<0x101>(ch);
        const uint32_t wl = dpp32<0x114>(cl), wh = dpp32<0x114>(ch);
        pl = L.s5 ? wl : pl;
        ph = L.s5 ? wh : ph;
        const uint32_t dl = ml ^ __builtin_amdgcn_alignbit(pl, ph, 31), dh = mh ^ __builtin_amdgcn_alignbit(ph, pl, 31);
        lo ^= L.act ? dl : 0u;
        hi ^= L.act ? dh : 0u;
        // rho (right-rotate my word by 32*swap + rot) + pi (fetch the word that lands here, always from a primary lane)
        const uint32_t sl = L.swap ? hi : lo, sh = L.swap ? lo : hi;
        const uint32_t rl = __builtin_amdgcn_alignbit(sh, sl, L.rot), rh = __builtin_amdgcn_alignbit(sl, sh, L.rot);
        const uint32_t bl = __shfl(rl, L.src_pi, 64), bh = __shfl(rh, L.src_pi, 64);
        // chi + iota: row neighbours x+1, x+2 are slots s+1, s+2
        lo = bl ^ (~dpp32<0x101>(bl) & dpp32<0x102>(bl));
        hi = bh ^ (~dpp32<0x101>(bh) & dpp32<0x102>(bh));
        // iota: the round constant sits in lane `round` of L.rc_* (a scalar load here would share lgkmcnt with the
        // ds_bpermute above and serialise on its latency)
        const uint32_t rcl = __builtin_amdgcn_readlane(L.rc_lo, round), rch = __builtin_amdgcn_readlane(L.rc_hi, round);
        if (L.index == 0) {
            lo ^= rcl;
            hi ^= rch;
        }
    }
    return ((uint64_t)hi << 32) | lo;
}
struct LaneSponge {   // word-cursor sponge (see WordSponge) spread over the primary lanes
    uint64_t a;
    uint32_t pos;
};
ZK_D void lane_absorb_word(LaneSponge &sp, uint64_t w, const LaneKeccak &L) {   // w wave-uniform
    if ((uint32_t)L.index == sp.pos) sp.a ^= w;
    if (++sp.pos == 17) {
        sp.a = lane_keccak_f1600(sp.a, L);
        sp.pos = 0;
    }
}

// One transcript step on one wave (all 64 lanes execute it; lanes >= 25 idle along):
// absorb the round polynomial as 32-byte big-endian canonical elements (sumcheck/src/lib.rs:23-29), squeeze the
// challenge (transcript/src/lib.rs:20-30) and return it in Montgomery form (wave-uniform).
ZK_D LaneSponge lane_sponge_load(const WordSponge *gsp, const LaneKeccak &L) {
    LaneSponge sp;
    sp.a = (L.index >= 0) ? gsp->s[L.index] : 0ull;
    sp.pos = __builtin_amdgcn_readfirstlane(gsp->pos);
    return sp;
}
ZK_D void lane_sponge_store(WordSponge *gsp, const LaneSponge &sp, const LaneKeccak &L) {
    if (L.index >= 0) gsp->s[L.index] = sp.a;
    if (L.lane == 0) gsp->pos = sp.pos;
}
// absorb ns elements (Montgomery form; read by lane, so `sums` may be LDS or global memory) as 32-byte big-endian canonical images
// CANON: the elements are already canonical integers (the pipelined rounds evaluate their sums straight into that form).
template <bool CANON = false>
ZK_D void lane_absorb_elems(LaneSponge &sp, const LaneKeccak &L, const Fe *sums, uint32_t ns, const FieldParams &P) {
    for (uint32_t base = 0; base < ns; base += 64) {
        // lane t converts sum (base + t): Montgomery -> canonical, in parallel across lanes
        const uint32_t mine = base + (uint32_t)L.lane < ns ? base + (uint32_t)L.lane : ns - 1;
        const Fe c = CANON ? sums[mine] : fe_to_canonical(sums[mine], P);
        uint64_t w[4];   // the element's 32-byte big-endian image as 4 little-endian lane words
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t limb = (uint64_t)c.v[2 * (3 - k)] | ((uint64_t)c.v[2 * (3 - k) + 1] << 32);
            w[k] = WordSponge::bswap64(limb);
        }
        const uint32_t cnt = ns - base < 64 ? ns - base : 64, total = 4 * cnt;
        // absorb the batch rate-block by rate-block: state word i receives message word done + (i - pos) = word k of sum t,
        // one gather per k instead of a dependent step per word (a batch may straddle a permutation: D = 3 absorbs 16 words
        // at cursor 4)
        for (uint32_t done = 0; done < total;) {
            const uint32_t room = 17 - sp.pos, take = total - done < room ? total - done : room;
            const int rel = L.index - (int)sp.pos;
            const uint32_t m = done + (rel >= 0 ? (uint32_t)rel : 0u);
            const int t = (int)(m >> 2), k = (int)(m & 3);
            const uint64_t g0 = shfl64(w[0], t), g1 = shfl64(w[1], t), g2 = shfl64(w[2], t), g3 = shfl64(w[3], t);
            const uint64_t g = k == 0 ? g0 : (k == 1 ? g1 : (k == 2 ? g2 : g3));
            if (L.index >= 0 && rel >= 0 && rel < (int)take) sp.a ^= g;
            sp.pos += take;
            done += take;
            if (sp.pos == 17) {
                sp.a = lane_keccak_f1600(sp.a, L);
                sp.pos = 0;
            }
        }
    }
}
// sample_field_element (transcript/src/lib.rs:20-30) in three pieces, so that a kernel can keep only what the NEXT round needs on
// the transcript wave (a lone wave issues one instruction per ~10 cycles: every instruction there is on the critical path):
//   lane_squeeze_x   : pad, permute, reset to the digest (transcript/src/lib.rs:22-23); returns int(digest, big endian) as a raw
//                      256-bit integer x (wave-uniform), NOT reduced mod p
//   challenge29_of(x): the prepared multiplier form of the challenge (what folds and the next evaluation read)
//   challenge_fe_of(x): the challenge in Montgomery form (what the proof records) -- any wave may compute it later
// No reduction of x first: the carry-free multiplier takes any 256-bit integer (nine 29-bit limbs hold 261 bits; a * c / 2^261
// < 2^250 for a < 2^256, c < p, so the value is < 2p before its final conditional subtraction) and returns the canonical
// representative of x * c * 2^-261 mod p either way; fe_reduce_u256's five conditional 9-limb subtractions are not needed.
ZK_D Fe lane_squeeze_x(LaneSponge &sp, const LaneKeccak &L) {
    // squeeze: pad10*1 with Keccak's 0x01 domain byte, permute, digest = words 0..3
    if ((uint32_t)L.index == sp.pos) sp.a ^= 0x01ull;
    if (L.index == 16) sp.a ^= 0x8000000000000000ull;
    sp.a = lane_keccak_f1600(sp.a, L);
    const uint64_t d0 = shfl64(sp.a, keccak_lane_of(0, 0)), d1 = shfl64(sp.a, keccak_lane_of(1, 0));
    const uint64_t d2 = shfl64(sp.a, keccak_lane_of(2, 0)), d3 = shfl64(sp.a, keccak_lane_of(3, 0));
    // finalize_reset + update(digest) (transcript/src/lib.rs:22-23): state = digest words, cursor 4
    sp.a = (L.index >= 0 && L.index < 4) ? sp.a : 0ull;
    sp.pos = 4;
    // int(digest, big endian) (transcript/src/lib.rs:29)
    const uint64_t h[4] = {WordSponge::bswap64(d3), WordSponge::bswap64(d2), WordSponge::bswap64(d1), WordSponge::bswap64(d0)};
    Fe x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x.v[2 * i] = (uint32_t)h[i];
        x.v[2 * i + 1] = (uint32_t)(h[i] >> 32);
    }
    return x;
}
ZK_D Mul29 challenge29_of(const Fe &x, const FieldParams &P) {
    Mul29 k1, ch29;
#pragma unroll
    for (int i = 0; i < 9; ++i) k1.l[i] = P.r2s_29[i];   // prepare(R^2 * 2^5): fe_mul29(x, k1) = x * R^2 * 2^5 * 2^-256 = (x * R) * 2^5
    const Fe chs = fe_mul29(x, k1, P);
    split29(chs.v, ch29.l);
    return ch29;
}
// Both prepared forms of the challenge in ONE multiplication, lane-parallel: lane 0 multiplies the digest by R^2 * 2^5 (the
// Montgomery-form challenge times 2^5: split, it is the multiplier folds use), lane 16 by 2^266 (the canonical challenge times
// 2^5: split, the multiplier that turns a Montgomery-form value into a CANONICAL product).  Valid on lanes 0 / 16 respectively.
ZK_D Mul29 challenge29_both(const Fe &x, const Mul29 &k266, uint32_t lane, const FieldParams &P) {
    Mul29 k;
#pragma unroll
    for (int i = 0; i < 9; ++i) k.l[i] = lane == 16 ? k266.l[i] : P.r2s_29[i];
    const Fe v = fe_mul29(x, k, P);
    Mul29 m;
    split29(v.v, m.l);
    return m;
}
// lane 0 stores the Montgomery prepared form, lane 16 the canonical one (the values challenge29_both left on them)
ZK_D void publish_challenge29_both(uint64_t *d_challenge, const Mul29 &m, uint32_t lane) {
    if (lane == 0 || lane == 16) {
        uint32_t *rec = reinterpret_cast<uint32_t *>(d_challenge) + (lane == 0 ? 8 : kChalCanonWord);
#pragma unroll
        for (int i = 0; i < 9; ++i) rec[i] = m.l[i];
    }
}
ZK_D Fe challenge_fe_of(const Fe &x, const FieldParams &P) {
    Mul29 k0;
#pragma unroll
    for (int i = 0; i < 9; ++i) k0.l[i] = P.r2_29[i];    // prepare(R^2): fe_mul29(x, k0) = x * R^2 * 2^-256 = x * R
    return fe_mul29(x, k0, P);
}
// the three together: challenge in Montgomery form (wave-uniform), ch29 = its prepared multiplier form
ZK_D Fe lane_squeeze(LaneSponge &sp, const LaneKeccak &L, const FieldParams &P, Mul29 &ch29) {
    const Fe x = lane_squeeze_x(sp, L);
    ch29 = challenge29_of(x, P);
    return challenge_fe_of(x, P);
}
// The two forms the classic tails publish -- the prepared multiplier (valid on lane 0) and the Montgomery-form challenge
// (valid on lane 16) -- from ONE multiplication with per-lane multipliers R^2 * 2^5 / R^2, instead of two one after the other
// (a lone wave pays ~0.8 us for each; they sit on the serial chain of every round that has a tail kernel).
ZK_D void challenge_forms(const Fe &x, uint32_t lane, const FieldParams &P, Mul29 &ch29, Fe &ch) {
    Mul29 k;
#pragma unroll
    for (int i = 0; i < 9; ++i) k.l[i] = lane == 16 ? P.r2_29[i] : P.r2s_29[i];
    ch = fe_mul29(x, k, P);
    split29(ch.v, ch29.l);
}
ZK_D Fe transcript_step(LaneSponge &sp, const LaneKeccak &L, const Fe *sums, uint32_t ns, const FieldParams &P, Mul29 &ch29) {
    lane_absorb_elems(sp, L, sums, ns, P);
    return lane_squeeze(sp, L, P, ch29);
}
// publish a challenge for the next round's fused fold: [Fe r][Mul29 of r] (common.cuh, kChallengeBytes); the two halves may
// come from different waves (publish_challenge29 from the transcript wave, publish_challenge_fe from whoever converts)
ZK_D void publish_challenge29(uint64_t *d_challenge, const Mul29 &ch29, int lane) {
    if (lane == 0) {
        uint32_t *rec = reinterpret_cast<uint32_t *>(d_challenge) + 8;
#pragma unroll
        for (int i = 0; i < 9; ++i) rec[i] = ch29.l[i];
    }
}
ZK_D void publish_challenge_fe(uint64_t *d_challenge, uint64_t *out_ch, const Fe &ch, int lane) {
    if (lane == 0) {
        if (d_challenge) fe_store(d_challenge, 0, ch);
        if (out_ch) fe_store(out_ch, 0, ch);
    }
}
ZK_D void publish_challenge_forms(uint64_t *d_challenge, uint64_t *out_ch, const Fe &ch, const Mul29 &ch29, uint32_t lane) {
    if (lane == 0) publish_challenge29(d_challenge, ch29, 0);
    if (lane == 16) publish_challenge_fe(d_challenge, out_ch, ch, 0);
}
ZK_D void publish_challenge(uint64_t *d_challenge, uint64_t *out_ch, const Fe &ch, const Mul29 &ch29, int lane) {
    publish_challenge_fe(d_challenge, out_ch, ch, lane);
    publish_challenge29(d_challenge, ch29, lane);
}
ZK_D void transcript_round(WordSponge *gsp, const Fe *sums, uint32_t ns, uint64_t *d_challenge, uint64_t *out_ch,
                           const FieldParams &P) {
    const LaneKeccak L = lane_keccak_init();
    LaneSponge sp = lane_sponge_load(gsp, L);
    lane_absorb_elems(sp, L, sums, ns, P);
    Mul29 ch29;
    Fe ch;
    challenge_forms(lane_squeeze_x(sp, L), (uint32_t)L.lane, P, ch29, ch);
    publish_challenge_forms(d_challenge, out_ch, ch, ch29, (uint32_t)L.lane);
    lane_sponge_store(gsp, sp, L);
}

}  // namespace zk
