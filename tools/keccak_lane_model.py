"""numpy model of a SPLIT-HALF lane-parallel Keccak-f[1600] (one wave: low 32 bits of word (x, y) on lane 16*(y >= 3) + 5*(y % 3) + x,
high 32 bits 32 lanes up; every step ONE 32-bit instruction for the whole state; pi + chi as three ds_bpermute fetches).
Round 3 built it in HIP from this model (bit-exact on the GPU: 117 transcript tests) and measured it against the shipped layout
(both halves on one lane, everything issued twice): 30 VALU + 4 ds_bpermute per round instead of 44 + 2, yet 2.88 us per
permutation against 3.04 us on the same box (profiles/r03_keccak_split_half_microbench_ab.log) -- a lone wave is bound by its
DEPENDENT chain (~24 steps + one LDS trip per round either way, ~8 cycles per step), not by its instruction count -- and its
per-lane constant tables (two global loads in front of the first instruction) made the one-block tail kernel 1.8 us slower.
Not shipped; the model stays as the record of the layout.  python3 tools/keccak_lane_model.py -> tables + 'model ok'."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref  # noqa: E402

U = np.uint32
RHO = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]   # index x + 5y


def lane_of(x, y, h):   # word (x, y), half h (0 = low 32 bits, 1 = high)
    return 32 * h + 16 * (1 if y >= 3 else 0) + 5 * (y % 3) + x


ZERO_LANE = 15   # an unused lane that always holds zero


# ---- wave primitives ----
def dpp(old, src, kind, n, bound_ctrl):
    out = old.copy()
    for i in range(64):
        row, pos = i // 16, i % 16
        s = pos + n if kind == "shl" else pos - n
        if 0 <= s < 16:
            out[i] = src[16 * row + s]
        elif bound_ctrl:
            out[i] = 0
    return out


def permlane16_swap(a, b):
    ra, rb = a.copy(), b.copy()
    for r in (0, 2):
        ra[16 * (r + 1):16 * (r + 2)] = b[16 * r:16 * (r + 1)]
        rb[16 * r:16 * (r + 1)] = a[16 * (r + 1):16 * (r + 2)]
    return ra, rb


def permlane32_swap(a, b):
    ra, rb = a.copy(), b.copy()
    ra[32:] = b[:32]
    rb[:32] = a[32:]
    return ra, rb


def alignbit(hi, lo, sh):   # ({hi, lo} >> sh) low 32 bits, per-lane shift
    v = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    return ((v >> sh.astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(U)


def bpermute(src_lane, v):
    return v[src_lane]


# ---- per-lane constants ----
def constants():
    act = np.zeros(64, dtype=U)
    is_lo = np.array([1 if i < 32 else 0 for i in range(64)], dtype=bool)
    rot = np.zeros(64, dtype=U)          # right-rotate amount mod 32
    src = np.full((3, 64), ZERO_LANE, dtype=np.int64)
    for h in range(2):
        for y in range(5):
            for x in range(5):
                ln = lane_of(x, y, h)
                act[ln] = 0xFFFFFFFF
                q = (64 - RHO[x + 5 * y]) % 64        # rotl by rho == rotr by q
                rot[ln] = q % 32
    # pi + chi: destination (X, Y, h) needs B[X+d][Y], d = 0, 1, 2; B[X'][Y] = rot(A[x'][y']) with x' = (X' + 3Y) % 5, y' = X'.
    # A word rotated by q >= 32 sits with its halves swapped (the lane of half g computed half g ^ 1 of the result).
    for h in range(2):
        for Y in range(5):
            for X in range(5):
                for d in range(3):
                    Xp = (X + d) % 5
                    xs, ys = (Xp + 3 * Y) % 5, Xp
                    q = (64 - RHO[xs + 5 * ys]) % 64
                    swap = 1 if q >= 32 else 0
                    src[d][lane_of(X, Y, h)] = lane_of(xs, ys, h ^ swap)
    return act, is_lo, rot, src


def rc_lanes():
    k = np.zeros((24, 64), dtype=U)
    for r in range(24):
        k[r][lane_of(0, 0, 0)] = pyref._RC[r] & 0xFFFFFFFF
        k[r][lane_of(0, 0, 1)] = pyref._RC[r] >> 32
    return k


def keccak_f_lanes(a):
    act, is_lo, rot, src = constants()
    K = rc_lanes()
    z = np.zeros(64, dtype=U)
    for r in range(24):
        # theta: column parity.  Same-x lanes of a row sit 5 apart; rows 0/1 (2/3) hold planes 0-2 / 3-4 of a half
        p1 = dpp(z, a, "shl", 5, True) ^ a
        p2 = dpp(z, a, "shl", 10, True) ^ p1
        s0, s1 = permlane16_swap(p2, p2.copy())
        c = s0 ^ s1                                            # pos 0..4 of every row: C[x] of the row's half
        t = dpp(z, c, "shl", 4, True)
        cm = dpp(t, c, "shr", 1, False)                        # C[x-1] (pos 0 keeps C[4])
        t = dpp(z, c, "shl", 1, True)
        cp = dpp(t, c, "shr", 4, False)                        # C[x+1] (pos 4 takes C[0])
        w0, w1 = permlane32_swap(cp, cp.copy())
        partner = np.where(is_lo, w1, w0)                      # the other half's C[x+1]
        d = cm ^ alignbit(cp, partner, np.full(64, 31, dtype=U))   # rotl64(C[x+1], 1), my half
        d1 = dpp(d, d, "shr", 5, False)
        d2 = dpp(d1, d, "shr", 10, False)
        a = a ^ (d2 & act)
        # rho: right-rotate by q (mod 32 here; q >= 32 leaves the halves swapped, which pi's source lanes account for)
        w0, w1 = permlane32_swap(a, a.copy())
        partner = np.where(is_lo, w1, w0)
        b = alignbit(partner, a, rot)
        # pi + chi: three fetches per lane
        f0, f1, f2 = bpermute(src[0], b), bpermute(src[1], b), bpermute(src[2], b)
        a = f0 ^ (~f1 & f2)
        a = a ^ K[r]
    return a


def to_lanes(A):
    v = np.zeros(64, dtype=U)
    for y in range(5):
        for x in range(5):
            v[lane_of(x, y, 0)] = A[x][y] & 0xFFFFFFFF
            v[lane_of(x, y, 1)] = A[x][y] >> 32
    return v


def from_lanes(v):
    return [[int(v[lane_of(x, y, 0)]) | (int(v[lane_of(x, y, 1)]) << 32) for y in range(5)] for x in range(5)]


if __name__ == "__main__":
    rng = random.Random(1)
    for trial in range(20):
        A = [[rng.getrandbits(64) if trial else 0 for _ in range(5)] for _ in range(5)]
        want = pyref._keccak_f([row[:] for row in A])
        got = from_lanes(keccak_f_lanes(to_lanes(A)))
        assert got == want, f"trial {trial}"
        v = keccak_f_lanes(to_lanes(A))
        used = {lane_of(x, y, h) for x in range(5) for y in range(5) for h in range(2)}
        assert all(v[i] == 0 for i in range(64) if i not in used), "unused lanes must stay zero"
    act, is_lo, rot, src = constants()
    print("rot  ", list(map(int, rot)))
    for d in range(3):
        print(f"src{d} ", list(map(int, src[d])))
    print("model ok")
