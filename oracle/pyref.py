"""pyref -- independent Python big-int model of the reference path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

A second, structurally different restatement (canonical Python ints, no Montgomery form, its own Keccak)
of the same reference lines as oracle/zk_oracle.c, used to cross-check the C oracle on small cases.
Only tests/ may import it.  Pure-Python loops: keep inputs small (<= ~2^12 elements).

Reference lines followed (paths relative to the reference checkout):
  polynomial/src/multilinear/pairing_index.rs:2-26, evaluation_form.rs:15-103,
  polynomial/src/product_poly.rs:14-88, sumcheck/src/prover.rs:15-73, sumcheck/src/lib.rs:23-29,
  sumcheck/src/verifier.rs:15-78, polynomial/src/univariate_poly.rs:29-80,
  transcript/src/lib.rs:9-34, fft/src/lib.rs:4-61.
"""

FIELDS = {
    # id: (name, p, generator, two_adicity)   -- ark-ff 0.5.0 / ark-curves conventions
    0: ("bn254_fr", 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 5, 28),
    1: ("bls12_381_fr", 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 7, 32),
    2: ("bls12_377_fr", 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001, 22, 47),
}
R = 1 << 256


def modulus(field):
    return FIELDS[field][1]


def to_mont_limbs(field, x):
    """canonical int -> 4 LE u64 limbs of x*R mod p (ark-ff in-memory layout)."""
    v = (x % modulus(field)) * R % modulus(field)
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_mont_limbs(field, limbs):
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return v * pow(R, -1, modulus(field)) % modulus(field)


# ---- pairing_index.rs ----
def mask(n):  # :24-26
    return (1 << n) - 1


def insert_bit(val, index, bit):  # :16-20
    high = val >> index
    low = val & mask(index)
    return high << (index + 1) | bit << index | low


def index_pair(n_vars, index):  # :2-9
    base = n_vars - 1
    if base < 0 or base - index < 0:
        raise OverflowError("u8 underflow (reference panics)")
    out = []
    for val in range(1 << base):
        l = insert_bit(val, base - index, 0)
        out.append((l, l | (1 << (base - index))))
    return out


# ---- evaluation_form.rs ----
class MLE:
    def __init__(self, field, n_vars, evals):  # :15-27
        if len(evals) != (1 << n_vars):
            raise ValueError("evaluation vec len should equal 2^n_vars")
        self.field, self.n_vars, self.evals = field, n_vars, [e % modulus(field) for e in evals]

    def partial_evaluate(self, initial_var, assignments):  # :40-80
        p = modulus(self.field)
        new = list(self.evals)
        for i, a in enumerate(assignments):
            for j, (lp, rp) in enumerate(index_pair(self.n_vars - i, initial_var)):
                left, right = new[lp], new[rp]
                new[j] = (left - a * (left - right)) % p
        nv = self.n_vars - len(assignments)
        if nv < 0:
            raise OverflowError("usize underflow (reference panics)")
        return MLE(self.field, nv, new[: 1 << nv])

    def evaluate(self, point):  # :83-89
        if len(point) != self.n_vars:
            raise ValueError("evaluate must assign to all variables")
        return self.partial_evaluate(0, point).evals[0]

    def to_bytes(self):  # :97-103
        return b"".join(e.to_bytes(32, "big") for e in self.evals)


# ---- coefficient_form.rs:340-347 (to_evaluation_form) over boolean_hypercube.rs:27-45 ----
def coeff_to_evaluation(field, n_vars, terms):
    """terms: {key: coeff}, key bit v <-> variable v (selector_to_index, coefficient_form.rs:418-430)."""
    pm = modulus(field)
    out = []
    for idx in range(1 << n_vars if n_vars else 0):
        point = [int(ch) for ch in format(idx, "0%db" % n_vars)]          # binary_string(index, n): variable 0 first
        acc = 0
        for key, cf in terms.items():
            term = cf
            for v in range(n_vars):
                if (key >> v) & 1:
                    term = term * point[v] % pm
            acc = (acc + term) % pm
        out.append(acc)
    return out


# ---- product_poly.rs ----
class Product:
    def __init__(self, polys):  # :14-32
        if len(polys) == 0:
            raise ValueError("cannot create product polynomial from empty polynomials")
        if any(q.n_vars != polys[0].n_vars for q in polys):
            raise ValueError("cannot create product polynomial from polynomial that don't share the same number of variables")
        self.polys, self.n_vars, self.field = polys, polys[0].n_vars, polys[0].field

    def evaluate(self, point):  # :36-44
        if len(point) != self.n_vars:
            raise ValueError("evaluate must assign to all variables")
        out = 1
        for q in self.polys:
            out = out * q.evaluate(point) % modulus(self.field)
        return out

    def partial_evaluate(self, initial_var, assignments):  # :48-63
        return Product([q.partial_evaluate(initial_var, assignments) for q in self.polys])

    def prod_reduce(self):  # :66-74
        res = list(self.polys[0].evals)
        for q in self.polys[1:]:
            for i, e in enumerate(q.evals):
                res[i] = res[i] * e % modulus(self.field)
        return res

    def to_bytes(self):  # :77-83
        return b"".join(q.to_bytes() for q in self.polys)


# ---- Keccak-256 (sha3 0.10.8 Keccak256: pad 0x01, rate 136) ----
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
    0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
    0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
    0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def _keccak_f(A):
    # A[x][y] lanes; rotation offsets generated by the (x,y)->(y,2x+3y) walk instead of a table
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        x, y, cur = 1, 0, A[1][0]
        B[0][0] = A[0][0]
        for t in range(24):
            X, Y = y, (2 * x + 3 * y) % 5
            nxt = A[X][Y]
            B[X][Y] = _rol(cur, (t + 1) * (t + 2) // 2)
            x, y, cur = X, Y, nxt
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & _M64 & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    return A


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    padlen = rate - (len(msg) % rate)
    pad = bytearray(padlen)
    pad[0] ^= 0x01
    pad[-1] ^= 0x80
    msg += pad
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i : off + 8 * i + 8], "little")
        A = _keccak_f(A)
    return b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


# ---- transcript/src/lib.rs ----
class Transcript:
    def __init__(self):  # :10-14
        self.buf = bytearray()

    def append(self, data: bytes):  # :16-18
        self.buf += data

    def sample_challenge(self) -> bytes:  # :20-25
        h = keccak256(bytes(self.buf))
        self.buf = bytearray(h)
        return h

    def sample_field_element(self, field):  # :27-30
        return int.from_bytes(self.sample_challenge(), "big") % modulus(field)


# ---- sumcheck/src/prover.rs ----
def sumcheck_prove(poly: Product, claimed_sum, D, absorb_table):  # :15-73
    field = poly.field
    p = modulus(field)
    tr = Transcript()
    if absorb_table:
        tr.append(poly.to_bytes())  # :17
    tr.append((claimed_sum % p).to_bytes(32, "big"))  # :42
    round_polys, challenges = [], []
    for _ in range(poly.n_vars):  # :44
        rp = []
        for i in range(D + 1):  # :49
            rp.append(sum(poly.partial_evaluate(0, [i % p]).prod_reduce()) % p)
        tr.append(b"".join(v.to_bytes(32, "big") for v in rp))  # :59
        c = tr.sample_field_element(field)  # :62
        poly = poly.partial_evaluate(0, [c])  # :64
        round_polys.append(rp)
        challenges.append(c)
    return round_polys, challenges


# ---- univariate_poly.rs (verifier's needs) ----
def _interp_eval(field, ys, x):
    """value at x of the unique degree < len(ys) polynomial through (i, ys[i]) -- Lagrange, exact."""
    p = modulus(field)
    n = len(ys)
    acc = 0
    for i in range(n):
        num, den = 1, 1
        for j in range(n):
            if j != i:
                num = num * (x - j) % p
                den = den * (i - j) % p
        acc = (acc + ys[i] * num * pow(den, -1, p)) % p
    return acc


# ---- sumcheck/src/verifier.rs ----
def sumcheck_verify_partial(field, claimed_sum, round_polys, table_bytes=None):  # :38-41, :44-78
    p = modulus(field)
    tr = Transcript()
    if table_bytes is not None:
        tr.append(table_bytes)
    tr.append((claimed_sum % p).to_bytes(32, "big"))
    claim = claimed_sum % p
    challenges = []
    for rp in round_polys:
        tr.append(b"".join(v.to_bytes(32, "big") for v in rp))
        if claim != (_interp_eval(field, rp, 0) + _interp_eval(field, rp, 1)) % p:
            raise ValueError("verifier check failed: claimed_sum != p(0) + p(1)")
        c = tr.sample_field_element(field)
        claim = _interp_eval(field, rp, c)
        challenges.append(c)
    return claim, challenges


def sumcheck_verify(poly: Product, claimed_sum, round_polys):  # :15-33
    if len(round_polys) != poly.n_vars:
        raise ValueError("invalid proof: require 1 round poly for each variable in poly")
    claim, challenges = sumcheck_verify_partial(poly.field, claimed_sum, round_polys, poly.to_bytes())
    return poly.evaluate(challenges) == claim


# ---- fft/src/lib.rs ----
def root_of_unity(field, n):
    _, p, g, s = FIELDS[field]
    if n == 0 or n & (n - 1) or n.bit_length() - 1 > s:
        return None
    return pow(g, (p - 1) // n, p)


def fft_internal(field, values, omega):  # :21-46
    p = modulus(field)
    n = len(values)
    if n == 1:
        return list(values)
    if n & (n - 1):
        raise ValueError("values must be a power of 2")
    even = fft_internal(field, values[0::2], omega * omega % p)
    odd = fft_internal(field, values[1::2], omega * omega % p)
    out = [0] * n
    for i in range(n // 2):
        out[i] = (even[i] + pow(omega, i, p) * odd[i]) % p
        out[i + n // 2] = (even[i] + pow(omega, i + n // 2, p) * odd[i]) % p
    return out


def fft(field, coeffs):  # :4-8
    w = root_of_unity(field, len(coeffs))
    if w is None:
        raise ValueError("no root of unity (reference unwraps None)")
    return fft_internal(field, coeffs, w)


def ifft(field, evals):  # :11-19
    p = modulus(field)
    n = len(evals)
    w = root_of_unity(field, n)
    if w is None:
        raise ValueError("no root of unity (reference unwraps None)")
    ninv = pow(n, -1, p)
    return [v * ninv % p for v in fft_internal(field, evals, pow(w, -1, p))]


def dft_naive(field, values):
    """O(n^2) definition out[i] = sum_j in[j] * omega^(i*j) -- a third opinion for tiny n."""
    p = modulus(field)
    n = len(values)
    w = root_of_unity(field, n)
    return [sum(values[j] * pow(w, i * j, p) for j in range(n)) % p for i in range(n)]


# ---- synthetic inputs (same generator as orc_fill_random / the device kernel) ----
def _splitmix64(x):
    z = (x + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def random_element(field, seed, index):
    p = modulus(field)
    bits = p.bit_length()
    h0 = _splitmix64(seed ^ _splitmix64(index))
    attempt = 0
    while True:
        limbs = [_splitmix64((h0 + 4 * attempt + j) & _M64) for j in range(4)]
        v = sum(l << (64 * j) for j, l in enumerate(limbs)) & ((1 << bits) - 1)
        if v < p:
            return v
        attempt += 1
