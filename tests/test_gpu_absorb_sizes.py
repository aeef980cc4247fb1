"""`prove` (tables absorbed, prover.rs:15-20) and `to_bytes` (evaluation_form.rs:97-103, product_poly.rs:77-83) at the sizes where
their code path changes: from 2^19 elements a table crosses the device serialiser in 16-MiB chunks through two device + two pinned
buffers (capi.hip: stream_table_bytes), a host sponge (absorb_tables) or a pool of copy-out helpers (zk_mle_to_bytes) behind them.
Below 2^19 elements none of that machinery runs, so the small-size parity tests say nothing about it.

Every case is bit-compared with the oracle on the same seeded tables: whole proofs (every round polynomial), and every byte of the
serialisation.  bench.py publishes timings of exactly these calls; its parity gate checks them too (prove_absorbing_n20 / n24,
to_bytes_2p24).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import zk_amd
from oracle import binding as orc
from zk_amd import MultiLinearPolynomial as MLE
from zk_amd import ProductPoly, SumcheckProver, SumcheckVerifier

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ctx = {}


def ctx_for(field):
    if field not in _ctx:
        _ctx[field] = zk_amd.Context(field, 0)
    return _ctx[field]


def _absorbing_prove_vs_oracle(field, n, k, D, seed):
    """SumcheckProver::<D>::prove and SumcheckVerifier::verify (both absorb poly.to_bytes() first: prover.rs:17, verifier.rs:24)"""
    c = ctx_for(field)
    polys = [MLE.random(c, n, seed, f << n) for f in range(k)]
    tabs = [q.evaluation_slice() for q in polys]
    pp = ProductPoly.new(polys)
    s = pp.round_sums(1)
    claimed = orc.add(field, s[0], s[1])
    want_rp, want_ch = orc.sumcheck_prove(field, n, tabs, D, claimed, True)        # the faithful restatement, table bytes absorbed
    partial_rp, _ = orc.sumcheck_prove(field, n, tabs, D, claimed, False) if n <= 20 else (None, None)
    proof = SumcheckProver(D).prove(pp, claimed)
    assert np.array_equal(proof.round_polys, want_rp), "round polynomials differ from the oracle"
    if partial_rp is not None:   # the absorb changed every challenge: rounds >= 1 differ from prove_partial's
        assert np.array_equal(proof.round_polys[0], partial_rp[0]) and not np.array_equal(proof.round_polys[1], partial_rp[1])
    if n <= 20:   # prove() takes the polynomial by value in the reference; here the handles stay valid and intact
        for q, t in zip(polys, tabs):
            assert np.array_equal(q.evaluation_slice(), t)
    assert SumcheckVerifier.verify(pp, proof) is True                               # verifier.rs:15-33 (same chunked absorb)
    bad = proof.round_polys.copy()
    bad[n - 1, 0] = orc.add(field, bad[n - 1, 0], orc.from_int(field, 1))
    try:
        ok = SumcheckVerifier.verify(pp, zk_amd.SumcheckProof(claimed, bad))
    except zk_amd.ZkError:
        ok = False
    assert ok is False
    # the consuming variant (tables folded in place) absorbs the same bytes first
    proof2, ch2 = SumcheckProver(D)._run(pp, claimed, True, True)
    assert np.array_equal(proof2.round_polys, want_rp) and np.array_equal(ch2, want_ch)
    for q in polys:
        q.free()


@pytest.mark.parametrize("field,k,D", [(zk_amd.BN254_FR, 2, 2), (zk_amd.BN254_FR, 3, 3), (zk_amd.BLS12_381_FR, 2, 2)])
def test_prove_absorbing_n20_bit_exact(field, k, D):
    """config[1] "20-var MLE fold + full sumcheck": two 16-MiB chunks per table, so a table boundary falls inside the double-buffer
    rotation (k = 3: chunk 2 of table 0 and chunk 1 of table 1 share a buffer parity)."""
    _absorbing_prove_vs_oracle(field, 20, k, D, 0x5EED0000 + 20)


def test_prove_absorbing_n21_single_factor():
    """k = 1 (D = 1): four chunks of one table; the reference's own sumcheck test shape (sumcheck/src/lib.rs:53-64) at size"""
    _absorbing_prove_vs_oracle(zk_amd.BN254_FR, 21, 1, 1, 0x5EED0000 + 21)


def test_prove_absorbing_n24_bit_exact():
    """the metric's own size: 32 chunks per table, 1 GiB through the sponge"""
    _absorbing_prove_vs_oracle(zk_amd.BN254_FR, 24, 2, 2, 0x5EED0000 + 24)


@pytest.mark.parametrize("field,n", [(zk_amd.BN254_FR, 19), (zk_amd.BN254_FR, 20), (zk_amd.BLS12_381_FR, 21), (zk_amd.BN254_FR, 24)])
def test_to_bytes_chunked_sizes(field, n):
    """to_bytes at 1, 2, 4 and 32 chunks: into a fresh Vec-like destination (page faults taken by the copy-out helpers), into a
    pre-faulted one, and through the bytes-returning call; then ProductPoly::to_bytes (concatenation, product_poly.rs:77-83)."""
    c = ctx_for(field)
    t = MLE.random(c, n, 0x70B17E5 + n, 0)
    tab = t.evaluation_slice()
    want = np.frombuffer(orc.mle_to_bytes(field, n, tab), dtype=np.uint8)
    fresh = t.to_bytes_array()
    assert np.array_equal(fresh, want), "fresh destination"
    warm = np.full(32 << n, 0xA5, dtype=np.uint8)      # every page touched before the call
    got = t.to_bytes_array(out=warm)
    assert got is warm and np.array_equal(warm, want), "pre-faulted destination"
    odd = np.full((32 << n) + 64, 0x5A, dtype=np.uint8)   # a destination that is not page aligned
    view = odd[24:24 + (32 << n)]
    t.to_bytes_array(out=view)
    assert np.array_equal(view, want) and (odd[:24] == 0x5A).all() and (odd[24 + (32 << n):] == 0x5A).all(), "unaligned destination"
    if n <= 21:
        assert t.to_bytes() == want.tobytes()
        u = MLE.random(c, n, 0x70B17E5 + n, 1 << n)
        both = ProductPoly.new([t, u]).to_bytes()
        assert both[: 32 << n] == want.tobytes() and both[32 << n:] == orc.mle_to_bytes(field, n, u.evaluation_slice())
        u.free()
    t.free()


_CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import zk_amd
from oracle import binding as orc
from zk_amd import MultiLinearPolynomial as MLE
c = zk_amd.Context(zk_amd.BN254_FR, 0)
for n in (18, 20, 21, 24):
    t = MLE.random(c, n, 0x70B17E5 + n, 0)
    want = np.frombuffer(orc.mle_to_bytes(zk_amd.BN254_FR, n, t.evaluation_slice()), dtype=np.uint8)
    assert np.array_equal(t.to_bytes_array(), want), ("fresh", n)
    warm = np.zeros(32 << n, dtype=np.uint8)
    assert np.array_equal(t.to_bytes_array(out=warm), want), ("warm", n)
    t.free()
print("to_bytes threads ok")
"""


@pytest.mark.parametrize("threads", ["1", "3"])
def test_to_bytes_with_the_helper_pool_forced(threads):
    """ZK_TO_BYTES_THREADS is read once per process: child processes with the copy-out on the caller's thread only (1) and on three
    threads (an odd split of every 16-MiB chunk) serialise 2^18..2^24 elements against the oracle."""
    r = subprocess.run([sys.executable, "-c", _CHILD % ROOT], env=dict(os.environ, ZK_TO_BYTES_THREADS=threads), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "to_bytes threads ok" in r.stdout, r.stdout + r.stderr
