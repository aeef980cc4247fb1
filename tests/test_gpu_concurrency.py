"""The C ABI's threading contract (include/zk_amd.h:18-19): a zk_ctx is not thread-safe, DISTINCT contexts are
independent -- the reference's prover / verifier hold no global state (sumcheck/src/prover.rs:15-30 builds its Transcript per
call).  Two host threads, one context each (ctypes releases the GIL around every library call, so the calls really overlap),
loop the prover, evaluate and the NTT for a few seconds; every result must equal the oracle's, bit for bit.  Then two contexts
share ONE caller-owned stream (zk_ctx_set_stream) and are interleaved call by call on one thread."""
import threading
import time

import numpy as np
import pytest

import zk_amd
from oracle import binding as orc
from zk_amd import MultiLinearPolynomial as MLE
from zk_amd import ProductPoly, SumcheckProver

pytestmark = pytest.mark.gpu


def _workload(field, seed):
    """inputs + oracle answers for one thread: n = 16 prover (k = 2, D = 2), n = 18 evaluate, 2^14-point NTT"""
    n = 16
    tabs = [orc.fill_random(field, seed + f, 1 << n) for f in range(2)]
    claimed = orc.sum_elems(field, orc.prod_reduce(field, n, tabs))   # iter().sum::<F>()
    rp, ch = orc.sumcheck_prove(field, n, tabs, 2, claimed, False)
    ev_tab = orc.fill_random(field, seed + 7, 1 << 18)
    ev_pt = orc.fill_random(field, seed + 8, 18)
    vec = orc.fill_random(field, seed + 9, 1 << 14)
    return {"field": field, "n": n, "tabs": tabs, "claimed": claimed, "rp": rp, "ch": ch, "ev_tab": ev_tab, "ev_pt": ev_pt,
            "ev": orc.mle_evaluate(field, 18, ev_tab, ev_pt), "vec": vec, "fft": orc.ntt_fast(field, vec)}


def _one_pass(ctx, w, state):
    """one prover + one evaluate + one NTT on ctx; tables are uploaded once per context and kept in `state`"""
    if "polys" not in state:
        state["polys"] = [MLE.new(ctx, w["n"], t) for t in w["tabs"]]
        state["ev"] = MLE.new(ctx, 18, w["ev_tab"])
    proof, ch = SumcheckProver(2).prove_partial(ProductPoly.new(state["polys"]), w["claimed"])
    assert np.array_equal(proof.round_polys, w["rp"]) and np.array_equal(ch, w["ch"]), "prover"
    assert np.array_equal(state["ev"].evaluate(w["ev_pt"]), w["ev"]), "evaluate"
    assert np.array_equal(zk_amd.fft(ctx, w["vec"]), w["fft"]), "fft"


def test_two_threads_one_context_each():
    loads = [_workload(zk_amd.BN254_FR, 9100), _workload(zk_amd.BLS12_381_FR, 9200)]
    errors, counts = [], [0, 0]
    start = threading.Barrier(2)

    def run(i):
        try:
            ctx = zk_amd.Context(loads[i]["field"], 0)
            state = {}
            _one_pass(ctx, loads[i], state)   # first call allocates: keep it out of the overlapped loop's timing
            start.wait()
            t_end = time.time() + 4.0
            while time.time() < t_end:
                _one_pass(ctx, loads[i], state)
                counts[i] += 1
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors
    assert all(not t.is_alive() for t in ts)
    assert min(counts) >= 20, counts   # both threads made progress side by side (a pass is ~1 ms of GPU work)


def test_two_contexts_interleaved_on_one_caller_stream():
    import torch

    stream = torch.cuda.Stream(device=0)
    loads = [_workload(zk_amd.BN254_FR, 9300), _workload(zk_amd.BLS12_377_FR, 9400)]
    ctxs = [zk_amd.Context(w["field"], 0) for w in loads]
    for c in ctxs:
        c.set_stream(stream.cuda_stream)
    states = [{}, {}]
    for _ in range(10):
        for c, w, s in zip(ctxs, loads, states):
            _one_pass(c, w, s)
    # and back on their own streams
    for c in ctxs:
        c.use_own_stream()
    for c, w, s in zip(ctxs, loads, states):
        _one_pass(c, w, s)


def test_two_threads_batching_on_their_own_contexts():
    """zk_sumcheck_prove_batch keeps its launch recorder per THREAD (launch.hpp: thread_local): two host threads, a context each, batch
    different proofs at the same time -- every proof equals the oracle's, and each thread's zk_batch_last_stats is its own."""
    fields = [zk_amd.BN254_FR, zk_amd.BLS12_381_FR]
    shapes = [(2, 2, 13, 5), (3, 3, 12, 3)]   # (k, D, n, B)
    work = []
    for f, (k, D, n, B) in zip(fields, shapes):
        cases = []
        for b in range(B):
            tabs = [orc.fill_random(f, 9500 + 16 * b + j, 1 << n) for j in range(k)]
            s = orc.sum_elems(f, orc.prod_reduce(f, n, tabs))
            cases.append((tabs, s, orc.sumcheck_prove(f, n, tabs, D, s, False)))
        work.append(cases)
    errors, stats = [], [None, None]
    start = threading.Barrier(2)

    def run(i):
        try:
            f, (k, D, n, B) = fields[i], shapes[i]
            ctx = zk_amd.Context(f, 0)
            polys = [ProductPoly.new([MLE.new(ctx, n, t) for t in tabs]) for tabs, _, _ in work[i]]
            sums = np.stack([s for _, s, _ in work[i]])
            SumcheckProver(D).prove_partial_batch(polys, sums)   # first call allocates
            start.wait()
            for _ in range(40):
                got = SumcheckProver(D).prove_partial_batch(polys, sums)
                for (proof, ch), (_, _, (want_rp, want_ch)) in zip(got, work[i]):
                    assert np.array_equal(proof.round_polys, want_rp) and np.array_equal(ch, want_ch), "batched proof"
            stats[i] = zk_amd.batch_last_stats()
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=180)
    assert not errors, errors
    assert all(not t.is_alive() for t in ts)
    assert stats[0] is not None and stats[1] is not None and stats[0][0] > 0 and stats[1][0] > 0 and stats[0][1] == 0 and stats[1][1] == 0, stats
