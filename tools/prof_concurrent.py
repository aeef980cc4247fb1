"""B independent proofs in flight with what exists: B contexts (one stream each) on B host threads, each proving
its own ProductPoly -- aggregate wall clock for `total` proofs and per-proof latency, for (k, D, n) shapes and for evaluate.
ctypes releases the GIL around every library call, so the enqueues of different threads overlap.

usage: python tools/prof_concurrent.py [total_proofs=8] [reps=7]
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import zk_amd  # noqa: E402
from zk_amd import MultiLinearPolynomial as MLE  # noqa: E402

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
field = zk_amd.BN254_FR
claimed = zk_amd.fe_from_int(field, 7)   # timing only


def run_shape(k, D, n, B, what="prove"):
    ctxs = [zk_amd.Context(field, 0) for _ in range(B)]
    per = TOTAL // B
    work = []
    for b, c in enumerate(ctxs):
        items = []
        for j in range(per):
            polys = [MLE.random(c, n, 0x6000 + 64 * b + 8 * j + f, 0) for f in range(k)]
            items.append(zk_amd.ProductPoly.new(polys))
        work.append(items)
    prover = zk_amd.SumcheckProver(D)
    pt = np.stack([zk_amd.fe_from_int(field, 3 + i) for i in range(n)])

    def one(pp):
        if what == "prove":
            prover.prove_partial(pp, claimed)
        else:
            pp.polynomials[0].evaluate(pt)

    for items in work:
        for pp in items:
            one(pp)
    walls, lats = [], []
    for _ in range(REPS):
        bar = threading.Barrier(B + 1)
        lat = [[] for _ in range(B)]

        def th(b):
            bar.wait()
            for pp in work[b]:
                t = time.perf_counter()
                one(pp)
                lat[b].append(time.perf_counter() - t)
            bar.wait()

        ts = [threading.Thread(target=th, args=(b,)) for b in range(B)]
        for t in ts:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        walls.append(time.perf_counter() - t0)
        for t in ts:
            t.join()
        lats.append(float(np.median([x for l in lat for x in l])))
    for items in work:
        for pp in items:
            for q in pp.polynomials:
                q.free()
    for c in ctxs:
        c.close()
    return float(np.median(walls)) * 1e3, float(np.min(walls)) * 1e3, float(np.median(lats)) * 1e3


for what, k, D, n in (("prove", 3, 3, 20), ("prove", 2, 2, 20), ("prove", 2, 2, 16), ("evaluate", 1, 1, 20)):
    base = None
    for B in (1, 2, 4, 8):
        if B > TOTAL:
            continue
        med, mn, lat = run_shape(k, D, n, B, what)
        base = base or med
        print(f"{what} k={k} D={D} n={n}: {TOTAL} calls on {B} contexts/threads: wall median {med:.3f} ms (min {mn:.3f}), "
              f"per-call latency {lat:.3f} ms, {TOTAL / med * 1e3:.0f} calls/s, x{base / med:.2f} vs B=1", flush=True)
