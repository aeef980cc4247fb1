"""Config 4 as SURVEY 8d words it -- B independent (k, D = k, n) prove_partial calls -- back to back against ONE zk_sumcheck_prove_batch.
usage: python tools/prof_batch.py [n=20] [reps=9]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import zk_amd  # noqa: E402
from zk_amd import MultiLinearPolynomial as MLE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
claimed = zk_amd.fe_from_int(field, 7)   # timing only
for k in (3, 2):
    prover = zk_amd.SumcheckProver(k)
    layers = [zk_amd.ProductPoly.new([MLE.random(ctx, n, 0x6000 + 16 * layer + f, 0) for f in range(k)]) for layer in range(8)]
    for B in (1, 2, 4, 8):
        sums = np.stack([claimed] * B)
        for pp in layers[:B]:
            prover.prove_partial(pp, claimed)
        prover.prove_partial_batch(layers[:B], sums)
        seq, bat = [], []
        for _ in range(reps):
            ctx.synchronize()
            t = time.perf_counter()
            for pp in layers[:B]:
                prover.prove_partial(pp, claimed)
            seq.append(time.perf_counter() - t)
            ctx.synchronize()
            t = time.perf_counter()
            got = prover.prove_partial_batch(layers[:B], sums)
            bat.append(time.perf_counter() - t)
        m, r = zk_amd.batch_last_stats()
        one = prover.prove_partial(layers[B - 1], claimed)
        same = np.array_equal(one[0].round_polys, got[B - 1][0].round_polys) and np.array_equal(one[1], got[B - 1][1])
        print(f"k={k} D={k} n={n} B={B}: back to back {np.median(seq) * 1e3:.3f} ms (min {min(seq) * 1e3:.3f}), batched {np.median(bat) * 1e3:.3f} ms "
              f"(min {min(bat) * 1e3:.3f}) = x{np.median(seq) / np.median(bat):.2f}; launches merged {m} replayed {r}; last proof identical: {same}", flush=True)
    for pp in layers:
        for q in pp.polynomials:
            q.free()
