import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zk_amd
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
tr = zk_amd.Transcript(); tr.append(b"pt")
for n in (18, 19, 20, 21, 24):
    t = zk_amd.MultiLinearPolynomial.random(ctx, n, 3, 0)
    pt = tr.sample_n_field_elements(field, n)
    t.evaluate(pt)
    ts = []
    for _ in range(20):
        ctx.synchronize(); t0 = time.perf_counter(); t.evaluate(pt); ts.append(time.perf_counter() - t0)
    print("evaluate n=%d: median %.1f us  min %.1f us" % (n, sorted(ts)[10] * 1e6, min(ts) * 1e6))
    t.free()
