"""bench.py times prod_reduce / partial_evaluate at 2^24 as single calls bracketed by a synchronise; this probe repeats those rows in one
process and idles in between: the first pass after idle seconds measures var23 / var0 / k = 3 up to 40 % longer than the passes behind it
(profiles/r06_a_rows_probe.log), which is why bench.py runs 0.8 s of untimed folds in front of them."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import zk_amd
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
n = 24
tabs = [zk_amd.MultiLinearPolynomial.random(ctx, n, 0x5EED0F00 + f, 0) for f in range(3)]
tr = zk_amd.Transcript(); tr.append(b"x")
asg = tr.sample_n_field_elements(field, 1)
def timed(fn, reps=9):
    ts = []
    for _ in range(reps):
        ctx.synchronize(); t1 = time.perf_counter(); keep = fn(); ctx.synchronize(); ts.append(time.perf_counter() - t1)
        if keep is not None: keep.free()
    return sorted(ts)[len(ts) // 2] * 1e6, min(ts) * 1e6
for rep in range(7):
    out = []
    for v in (1, 12, 23, 0):
        tabs[0].partial_evaluate(v, asg).free()
        out.append("var%d %.1f/%.1f" % ((v,) + timed(lambda: tabs[0].partial_evaluate(v, asg))))
    for k in (2, 3):
        pk = zk_amd.ProductPoly.new(tabs[:k]); pk.prod_reduce_device().free()
        out.append("k%d %.1f/%.1f" % ((k,) + timed(lambda: pk.prod_reduce_device())))
    print(" | ".join(out), flush=True)
    if rep == 3:
        host0 = tabs[0].evaluation_slice(); time.sleep(2.0); del host0   # what the gate does in between: a big download, seconds of CPU work
