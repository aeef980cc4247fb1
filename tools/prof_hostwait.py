"""C-level wall clock of prove_partial (zk_bench_prove_partial) and Python-level evaluate at small sizes: what the host-side
launch / completion path costs.  python3 tools/prof_hostwait.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_amd
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
for n in (12, 16, 20, 24):
    polys = [zk_amd.MultiLinearPolynomial.random(ctx, n, 77 + n, f << n) for f in range(2)]
    pp = zk_amd.ProductPoly.new(polys)
    s = pp.round_sums(1)
    claimed = zk_amd.fe_from_int(field, zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1]))
    zk_amd.bench_prove_partial(pp, 2, claimed, 3)
    ms = sorted(zk_amd.bench_prove_partial(pp, 2, claimed, 21))
    print("prove_partial n=%d: median %.4f ms  min %.4f ms" % (n, ms[10], ms[0]))
    for q in polys: q.free()
tr = zk_amd.Transcript(); tr.append(b"pt")
for n in (4, 12, 18, 20):
    t = zk_amd.MultiLinearPolynomial.random(ctx, n, 3, 0)
    pt = tr.sample_n_field_elements(field, n)
    t.evaluate(pt)
    ts = []
    for _ in range(50):
        ctx.synchronize(); t0 = time.perf_counter(); t.evaluate(pt); ts.append(time.perf_counter() - t0)
    print("evaluate n=%d: median %.1f us  min %.1f us" % (n, sorted(ts)[25] * 1e6, min(ts) * 1e6))
    t.free()
