import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zk_amd
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
tr = zk_amd.Transcript(); tr.append(b"pt")
for n in (4, 8, 12, 13, 16, 18):
    t = zk_amd.MultiLinearPolynomial.random(ctx, n, 3, 0)
    pt = tr.sample_n_field_elements(field, n)
    t.evaluate(pt)
    ts = []
    for _ in range(50):
        ctx.synchronize(); t0 = time.perf_counter(); t.evaluate(pt); ts.append(time.perf_counter() - t0)
    print("evaluate n=%d: median %.1f us  min %.1f us" % (n, sorted(ts)[25] * 1e6, min(ts) * 1e6))
    t.free()
