import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, zk_amd
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
tr = zk_amd.Transcript(); tr.append(b"x"); r = tr.sample_field_element(field)
for n in (24, 23, 22, 21):
    t = zk_amd.MultiLinearPolynomial.random(ctx, n, 1, 0); o = zk_amd.MultiLinearPolynomial.alloc(ctx, n - 1)
    for _ in range(50): t.fold_into(r, o)
    ctx.synchronize()
    t0 = time.perf_counter(); s = t.bench_fold_samples(r, o, 2000); ctx.synchronize(); dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(2000): t.fold_into(r, o)
    ctx.synchronize(); dt2 = time.perf_counter() - t0
    print(f"n={n}: samples loop {dt/2000*1e6:.2f} us/step (kernel mean {s.mean()*1e3:.2f} median {np.median(s)*1e3:.2f} us); plain fold_into loop {dt2/2000*1e6:.2f} us/step", flush=True)
    t.free(); o.free()
