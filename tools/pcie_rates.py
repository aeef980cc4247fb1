import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import zk_amd
from zk_amd import MultiLinearPolynomial as MLE
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
for n in (20, 24):
    host = np.zeros((1 << n, 4), dtype=np.uint64); host[:, 0] = np.arange(1 << n, dtype=np.uint64)
    t = MLE.new(ctx, n, host); t.free()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); t = MLE.new(ctx, n, host); ctx.synchronize(); ts.append(time.perf_counter() - t0); 
        if _ < 2: t.free()
    up = min(ts)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = t.evaluation_slice(); ts.append(time.perf_counter() - t0)
    dn = min(ts)
    b = (32 << n) / 1e9
    print(f"n={n}: upload {up*1e3:.2f} ms = {b/up:.1f} GB/s; download {dn*1e3:.2f} ms = {b/dn:.1f} GB/s")
    t.free()
# download into a buffer whose pages are already mapped (what a caller reusing its Vec<F> sees)
from zk_amd._lib import lib, check, u64p
for n in (20, 24):
    t = MLE.random(ctx, n, 1, 0)
    out = np.ones((1 << n, 4), dtype=np.uint64)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); check(lib.zk_mle_download(ctx._h, t._h, out.ctypes.data_as(u64p))); ts.append(time.perf_counter() - t0)
    b = (32 << n) / 1e9
    print(f"n={n}: download into mapped pages {min(ts)*1e3:.2f} ms = {b/min(ts):.1f} GB/s")
    t.free()
