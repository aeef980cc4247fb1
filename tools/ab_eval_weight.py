import os, subprocess, sys
child = '''
import sys; sys.path.insert(0, %r)
import zk_amd
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
tr = zk_amd.Transcript(); tr.append(b"pt")
for n in (20, 21, 22, 24):
    t = zk_amd.MultiLinearPolynomial.random(ctx, n, 3, 0)
    pt = tr.sample_n_field_elements(zk_amd.BN254_FR, n)
    t.evaluate(pt)
    ms = sorted(zk_amd.bench_evaluate(t, pt, reps=41))
    dev = zk_amd.bench_evaluate_device(t, pt, 40) * 1e3
    print("n=%%d call median %%.1f us, device %%.1f us" %% (n, ms[20] * 1e3, dev), flush=True)
    t.free()
''' % os.getcwd()
for rep in range(2):
    for mode in ("0", "1", "3"):
        e = dict(os.environ, ZK_EVAL_WEIGHT=mode)
        r = subprocess.run([sys.executable, "-c", child], env=e, capture_output=True, text=True, timeout=300)
        print("== ZK_EVAL_WEIGHT=%s (pass %d)" % (mode, rep)); print(r.stdout.strip() or r.stderr[-1500:], flush=True)
