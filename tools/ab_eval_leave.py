"""A/B of how many variables k_eval_stream leaves (= log2 of its grid): ZK_EVAL_STREAM_LEAVE = 9 (512 workgroups), 10, 11; child processes."""
import os, subprocess, sys
child = '''
import sys; sys.path.insert(0, %r)
import zk_amd
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
tr = zk_amd.Transcript(); tr.append(b"pt")
for n in (21, 22, 23):
    t = zk_amd.MultiLinearPolynomial.random(ctx, n, 3, 0)
    pt = tr.sample_n_field_elements(zk_amd.BN254_FR, n)
    t.evaluate(pt)
    ms = sorted(zk_amd.bench_evaluate(t, pt, reps=41))
    dev = zk_amd.bench_evaluate_device(t, pt, 40) * 1e3
    print("n=%%d call median %%.1f us, device %%.1f us" %% (n, ms[20] * 1e3, dev), flush=True)
    t.free()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for leave in ("9", "8", "7"):
        e = dict(os.environ, ZK_EVAL_STREAM_LEAVE=leave)
        r = subprocess.run([sys.executable, "-c", child], env=e, capture_output=True, text=True, timeout=300)
        print("== ZK_EVAL_STREAM_LEAVE=%s (pass %d)" % (leave, rep)); print(r.stdout.strip() or r.stderr[-1500:], flush=True)
