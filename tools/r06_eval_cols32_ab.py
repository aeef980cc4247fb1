"""evaluate's streaming kernel with 36 unsplit columns (ZK_EVAL_COLS32=1: four 32-bit data words x nine 29-bit weight limbs, no split29_half)
vs thirteen columns of 29-bit limbs (=0): whole-call wall clock (median / min of 41) and device time (40 back-to-back enqueues between two
HIP events) at n = 21..24, BN254 Fr and BLS12-381 Fr; arms in child processes, interleaved three times.  (Measured and not kept:
profiles/r06_eval_cols32_variant.patch applies the variant.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import zk_amd

    out = []
    for field, name in ((zk_amd.BN254_FR, "bn254"), (zk_amd.BLS12_381_FR, "bls381")):
        ctx = zk_amd.Context(field, 0)
        tr = zk_amd.Transcript()
        tr.append(b"pt")
        for n in (21, 22, 23, 24):
            t = zk_amd.MultiLinearPolynomial.random(ctx, n, 3, 0)
            pt = tr.sample_n_field_elements(field, n)
            t.evaluate(pt)
            ms = sorted(zk_amd.bench_evaluate(t, pt, reps=41))
            dev = zk_amd.bench_evaluate_device(t, pt, reps=40)
            out.append("%s n%d %.1f/%.1f dev %.1f" % (name, n, ms[20] * 1e3, ms[0] * 1e3, dev * 1e3))
            t.free()
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("ZK_AB_CHILD"):
        child()
        sys.exit(0)
    for rep in range(3):
        for arm in ("0", "1"):
            e = dict(os.environ, ZK_EVAL_COLS32=arm, ZK_AB_CHILD="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=600)
            print("[ZK_EVAL_COLS32=%s] %s" % (arm, r.stdout.strip() or r.stderr[-2000:]), flush=True)
