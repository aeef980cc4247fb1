"""evaluate (evaluation_form.rs:83-89; the reference's criterion bench, polynomial/benches/polynomial_evaluation.rs:85-105) timed
inside the library (zk_bench_evaluate: std::chrono around the whole call) at n = 18..24, on BN254 Fr and BLS12-381 Fr (the bench's
own field).  Run as `python tools/ab_evaluate.py`: it re-runs itself in child processes with ZK_EVAL_STREAM_MIN set per arm
(99 = k_eval_low everywhere, the round-3 path; unset = the shipped threshold), interleaved, same box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np

    import zk_amd

    for field, name in ((zk_amd.BN254_FR, "bn254"), (zk_amd.BLS12_381_FR, "bls12_381")):
        ctx = zk_amd.Context(field, 0)
        tr = zk_amd.Transcript()
        tr.append(b"pt")
        for n in (18, 19, 20, 21, 22, 23, 24):
            t = zk_amd.MultiLinearPolynomial.random(ctx, n, 3, 0)
            pt = tr.sample_n_field_elements(field, n)
            t.evaluate(pt)
            ms = sorted(zk_amd.bench_evaluate(t, pt, reps=41))
            print("%s n=%d median %.1f us min %.1f us" % (name, n, ms[20] * 1e3, ms[0] * 1e3), flush=True)
            t.free()


if __name__ == "__main__":
    if os.environ.get("ZK_AB_CHILD"):
        child()
        sys.exit(0)
    arms = [("k_eval_low only (ZK_EVAL_STREAM_MIN=99)", {"ZK_EVAL_STREAM_MIN": "99"}), ("shipped", {}),
            ("stream from 19 (ZK_EVAL_STREAM_MIN=19)", {"ZK_EVAL_STREAM_MIN": "19"})]
    for rep in range(2):
        for label, env in arms:
            e = {k: v for k, v in os.environ.items() if k != "ZK_EVAL_STREAM_MIN"}
            e.update(env, ZK_AB_CHILD="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=600)
            print("== %s (pass %d)" % (label, rep))
            print(r.stdout.strip() or r.stderr[-2000:], flush=True)
