"""why did `evaluate` at n = 18..21 lose 5-11 % between the round-4 and round-5 driver runs?  Same-box A/B of the
round-4 library (ab_tmp/libzk_r04.so, built from ee4257d) against HEAD (ab_tmp/libzk_head.so) through bench.py's OWN row code --
10 warm-up calls, then zk_bench_evaluate (std::chrono around the whole zk_mle_evaluate inside the library), median of 11 (BN254) /
21 (BLS12-381) -- and through round 4's row code (no warm-up calls beyond the first).  Plain ctypes on the named .so (the two builds
have different ABI numbers; the five entry points used here did not change).  Child processes alternate, three passes.

usage: python tools/r06_eval_ab.py            (parent)
"""
import ctypes as c
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(path, warm):
    import torch  # noqa: F401  (one libamdhip64 in the process, as zk_amd/_lib.py does)
    import numpy as np

    lib = c.CDLL(path)
    u64p = c.POINTER(c.c_uint64)

    def chk(rc):
        assert rc == 0, rc

    for field, tag, reps in ((0, "bn254", 11), (1, "bls12_381", 21)):
        ctx = c.c_void_p()
        chk(lib.zk_ctx_create(field, 0, c.byref(ctx)))
        tr = c.c_void_p()
        chk(lib.zk_transcript_new(c.byref(tr)))
        chk(lib.zk_transcript_append(tr, b"zk_amd bench evaluate", 21))
        for n in (18, 19, 20, 21):
            t = c.c_void_p()
            chk(lib.zk_mle_alloc(ctx, c.c_uint64(n), c.byref(t)))
            chk(lib.zk_mle_fill_random(ctx, t, c.c_uint64(0x5EED0E00 + n), c.c_uint64(0)))
            pt = np.zeros((n, 4), dtype=np.uint64)
            chk(lib.zk_transcript_sample_n_field_elements(tr, field, c.c_uint64(n), pt.ctypes.data_as(u64p)))
            out = np.zeros(4, dtype=np.uint64)
            for _ in range(warm):
                chk(lib.zk_mle_evaluate(ctx, t, pt.ctypes.data_as(u64p), c.c_uint64(n), out.ctypes.data_as(u64p)))
            ms = (c.c_double * reps)()
            chk(lib.zk_bench_evaluate(ctx, t, pt.ctypes.data_as(u64p), c.c_uint64(n), reps, ms))
            v = sorted(ms)
            print(f"{tag} n={n}: median {v[reps // 2] * 1e3:.2f} us  min {v[0] * 1e3:.2f} us", flush=True)
            chk(lib.zk_mle_free(ctx, t))
        lib.zk_ctx_destroy(ctx)


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    for p in range(3):
        for name in ("r04", "head"):
            for warm in (10, 1):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.join(ROOT, "ab_tmp", f"libzk_{name}.so"), str(warm)],
                                   capture_output=True, text=True, timeout=600)
                print(f"== pass {p} {name} warm-up calls {warm}")
                print(r.stdout.strip() if r.returncode == 0 else r.stdout + r.stderr[-3000:], flush=True)
