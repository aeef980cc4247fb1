import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zk_amd
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
t = zk_amd.Transcript(); buf = os.urandom(64 << 20)
for _ in range(2):
    t0 = time.perf_counter(); t.append(buf); dt = time.perf_counter() - t0
print("host absorb %.3f GB/s" % (len(buf) / dt / 1e9))
for n in (20, 24):
    A = zk_amd.MultiLinearPolynomial.random(ctx, n, 1, 0); B = zk_amd.MultiLinearPolynomial.random(ctx, n, 1, 1 << n)
    pp = zk_amd.ProductPoly.new([A, B])
    s = pp.round_sums(1)
    claimed = zk_amd.fe_from_int(field, (zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1])) % zk_amd.modulus(field))
    pr = zk_amd.SumcheckProver(2)
    for _ in range(2):
        t0 = time.perf_counter(); pr.prove(pp, claimed); dt = time.perf_counter() - t0
    print("prove (absorbing) n=%d: %.1f ms" % (n, dt * 1e3))
