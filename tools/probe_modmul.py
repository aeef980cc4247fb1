import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_amd
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
for v in (0, 1, 0, 1):
    print("variant", v, "%.3e modmul/s" % ctx.bench_modmul(3000, v))
