"""One-rank sharded prover through RCCL (zk_shard_prover_run, n = 24, k = 2, D = 2) for kernel traces: python3 tools/prof_shard.py [gather_below] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zk_amd
from zk_amd.distributed import GpuShardBackend, RcclComm
gb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
field = zk_amd.BN254_FR
ctx = zk_amd.Context(field, 0)
comm = RcclComm(ctx)
shards = [zk_amd.MultiLinearPolynomial.random(ctx, n, sd, 0) for sd in (0x5EED0100, 0x5EED0200)]
s = zk_amd.ProductPoly.new(shards).round_sums(1)
claimed = zk_amd.fe_from_int(field, zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1]))
ts = []
for _ in range(reps):
    pp = zk_amd.ProductPoly.new([q.clone() for q in shards])
    b = GpuShardBackend(pp, 2, claimed, 1, torch_stream=False)
    ctx.synchronize(); t = time.perf_counter(); b.run(comm, gb); ts.append(time.perf_counter() - t)
    b.close()
    for q in pp.polynomials: q.free()
print("shard world1 gb", gb, "ms", [round(x * 1e3, 3) for x in ts])
comm.close()
