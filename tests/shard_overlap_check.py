"""Run by tests/test_gpu_shard.py::test_overlapped_exchange_schedule_bit_exact in child processes with ZK_SHARD_OVERLAP=1 (the library reads its
switches once per process): zk_shard_prover_run through RCCL at one rank with the three-stream schedule (work / collective / transcript,
comm_host.inc) -- the pending-challenge sums all-reduced as digit lanes one round ahead -- must return the plain prover's proof and the
oracle's, bit for bit, at every (size, shape, gather point), with and without an injected all-reduce latency."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import zk_amd  # noqa: E402
from oracle import binding as orc  # noqa: E402
from zk_amd import MultiLinearPolynomial as MLE  # noqa: E402
from zk_amd import ProductPoly  # noqa: E402
from zk_amd.distributed import GpuShardBackend, RcclComm  # noqa: E402

CASES = [(12, 2, 2, 10), (14, 2, 2, 0), (16, 2, 2, 10), (13, 3, 3, 4), (18, 2, 2, 10), (17, 3, 3, 3), (10, 1, 1, 0), (12, 2, 3, 4), (9, 2, 2, 0), (6, 2, 2, 0),
         (15, 1, 2, 5), (20, 2, 2, 13), (19, 3, 3, 12), (11, 4, 4, 2), (16, 2, 2, 16), (16, 2, 2, 15)]
checked = 0
for field in (zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR)[: int(os.environ.get("ZK_CHECK_FIELDS", "2"))]:
    ctx = zk_amd.Context(field, 0)
    comm = RcclComm(ctx)
    for n, k, D, gather_below in CASES:
        tabs = [orc.fill_random(field, 1600 + 8 * n + f, 1 << n) for f in range(k)]
        claimed = orc.sum_elems(field, orc.prod_reduce(field, n, tabs))
        for wrong in (0, 3):
            s = orc.add(field, claimed, orc.from_int(field, wrong)) if wrong else claimed
            poly = ProductPoly.new([MLE.new(ctx, n, t) for t in tabs])
            plain_proof, plain_ch = zk_amd.SumcheckProver(D).prove_partial(poly, s)
            rp, ch = GpuShardBackend(poly, D, s, 1).run(comm, gather_below)
            assert np.array_equal(rp, plain_proof.round_polys) and np.array_equal(ch, plain_ch), (field, n, k, D, gather_below, wrong)
            if n <= 18:
                want_rp, want_ch = orc.sumcheck_prove(field, n, tabs, D, s, False)
                assert np.array_equal(rp, want_rp) and np.array_equal(ch, want_ch), (field, n, k, D, gather_below, wrong)
            checked += 1
    ctx.use_own_stream()
    comm.close()
    ctx.close()
print(f"shard overlap ok: {checked} proofs bit-exact (ZK_SHARD_OVERLAP={os.environ.get('ZK_SHARD_OVERLAP')} "
      f"ZK_SHARD_OVERLAP_MAX_PAIRS={os.environ.get('ZK_SHARD_OVERLAP_MAX_PAIRS')} ZK_SHARD_FAKE_ALLREDUCE_US={os.environ.get('ZK_SHARD_FAKE_ALLREDUCE_US')})")
