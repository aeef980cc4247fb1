"""GPU rehearsal of the sharded prover (zk_shard_prover_* through zk_amd.distributed.GpuShardBackend): W shard
provers in ONE process on one GPU, the per-round lane all-reduce and the tail all-gather done by hand with torch ops,
against the oracle's single-process prover on the unsharded table.  (The real multi-process run uses the same backend
with torch.distributed/RCCL; its orchestration is covered on CPU by tests/test_dist_gloo.py.)"""
import numpy as np
import pytest

import zk_amd
from oracle import binding as orc
from zk_amd import MultiLinearPolynomial as MLE
from zk_amd import ProductPoly
from zk_amd.distributed import GpuShardBackend, ShardedSumcheckProver, shard_of

pytestmark = pytest.mark.gpu


def claimed_sum(field, n, tabs):
    acc = np.zeros(4, dtype=np.uint64)
    for e in orc.prod_reduce(field, n, tabs):
        acc = orc.add(field, acc, e)
    return acc


@pytest.mark.parametrize("field", [zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR])
@pytest.mark.parametrize("gather_below", [0, 3, 12])
@pytest.mark.parametrize("world,k,D,n_vars", [(1, 2, 2, 8), (2, 2, 2, 8), (4, 1, 1, 7), (8, 2, 2, 10), (8, 3, 3, 6), (4, 2, 2, 2),
                                               (2, 2, 5, 6), (8, 2, 2, 3)])
def test_shard_provers_match_unsharded_oracle(field, world, k, D, n_vars, gather_below):
    import torch

    ctx = zk_amd.Context(field, 0)
    tabs = [orc.fill_random(field, 1200 + f, 1 << n_vars) for f in range(k)]
    claimed = claimed_sum(field, n_vars, tabs)
    want_rp, want_ch = orc.sumcheck_prove(field, n_vars, tabs, D, claimed, False)
    w = int(np.log2(world))
    backends = []
    for g in range(world):
        poly = ProductPoly.new([MLE.new(ctx, n_vars - w, shard_of(t, g, world)) for t in tabs])
        backends.append(GpuShardBackend(poly, D, claimed, world))
    while backends[0].local_vars_left() > gather_below:   # 0: exchange every local round; 12: gather at once
        lanes = [b.round_begin() for b in backends]
        total = torch.stack(lanes).sum(dim=0)          # what all_reduce(SUM) leaves on every rank
        for b, l in zip(backends, lanes):
            l.copy_(total)
            b.round_finish()
    gathered = torch.cat([b.tail().clone() for b in backends])   # all_gather, rank-major
    for b in backends:
        b.tail_rounds(gathered)
    for g, b in enumerate(backends):
        rp, ch = b.results()
        assert np.array_equal(rp, want_rp), f"rank {g}"
        assert np.array_equal(ch, want_ch), f"rank {g}"
    for b in backends:
        b.close()
    ctx.close()


def test_single_rank_orchestration_equals_plain_prover():
    field = zk_amd.BN254_FR
    ctx = zk_amd.Context(field, 0)
    n, k, D = 12, 2, 2
    tabs = [orc.fill_random(field, 1300 + f, 1 << n) for f in range(k)]
    claimed = claimed_sum(field, n, tabs)
    poly = ProductPoly.new([MLE.new(ctx, n, t) for t in tabs])
    rp, ch = ShardedSumcheckProver(GpuShardBackend(poly, D, claimed, 1)).prove_partial()
    want_rp, want_ch = orc.sumcheck_prove(field, n, tabs, D, claimed, False)
    assert np.array_equal(rp, want_rp) and np.array_equal(ch, want_ch)
