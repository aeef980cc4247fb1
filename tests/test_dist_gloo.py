"""N > 1 path on CPU: the sharded prover's orchestration (zk_amd.distributed.ShardedSumcheckProver) under
torch.distributed/gloo with world_size 2 and 4, one process per rank, against the oracle's single-process prover on
the unsharded table.  The per-rank compute is the oracle-backed stand-in (tests/shard_backend_oracle.py); the GPU
backend runs the same protocol in tests/test_gpu_shard.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, field, k, D, n_vars, out_dir, gather_below=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from shard_backend_oracle import OracleShardBackend

        from oracle import binding as orc
        from zk_amd.distributed import ShardedSumcheckProver, shard_of

        tabs = [orc.fill_random(field, 900 + f, 1 << n_vars) for f in range(k)]
        claimed = orc.sum_elems(field, orc.prod_reduce(field, n_vars, tabs))   # iter().sum::<F>()
        backend = OracleShardBackend(field, [shard_of(t, rank, world) for t in tabs], D, claimed, world)
        rp, ch = ShardedSumcheckProver(backend, gather_below=gather_below).prove_partial()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rp=rp, ch=ch)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("field,k,D,n_vars,gather_below", [(0, 2, 2, 6, 0), (1, 1, 1, 5, 0), (0, 3, 3, 4, 0), (2, 2, 2, 2, 0),
                                                           (0, 2, 2, 6, 2), (1, 2, 1, 5, 10)])
def test_sharded_prover_matches_single_process_oracle(tmp_path, world, field, k, D, n_vars, gather_below):
    """gather_below = 0: a collective in every local round, gather only the final elements; 2: stop exchanging when the
    local tables have 4 elements; 10: gather at once (no per-round collective at all)."""
    if (1 << n_vars) < world:
        pytest.skip("table smaller than the world")
    from oracle import binding as orc

    port = _free_port()
    mp.spawn(_worker, args=(world, port, field, k, D, n_vars, str(tmp_path), gather_below), nprocs=world, join=True)
    tabs = [orc.fill_random(field, 900 + f, 1 << n_vars) for f in range(k)]
    claimed = orc.sum_elems(field, orc.prod_reduce(field, n_vars, tabs))   # iter().sum::<F>()
    want_rp, want_ch = orc.sumcheck_prove(field, n_vars, tabs, D, claimed, False)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.array_equal(got["rp"], want_rp), f"rank {r} round polys differ"
        assert np.array_equal(got["ch"], want_ch), f"rank {r} challenges differ"


def test_shard_of_is_the_suffix_shard():
    from zk_amd.distributed import shard_of

    t = np.arange(16 * 4, dtype=np.uint64).reshape(16, 4)
    for w in (1, 2, 4, 8):
        parts = [shard_of(t, g, w) for g in range(w)]
        for g, part in enumerate(parts):
            assert np.array_equal(part[:, 0] // 4, np.arange(g, 16, w))   # global idx = local*w + g


def _ntt_worker(rank, world, port, field, log_n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from shard_backend_oracle import OracleNttBackend

        from oracle import binding as orc
        from zk_amd.distributed import ShardedNtt, shard_of, sliced_shard_of

        x = orc.fill_random(field, 4242, 1 << log_n)
        fwd = ShardedNtt(OracleNttBackend(field, shard_of(x, rank, world), rank, world)).forward().copy()
        X = orc.ntt_fast(field, x, False)
        inv = ShardedNtt(OracleNttBackend(field, sliced_shard_of(X, rank, world), rank, world)).inverse().copy()
        np.savez(os.path.join(out_dir, f"ntt{rank}.npz"), fwd=fwd, inv=inv)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("field,log_n", [(0, 6), (2, 5), (1, 4)])
def test_sharded_ntt_matches_single_process_oracle(tmp_path, world, field, log_n):
    """four-step NTT over gloo: forward (strided shards -> sliced shards of fft(x)) and inverse (sliced -> strided),
    one all-to-all each, against the oracle's transform of the whole vector (fft/src/lib.rs:4-19)."""
    from oracle import binding as orc
    from zk_amd.distributed import shard_of, sliced_shard_of

    port = _free_port()
    mp.spawn(_ntt_worker, args=(world, port, field, log_n, str(tmp_path)), nprocs=world, join=True)
    x = orc.fill_random(field, 4242, 1 << log_n)
    X = orc.fft(field, x) if log_n <= 5 else orc.ntt_fast(field, x, False)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"ntt{r}.npz"))
        assert np.array_equal(got["fwd"], sliced_shard_of(X, r, world)), f"rank {r}: forward shard differs"
        assert np.array_equal(got["inv"], shard_of(x, r, world)), f"rank {r}: inverse shard differs"
