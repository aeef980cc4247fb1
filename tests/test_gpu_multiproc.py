"""BASELINE config 3 in its real form: several OS processes, each driving its own zk_shard_prover on the GPU, exchanging
through a process group -- against the oracle's single-process prover on the unsharded table
(sumcheck/src/prover.rs:24-30,44-68).

Only one GPU is available to the tests and RCCL refuses two ranks on one device, so the ranks (fresh child processes, the
children make their own first GPU call) all open cuda:0 and the exchange rides on a gloo group:
  * driver "py":  zk_amd.distributed.ShardedSumcheckProver stepping zk_shard_prover_round_begin / _finish, the lanes and
                  the tail staged through the host for the gloo collectives;
  * driver "lib": zk_shard_prover_run -- the whole loop inside the library -- with the host-callback transport (HostComm).
                  The RCCL transport of the same loop runs at world 1 in tests/test_gpu_shard.py.
The four-step NTT (ShardedNtt / GpuNttBackend and zk_ntt_sharded) is covered the same way.
One spawn per world size; every child runs the whole case grid and checks its own results (every rank must hold the
complete, identical proof)."""
import datetime
import json
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist

    # a rank that fails leaves the others inside a collective: bound that wait
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    report = {"rank": rank, "cases": 0, "errors": []}
    try:
        import zk_amd
        from oracle import binding as orc
        from zk_amd import MultiLinearPolynomial as MLE
        from zk_amd import ProductPoly
        from zk_amd.distributed import (GpuNttBackend, GpuShardBackend, HostComm, ShardedNtt, ShardedSumcheckProver, ntt_sharded,
                                        shard_of, sliced_shard_of)

        lw = world.bit_length() - 1
        for field in (zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR):
            ctx = zk_amd.Context(field, 0)
            comm = HostComm(ctx)
            for n in (12, 13, 14, 15, 16):
                k, D = (3, 3) if n == 13 else (2, 2)
                tabs = [orc.fill_random(field, 7000 + 16 * n + f, 1 << n) for f in range(k)]
                claimed = orc.sum_elems(field, orc.prod_reduce(field, n, tabs))   # iter().sum::<F>()
                want_rp, want_ch = orc.sumcheck_prove(field, n, tabs, D, claimed, False)
                for gather_below in (0, 10):
                    for driver in ("py", "lib"):
                        poly = ProductPoly.new([MLE.new(ctx, n - lw, shard_of(t, rank, world)) for t in tabs])
                        backend = GpuShardBackend(poly, D, claimed, world)
                        if driver == "py":
                            rp, ch = ShardedSumcheckProver(backend, gather_below=gather_below).prove_partial()
                        else:
                            rp, ch = backend.run(comm, gather_below)
                        report["cases"] += 1
                        if not (np.array_equal(rp, want_rp) and np.array_equal(ch, want_ch)):
                            report["errors"].append(f"prover field={field} n={n} gb={gather_below} driver={driver}")
                        backend.close()
                        for q in poly.polynomials:
                            q.free()
            # four-step NTT across the ranks: forward strided -> sliced, inverse sliced -> strided (fft/src/lib.rs:4-19)
            for log_n in (10, 13):
                x = orc.fill_random(field, 4300 + log_n, 1 << log_n)
                X = orc.ntt_fast(field, x, False)
                xs = MLE.new(ctx, log_n - lw, shard_of(x, rank, world))
                Xs = MLE.new(ctx, log_n - lw, sliced_shard_of(X, rank, world))
                got = {
                    "py_fwd": ShardedNtt(GpuNttBackend(xs, rank, world)).forward().evaluation_slice(),
                    "py_inv": ShardedNtt(GpuNttBackend(Xs, rank, world)).inverse().evaluation_slice(),
                    "lib_fwd": ntt_sharded(comm, xs, False).evaluation_slice(),
                    "lib_inv": ntt_sharded(comm, Xs, True).evaluation_slice(),
                }
                for name, val in got.items():
                    want = sliced_shard_of(X, rank, world) if name.endswith("fwd") else shard_of(x, rank, world)
                    report["cases"] += 1
                    if not np.array_equal(val, want):
                        report["errors"].append(f"ntt field={field} log_n={log_n} {name}")
            ctx.use_own_stream()
            comm.close()
            ctx.close()
    except Exception as e:   # reported, not raised: the other ranks must not be left waiting in a collective forever
        import traceback

        report["errors"].append("exception: " + repr(e) + "\n" + traceback.format_exc())
    finally:
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(report, f)
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multiprocess_gpu_shard_provers_and_ntt_match_unsharded_oracle(tmp_path, world):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        rep = json.load(open(os.path.join(str(tmp_path), f"rank{r}.json")))
        assert rep["errors"] == [], f"rank {r}: {rep['errors']}"
        assert rep["cases"] == 3 * (5 * 2 * 2 + 2 * 4), rep
