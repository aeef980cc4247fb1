"""bench.py's N > 1 branch (strong-scaling fold of the 2^24 table sharded by index mod N, the sharded n = 24 prover through
zk_shard_prover_run, the four-step NTT) only runs when the driver launches several ranks.  ZK_BENCH_REHEARSE=1 runs the same
code with every rank on ONE GPU (gloo group, host-staged collectives instead of RCCL): the line must come out well-formed and
the sharded proof must be RIGHT -- verified against the true claimed sum, identical on every rank and bit-identical with the
proof of the unsharded 2^24 tables -- not its numbers.  Four ranks: the GPU box admits at most six processes on its card, so
the next power of two (eight, the node the driver uses) cannot be rehearsed on it; world 2 and 4 cover log2 W = 1 and > 1."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra_env=None, plain=False):
    """plain = the command exactly as the driver gives it at N = 1, with N = world: `python3 bench.py --gpus N --steps K --warmup W`
    and no launcher environment (bench.py self_launch starts the ranks); otherwise the driver's documented N > 1 launch line"""
    env = dict(os.environ, ZK_BENCH_REHEARSE="1", **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "40", "--warmup", "5", "--prewarm-ms", "20"]
    if plain:
        cmd = [sys.executable] + tail
    else:
        env["MASTER_ADDR"] = "127.0.0.1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


@pytest.mark.gpu
@pytest.mark.parametrize("world,plain", [(2, True), (4, True), (2, False)])
def test_bench_ranks_on_one_gpu(world, plain):
    r, lines = _run(world, plain=plain)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    lv = 24 - (world.bit_length() - 1)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["steps"] == 40
    assert d["config"]["elements_per_gpu"] == 1 << lv
    assert d["value"] > 0 and d["roofline"]["algorithmic_bytes"] == 48 << lv
    ex = d["extra"]
    assert "sharded_error" not in ex and "sharded_ntt_error" not in ex, ex
    assert ex["comm_confirms_n_ranks"] is True and ex["comm_rank_ids"] == list(range(world))
    assert ex["sharded_proof_verified"] is True
    assert ex["sharded_proof_identical_on_all_ranks"] is True
    assert ex["sharded_proof_equals_unsharded_proof"] is True
    assert ex["sharded_sumcheck_local_vars"] == lv and ex[f"sharded_sumcheck_ms_n24_k2_d2_world{world}"] > 0
    ph = ex[f"sharded_sumcheck_phases_ms_n24_k2_d2_world{world}_gather_below10"]
    assert set(ph) == {"local_kernels_ms", "allreduce_ms", "gather_ms", "tail_rounds_ms"} and all(v > 0 for v in ph.values())
    assert ex[f"sharded_ntt_ms_2p24_world{world}"] > 0 and ex["sharded_ntt_roundtrip_exact"] is True


@pytest.mark.gpu
def test_bench_world1_under_torchrun_matches_the_plain_line_shape():
    """the driver's N = 1 line is the plain `python bench.py`; a world-1 torchrun run takes the distributed branch (RCCL at one
    rank) and must produce the same headline fields plus a verified sharded proof"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5",
           "--no-cpu-baseline", "--no-pmc", "--no-parity-gate"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["elements_per_gpu"] == 1 << 24
    ex = d["extra"]
    assert "sharded_error" not in ex, ex
    assert ex["sharded_proof_verified"] is True and ex["sharded_proof_equals_unsharded_proof"] is True
    assert 0.3 < d["roofline"]["frac"] < 1.0


def test_plain_multi_gpu_command_launches_children_and_never_hangs():
    """CPU box (no GPU): `python bench.py --gpus 2` must start the two ranks as child processes, relay their failure (no device
    here) as a non-zero exit with no JSON line, and the parent itself must not import torch (it must never touch the GPU)."""
    env = dict(os.environ, ZK_BENCH_TRACE_PARENT_IMPORTS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("GPU box: the plain command is covered by test_bench_ranks_on_one_gpu")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--launch-timeout", "240"], env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "parent_imported_torch=False" in r.stderr, r.stderr[-1500:]


def test_gpus_must_be_a_power_of_two():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, timeout=60,
                       cwd=ROOT)
    assert r.returncode != 0 and "power of two" in r.stderr
