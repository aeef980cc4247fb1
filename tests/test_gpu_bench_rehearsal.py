"""bench.py's N > 1 branch (strong-scaling fold of the 2^24 table sharded by index mod N, the sharded n = 24 prover through
zk_shard_prover_run, the four-step NTT) only runs when the driver launches several ranks.  ZK_BENCH_REHEARSE=1 runs the same
code with two ranks on ONE GPU (gloo group, host-staged collectives instead of RCCL): this checks that the line comes out, is
well-formed and that every rank derived the same challenges -- not its numbers."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, ZK_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 40
    assert d["config"]["elements_per_gpu"] == 1 << 23
    assert d["value"] > 0 and d["roofline"]["algorithmic_bytes"] == 48 << 23
    ex = d["extra"]
    assert "sharded_error" not in ex and "sharded_ntt_error" not in ex, ex
    assert ex["sharded_challenges_identical_on_all_ranks"] is True
    assert ex["sharded_sumcheck_local_vars"] == 23 and ex["sharded_sumcheck_ms_n24_k2_d2_world2"] > 0
    assert ex["sharded_ntt_ms_2p24_world2"] > 0
