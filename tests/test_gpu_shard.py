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
    return orc.sum_elems(field, orc.prod_reduce(field, n, tabs))   # iter().sum::<F>() (sumcheck/src/lib.rs:56)


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


SHARD_RUNS = {
    # SKIP1 + LEAD in every fused round of every shape that has them (no quad kernel: it would take the small rounds)
    "skip1_lead_everywhere": dict(ZK_SKIP1_MIN_PAIRS="1", ZK_LEAD_MIN_PAIRS="1", ZK_QUAD_MAX_PAIRS="0"),
    # LEAD only (round 0 included: sums-only LEAD kernels), then SKIP1 only
    "lead_only": dict(ZK_LEAD_MIN_PAIRS="1", ZK_QUAD_MAX_PAIRS="0"),
    "skip1_only": dict(ZK_SKIP1_MIN_PAIRS="1", ZK_QUAD_MAX_PAIRS="0"),
    # the shipped thresholds on shards of 2^18 (two ranks) and 2^17 elements (four ranks, three factors)
    "shipped_thresholds_big_shards": dict(ZK_CHECK_CASES="2:2:2:19,4:3:3:19", ZK_CHECK_FIELDS="1"),
    # round 4's behaviour: every sum formed by the round kernels
    "derivation_off": dict(ZK_SHARD_SKIP1="0", ZK_SKIP1_MIN_PAIRS="1", ZK_LEAD_MIN_PAIRS="1", ZK_QUAD_MAX_PAIRS="0", ZK_CHECK_FIELDS="1"),
    # the claim S_prev(r_prev) evaluated by k_lanes_transcript itself (no claim workgroup in the round kernels)
    # the LDS-DMA round kernels (k_round0_glds, k_round_fused_glds) on every shard of at least 64 pairs, cached and nontemporal half tables
    "glds_everywhere": dict(ZK_ROUND_GLDS_MIN_PAIRS="64", ZK_ROUND_GLDS_NT_MIN_PAIRS="256", ZK_SKIP1_MIN_PAIRS="1", ZK_LEAD_MIN_PAIRS="1",
                            ZK_QUAD_MAX_PAIRS="0", ZK_CHECK_CASES="2:2:2:10,2:3:3:10,4:2:2:11,1:2:2:9,1:3:3:9,4:3:3:12", ZK_CHECK_FIELDS="2"),
    # ... and switched off on the big shards (k_round0_dot29 / k_round_kd where the defaults now select the LDS-DMA forms)
    "glds_off_big_shards": dict(ZK_ROUND_GLDS="0", ZK_CHECK_CASES="2:2:2:20,2:3:3:20", ZK_CHECK_FIELDS="1"),
    "claim_in_lanes_transcript": dict(ZK_CLAIM_IN_ROUND="0", ZK_SKIP1_MIN_PAIRS="1", ZK_LEAD_MIN_PAIRS="1", ZK_QUAD_MAX_PAIRS="0", ZK_CHECK_FIELDS="1"),
}


@pytest.fixture(scope="module")
def shard_sweeps(tmp_path_factory):
    """oracle proofs of the runs' grids once (tests/oracle_cache.py), then every child process, three at a time; -> {name: Future}"""
    import os
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import oracle_cache
    import shard_skip_check

    drop = ("ZK_SKIP1_MIN_PAIRS", "ZK_QUAD_MAX_PAIRS", "ZK_PIPE_MAX_PAIRS", "ZK_LEAD_MIN_PAIRS", "ZK_SHARD_SKIP1", "ZK_CHECK_CASES", "ZK_CHECK_FIELDS",
            "ZK_CLAIM_IN_ROUND", "ZK_ROUND_GLDS", "ZK_ROUND_GLDS_MIN_PAIRS", "ZK_ROUND_GLDS_NT_MIN_PAIRS")
    cache = str(tmp_path_factory.mktemp("oracle_cache_shard"))
    base = {k: v for k, v in os.environ.items() if k not in drop}
    base["ZK_ORACLE_CACHE"] = cache
    spec = []
    for extra in SHARD_RUNS.values():
        spec += shard_skip_check.spec(shard_skip_check.parse_cases(extra.get("ZK_CHECK_CASES", shard_skip_check.DEFAULT_CASES)),
                                      int(extra.get("ZK_CHECK_FIELDS", "3")))
    os.environ["ZK_ORACLE_CACHE"] = cache
    try:
        oracle_cache.prefill(spec, workers=min(8, max(2, (os.cpu_count() or 4) - 2)))
    finally:
        del os.environ["ZK_ORACLE_CACHE"]

    def child(extra):
        return subprocess.run([sys.executable, os.path.join(root, "tests", "shard_skip_check.py")], env=dict(base, **extra), capture_output=True,
                              text=True, timeout=900)

    pool = ThreadPoolExecutor(max_workers=3)
    futures = {name: pool.submit(child, extra) for name, extra in SHARD_RUNS.items()}
    yield futures
    pool.shutdown(wait=True)


@pytest.mark.parametrize("name", list(SHARD_RUNS))
def test_shard_rounds_derive_behind_the_allreduce(shard_sweeps, name):
    """The sharded prover's big rounds run the kernels that leave out S(1) and accumulate the leading coefficient instead of S(D)
    (both linear in the shards); k_lanes_transcript derives the two from the all-reduced lanes.  Child processes force the variants
    on at every size (tests/shard_skip_check.py), run the default thresholds on shards big enough to reach them, and run with the
    derivation switched off; every proof is compared with the oracle's proof of the unsharded tables bit for bit."""
    r = shard_sweeps[name].result()
    assert r.returncode == 0 and "shard skip ok" in r.stdout, str(SHARD_RUNS[name]) + r.stdout + r.stderr


def test_interleaved_shard_provers_and_plain_proofs_on_one_context():
    """The claim S_prev(r_prev) a SKIP1 round kernel parks for k_lanes_transcript stays live from round_begin to round_finish, i.e.
    across API calls: it belongs to the prover (ProverScratch), not to the context.  Two DIFFERENT sharded proofs (one rank each,
    tables of 2^18 elements: the shipped thresholds select SKIP1 + LEAD from 2^16 pairs) stepped in lockstep on ONE context, with
    a plain prove_partial of a third polynomial enqueued between every begin and finish, each equal to the oracle's proof."""
    field = zk_amd.BN254_FR
    ctx = zk_amd.Context(field, 0)
    n, k, D = 18, 2, 2
    cases = []
    for j in range(3):
        tabs = [orc.fill_random(field, 9100 + 10 * j + f, 1 << n) for f in range(k)]
        claimed = claimed_sum(field, n, tabs)
        cases.append((tabs, claimed, orc.sumcheck_prove(field, n, tabs, D, claimed, False)))
    polys = [ProductPoly.new([MLE.new(ctx, n, t) for t in tabs]) for tabs, _, _ in cases]
    a, b = (GpuShardBackend(polys[j], D, cases[j][1], 1) for j in (0, 1))
    plain = zk_amd.SumcheckProver(D)
    while a.local_vars_left() > 3:
        a.round_begin()
        b.round_begin()
        proof, ch = plain.prove_partial(polys[2], cases[2][1])
        assert np.array_equal(proof.round_polys, cases[2][2][0]) and np.array_equal(ch, cases[2][2][1])
        b.round_finish()      # the other order than begin
        a.round_finish()
    for j, be in enumerate((a, b)):
        be.tail_rounds(be.tail().clone())
        rp, ch = be.results()
        assert np.array_equal(rp, cases[j][2][0]) and np.array_equal(ch, cases[j][2][1]), f"prover {j}"
        be.close()
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


# ---- four-step NTT across ranks (zk_amd.distributed.ShardedNtt / GpuNttBackend) --------------------------------------
@pytest.mark.parametrize("field", [zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR])
def test_mul_powers_and_dft_across_primitives(field):
    from zk_amd._lib import check, lib, u64p

    ctx = zk_amd.Context(field, 0)
    p = zk_amd.modulus(field)
    n = 13
    x = orc.fill_random(field, 77, 1 << n)
    xs = orc.to_ints(field, x)
    base, scale = 0x1234567 * 31337 % p, 0xABCDEF % p
    t = MLE.new(ctx, n, x)
    check(lib.zk_mle_mul_powers(ctx._h, t._h, zk_amd.fe_from_int(field, base).ctypes.data_as(u64p),
                                zk_amd.fe_from_int(field, scale).ctypes.data_as(u64p)))
    assert zk_amd.fe_to_ints(field, t.evaluation_slice()) == [v * scale * pow(base, j, p) % p for j, v in enumerate(xs)]
    for log_w in (0, 1, 3):
        W, L = 1 << log_w, 1 << (n - log_w)
        w = zk_amd.fe_to_int(field, zk_amd.root_of_unity(field, log_w))
        assert zk_amd.root_of_unity(field, log_w).tolist() == orc.root_of_unity(field, W).tolist()
        for inverse in (0, 1):
            ww = pow(w, -1, p) if inverse else w
            out, src = MLE.alloc(ctx, n), MLE.new(ctx, n, x)
            check(lib.zk_dft_across(ctx._h, src._h, out._h, log_w, inverse))
            got = zk_amd.fe_to_ints(field, out.evaluation_slice())
            for k in range(W):
                for j in (0, 1, L // 2, L - 1):
                    assert got[k * L + j] == sum(xs[r * L + j] * pow(ww, r * k, p) for r in range(W)) % p


@pytest.mark.parametrize("field", [zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR])
@pytest.mark.parametrize("world,log_n", [(1, 9), (2, 10), (4, 12), (8, 16), (8, 6), (2, 2)])
def test_sharded_ntt_rehearsal_matches_oracle(field, world, log_n):
    """W GpuNttBackends in ONE process on one GPU, the all-to-all done by hand with torch ops, against the oracle's
    transform of the whole vector (fft/src/lib.rs:4-19); forward then inverse."""
    import torch

    from zk_amd.distributed import GpuNttBackend, sliced_shard_of

    ctx = zk_amd.Context(field, 0)
    x = orc.fill_random(field, 31 + log_n, 1 << log_n)
    X = orc.ntt_fast(field, x, False)
    m = log_n - int(np.log2(world))

    def exchange(backends):
        sends = [b.send_tensor().view(world, -1) for b in backends]
        for s, b in enumerate(backends):
            b.recv_tensor().view(world, -1).copy_(torch.stack([sends[r][s] for r in range(world)]))

    fw = [GpuNttBackend(MLE.new(ctx, m, shard_of(x, r, world)), r, world) for r in range(world)]
    for b in fw:
        b.local_ntt(False)
        b.twiddle(False)
    exchange(fw)
    for r, b in enumerate(fw):
        b.across(False)
        assert np.array_equal(b.result().evaluation_slice(), sliced_shard_of(X, r, world)), f"forward, rank {r}"
    bw = [GpuNttBackend(MLE.new(ctx, m, sliced_shard_of(X, r, world)), r, world) for r in range(world)]
    for b in bw:
        b.across(True)
    exchange(bw)
    for r, b in enumerate(bw):
        b.twiddle(True)
        b.local_ntt(True)
        assert np.array_equal(b.result().evaluation_slice(), shard_of(x, r, world)), f"inverse, rank {r}"
    # and through the orchestration class itself (world of one process: the exchange is a local copy)
    if world == 1:
        from zk_amd.distributed import ShardedNtt

        assert np.array_equal(ShardedNtt(GpuNttBackend(MLE.new(ctx, m, x), 0, 1)).forward().evaluation_slice(), X)


def test_sharded_ntt_needs_world_elements_per_shard():
    from zk_amd.distributed import GpuNttBackend

    ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
    with pytest.raises(ValueError, match="at least `world` elements"):
        GpuNttBackend(MLE.random(ctx, 2, 1), 0, 8)
    with pytest.raises(ValueError, match="power of two"):
        GpuNttBackend(MLE.random(ctx, 4, 1), 0, 3)


# ---- the whole loop inside the library over RCCL (zk_shard_prover_run / zk_ntt_sharded), one rank ------------------------
# The multi-rank form of the same calls runs in tests/test_gpu_multiproc.py over the host-callback transport (RCCL refuses
# two ranks on one device); here the RCCL transport itself -- communicator creation from a unique id, ncclAllReduce /
# ncclAllGather / ncclAllToAll enqueued on the context's stream -- runs at world 1 and must reproduce prove_partial.
@pytest.mark.parametrize("field", [zk_amd.BN254_FR, zk_amd.BLS12_381_FR])
def test_rccl_world1_run_matches_prove_partial_and_oracle(field):
    from zk_amd.distributed import RcclComm, ntt_sharded

    ctx = zk_amd.Context(field, 0)
    comm = RcclComm(ctx)
    assert (comm.world, comm.rank) == (1, 0)
    for n, k, D, gather_below in [(12, 2, 2, 10), (14, 2, 2, 0), (16, 2, 2, 10), (13, 3, 3, 4), (18, 2, 2, 10)]:
        tabs = [orc.fill_random(field, 1400 + 8 * n + f, 1 << n) for f in range(k)]
        claimed = claimed_sum(field, n, tabs)
        poly = ProductPoly.new([MLE.new(ctx, n, t) for t in tabs])
        plain_proof, plain_ch = zk_amd.SumcheckProver(D).prove_partial(poly, claimed)
        rp, ch = GpuShardBackend(poly, D, claimed, 1).run(comm, gather_below)
        assert np.array_equal(rp, plain_proof.round_polys) and np.array_equal(ch, plain_ch), (n, k, D, gather_below)
        if n <= 16:
            want_rp, want_ch = orc.sumcheck_prove(field, n, tabs, D, claimed, False)
            assert np.array_equal(rp, want_rp) and np.array_equal(ch, want_ch)
    # at world 1 the sharded transform is the plain one (the all-to-all is the identity, the across-step a 1-point DFT)
    x = orc.fill_random(field, 1500, 1 << 12)
    xs = MLE.new(ctx, 12, x)
    X = ntt_sharded(comm, xs, False)
    assert np.array_equal(X.evaluation_slice(), orc.ntt_fast(field, x, False))
    assert np.array_equal(ntt_sharded(comm, X, True).evaluation_slice(), x)
    ctx.use_own_stream()
    comm.close()
    ctx.close()


def test_comm_argument_checks():
    from zk_amd._lib import ZkError, c, check, lib

    ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
    h = c.c_void_p()
    with pytest.raises(ZkError):   # world must be a power of two, rank < world
        check(lib.zk_comm_create_rccl(ctx._h, bytes(128), 3, 0, c.byref(h)))
    with pytest.raises(ZkError):
        check(lib.zk_comm_create_rccl(ctx._h, bytes(128), 2, 2, c.byref(h)))
    with pytest.raises(ZkError):   # host transport needs its callbacks
        check(lib.zk_comm_create_host(ctx._h, 2, 0, None, None, None, None, c.byref(h)))
    ctx.close()
