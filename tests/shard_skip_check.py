"""Run by tests/test_gpu_shard.py::test_shard_rounds_derive_behind_the_allreduce in child processes (the library reads its
switches once per process).  W shard provers in one process on one GPU, the per-round lane all-reduce done by hand, against the
oracle's prover on the unsharded tables (prover.rs:44-68) -- with the round kernels that leave out the t = 1 sums and / or
accumulate the leading coefficient (k_round_kd SKIP1 / LEAD) forced on at every size: the lanes then carry [S(0), 0, .., L] and
k_lanes_transcript derives S(1) and S(D) from the ALL-REDUCED values.  Also with a wrong claimed sum (the identity is about the
prover's own sums, not the claim) and with the derivation switched off (ZK_SHARD_SKIP1=0)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import zk_amd  # noqa: E402
from zk_amd import MultiLinearPolynomial as MLE  # noqa: E402
from zk_amd import ProductPoly  # noqa: E402
from zk_amd.distributed import GpuShardBackend, shard_of  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_cache  # noqa: E402  (oracle proofs: from $ZK_ORACLE_CACHE when the parent test prefilled it, else computed here)

DEFAULT_CASES = "1:2:2:8,2:2:2:9,4:3:3:8,2:2:3:7,8:2:2:10,2:1:1:6,4:2:2:5"
SEED = 4200


def parse_cases(text):
    return [tuple(int(x) for x in c.split(":")) for c in text.split(",")]


def spec(cases, n_fields):
    return [["sc", field, k, D, n, SEED + 10 * k, wrong] for field in range(n_fields) for _, k, D, n in cases for wrong in (0, 7)]


def main():
    cases = parse_cases(os.environ.get("ZK_CHECK_CASES", DEFAULT_CASES))
    fields = (zk_amd.BN254_FR, zk_amd.BLS12_381_FR, zk_amd.BLS12_377_FR)[: int(os.environ.get("ZK_CHECK_FIELDS", "3"))]
    checked = 0
    for field in fields:
        ctx = zk_amd.Context(field, 0)
        for world, k, D, n in cases:
            tabs = oracle_cache.sumcheck_tables(field, k, n, SEED + 10 * k)
            w = world.bit_length() - 1
            for wrong in (0, 7):
                _, s, want_rp, want_ch = oracle_cache.sumcheck_case(field, k, D, n, SEED + 10 * k, wrong, tabs=tabs)
                for gather_below in (0, 3):
                    backends = [GpuShardBackend(ProductPoly.new([MLE.new(ctx, n - w, shard_of(t, g, world)) for t in tabs]), D, s, world)
                                for g in range(world)]
                    while backends[0].local_vars_left() > gather_below:
                        lanes = [b.round_begin() for b in backends]
                        total = torch.stack(lanes).sum(dim=0)        # what all_reduce(SUM) leaves on every rank
                        for b, l in zip(backends, lanes):
                            l.copy_(total)
                            b.round_finish()
                    gathered = torch.cat([b.tail().clone() for b in backends])
                    for b in backends:
                        b.tail_rounds(gathered)
                    for g, b in enumerate(backends):
                        rp, ch = b.results()
                        assert np.array_equal(rp, want_rp), (field, world, k, D, n, wrong, gather_below, g)
                        assert np.array_equal(ch, want_ch), (field, world, k, D, n, wrong, gather_below, g)
                    for b in backends:
                        b.close()
                    checked += 1
        ctx.close()
    print(f"shard skip ok: {checked} sharded proofs bit-exact (ZK_SHARD_SKIP1={os.environ.get('ZK_SHARD_SKIP1')} "
          f"ZK_SKIP1_MIN_PAIRS={os.environ.get('ZK_SKIP1_MIN_PAIRS')} ZK_LEAD_MIN_PAIRS={os.environ.get('ZK_LEAD_MIN_PAIRS')})")


if __name__ == "__main__":
    main()
