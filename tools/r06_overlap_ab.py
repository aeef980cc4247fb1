"""[HISTORICAL -- applies to commit 9312777: the ZK_SHARD_OVERLAP schedule lost to the serial loop up to A = 40 us and was removed
together with k_round_mid; logs: profiles/r06_shard_overlap_ab.log, profiles/r06_shard_overlap_trace.log]
VERDICT r05 item 5: the sharded loop with the all-reduce off the critical path (ZK_SHARD_OVERLAP=1: three streams, pending-challenge sums
exchanged one round ahead) against the serial loop, at ONE rank through RCCL with an injected all-reduce latency A
(ZK_SHARD_FAKE_ALLREDUCE_US: a spin kernel on the collective's stream after every all-reduce) standing in for A_8.
  n = 24, gather_below 13: the verdict's case (11 exchanging rounds, tables 2^24 .. 2^14)
  n = 21, gather_below 13: the per-rank work of W = 8 at n = 24 (8 exchanging rounds on 2^21 .. 2^14-element shards)
Child processes (the switches are read once per process), alternating, median of `reps` runs each."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = sys.argv[1] if len(sys.argv) > 1 else "9"
arms = [("serial", {}), ("overlap<=2^17", {"ZK_SHARD_OVERLAP": "1"}), ("overlap<=2^20", {"ZK_SHARD_OVERLAP": "1", "ZK_SHARD_OVERLAP_MAX_PAIRS": str(1 << 20)})]
for n, gb in ((21, 13), (24, 13)):
    for A in (0, 10, 15, 25, 40):
        row = {}
        for name, env in arms:
            e = {k: v for k, v in os.environ.items() if not k.startswith("ZK_SHARD_")}
            e.update(env)
            if A:
                e["ZK_SHARD_FAKE_ALLREDUCE_US"] = str(A)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_shard.py"), str(gb), reps, str(n)], env=e, capture_output=True, text=True, timeout=600)
            m = re.search(r"ms \[(.*)\]", r.stdout)
            if not m:
                print(name, "FAILED", r.stdout[-500:], r.stderr[-1500:], flush=True)
                continue
            ts = sorted(float(x) for x in m.group(1).split(","))
            row[name] = ts[len(ts) // 2]
        if "serial" in row:
            rounds = n - gb
            print(f"n={n} gather_below={gb} ({rounds} exchanging rounds) A={A:2d} us: " + "  ".join(f"{k} {v:.3f} ms" for k, v in row.items()) +
                  "   | saved vs serial: " + "  ".join(f"{k} {(row['serial'] - v) * 1e3:.0f} us" for k, v in row.items() if k != "serial") +
                  f"   | 0.6 * 8 * A = {0.6 * 8 * A:.0f} us", flush=True)
