"""A/B of the GKR driver (depth 8, width 2^20) under environment switches, each arm in its own child process, interleaved
three times: python3 tools/ab_gkr_env.py "ZK_GKR_FUSE_ROUND0=0" "ZK_GKR_FUSE_BLOCKS=768" ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(3):
    for arm in sys.argv[1:]:
        env = dict(os.environ)
        for kv in arm.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                env[k] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_gkr.py"), "20", "8"], env=env, capture_output=True, text=True, timeout=300)
        ms = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("prove ms")]
        print(f"[{arm:>40}] prove min {min(ms):.3f} ms" if ms else f"[{arm}] {r.stderr[-300:]}", flush=True)
