#!/usr/bin/env python3
"""Summarise a tools/gpu_profile_all.sh output tree (gpurun_out/prof_all) into profiles/<tag>_prover_ntt_gkr_kernel_stats.md."""
import csv
import glob
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
titles = {"sumcheck_n24": "`python3 tools/prof_sumcheck.py 24 5` (6 prove_partial calls, k=2, D=2, BN254 Fr, n=24)",
          "sumcheck_n20": "`python3 tools/prof_sumcheck.py 20 5` (6 prove_partial calls, n=20)",
          "ntt": "2^24-point NTT (6 forward + 6 inverse)",
          "gkr": "`python3 tools/prof_gkr.py 20 8` (4 zk_gkr_prove + 3 zk_gkr_verify, depth 8, width 2^20, random add/mul wiring)",
          "sumcheck_k3_n20": "`python3 tools/prof_k3.py 20` (6 prove_partial calls on a product of 3 MLEs, D = 3, n = 20: one GKR-shaped layer)",
          "batch8_k3_n20": "`python3 tools/prof_batch.py 20 3` (B = 1, 2, 4, 8 independent prove_partial calls, k = 3 and k = 2, n = 20, back to back AND as one zk_sumcheck_prove_batch: the `_b` kernels are the batched launches, grid (x, B))",
          "evaluate": "`python3 tools/prof_evaluate.py` (21 evaluate calls each at n = 18, 19, 20, 21, 24: k_eval_stream takes the low 12 / 15 variables at n = 21 / 24, k_eval_low the rest and everything below 21)",
          "fold": "`python3 tools/pmc_fold.py 24 200` (200 launches of the headline kernel k_fold_msb, 2^24 -> 2^23 BN254 Fr)"}
lines = [f"# rocprofv3 kernel stats `{tag}`: prover, NTT, GKR driver", "",
         "Each section: `rocprofv3 --kernel-trace --stats --output-format csv -- <command>` on one MI355X.", ""]
for name in ("fold", "sumcheck_n24", "sumcheck_n20", "sumcheck_k3_n20", "batch8_k3_n20", "evaluate", "ntt", "gkr"):
    fs = sorted(glob.glob(os.path.join(src, name, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    lines += [f"## {name}", "", titles[name], "", "| kernel | calls | avg us | min us | max us | total us | % |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(fs[-1])):
        k = r["Name"].split("(")[0].replace("void ", "")
        lines.append(f"| `{k}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | "
                     f"{float(r['TotalDurationNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
    log = os.path.join(src, name + ".log")
    if os.path.exists(log):
        wall = [l.strip() for l in open(log) if l.startswith(("n ", "k3 ", "ntt ms", "prove ms", "verify ms", "evaluate n=", "k="))]
        lines += ["", "wall clock reported by the script (under the profiler): " + "; ".join(f"`{w}`" for w in wall), ""]
out = os.path.join(root, "profiles", f"{tag}_prover_ntt_gkr_kernel_stats.md")
open(out, "w").write("\n".join(lines) + "\n")
print(open(out).read())
