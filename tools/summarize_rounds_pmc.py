#!/usr/bin/env python3
"""HBM traffic of every launch of ONE prove_partial call from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB; gfx950:
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, /opt/skills/guides/MI355X_MICROARCH.md) next to the algorithmic bytes
k * (2^m * 32 read + 2^(m-1) * 32 written) of the round.  usage: summarize_rounds_pmc.py <dir> <tag> <n> <k>"""
import csv
import glob
import os
import sys

src, tag, n, k = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])


def seq(ctr):
    f = glob.glob(os.path.join(src, ctr, "**", "*_counter_collection.csv"), recursive=True)[0]
    return sorted((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"]))
                  for r in csv.DictReader(open(f)))


f, w = seq("FETCH_SIZE"), seq("WRITE_SIZE")
def fused_arg(name):   # k_round_kd<K, D, FUSED, EXTRA, SKIP1>
    return name.split("<")[1].split(",")[2].strip() == "true"


start = max(i for i, x in enumerate(f) if "k_round_kd<" in x[1] and not fused_arg(x[1]))   # round 0 of the last proof
lines = [f"# HBM traffic per launch of one prove_partial call (n = {n}, k = {k}, D = 2, BN254 Fr) `{tag}`", "",
         "`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes) on "
         f"`python3 tools/prof_sumcheck.py {n} 2`; bytes = (2·FETCH_SIZE + WRITE_SIZE)·1024 (gfx950 correction).", "",
         "| launch | kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MiB | algorithmic MiB | ratio |", "|---|---|---|---|---|---|---|"]
rnd = 0
for i, ((_, name, fv), (_, name2, wv)) in enumerate(zip(f[start:], w[start:])):
    assert name == name2
    if "copyBuffer" in name:
        continue
    hbm = (2 * fv + wv) * 1024 / 2 ** 20
    alg = ""
    if "k_round_kd" in name:
        m = n - rnd                      # variables of the table this round reads
        if not fused_arg(name):          # sums only: reads k tables of 2^m
            a = k * (2 ** m) * 32 / 2 ** 20
        else:                            # fused: reads the 2^(m+1) table of the previous round, writes 2^m
            a = k * (2 ** (m + 1) * 32 + 2 ** m * 32) / 2 ** 20
        alg = f"{a:.1f}"
        ratio = f"{hbm / a:.4f}"
        rnd += 1
    else:
        ratio = ""
    lines.append(f"| {i} | `{name}` | {fv:.0f} | {wv:.0f} | {hbm:.1f} | {alg} | {ratio} |")
    if "k_finish" in name:
        break
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_prover_round_traffic.md")
open(out, "w").write("\n".join(lines) + "\n")
print(open(out).read())
