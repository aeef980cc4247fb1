#!/usr/bin/env python3
"""Summarise a tools/gpu_profile.sh output tree (gpurun_out/prof) into profiles/<tag>_*.{csv,md,json}.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read,
so bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (all our loads/stores are 16 B per lane).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))


def short(name):
    return name.split("(")[0].replace("void ", "")


pmc = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(src, f"pmc_{ctr}", "*", "*_counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        pmc[k][ctr] = sum(v) / len(v)
        pmc[k]["launches"] = len(v)

traffic = {}
lines = [f"# rocprofv3 summary `{tag}`", "",
         "| kernel | calls | avg us | min us | max us | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (2*F+W)*1024 |",
         "|---|---|---|---|---|---|---|---|"]
for r in rows:
    k = short(r["Name"])
    f, w = pmc.get(k, {}).get("FETCH_SIZE"), pmc.get(k, {}).get("WRITE_SIZE")
    tb = (2 * f + w) * 1024 if f is not None and w is not None else None
    if tb is not None:
        traffic[k] = {"hbm_bytes_per_launch": tb, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                      "avg_ns": float(r["AverageNs"]), "calls": int(r["Calls"])}
    lines.append(f"| `{k}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                 f"{float(r['MaxNs'])/1e3:.1f} | {'' if f is None else f'{f:.1f}'} | {'' if w is None else f'{w:.1f}'} | "
                 f"{'' if tb is None else f'{tb:.0f}'} |")
lines += ["", f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}`",
          "PMC: two further runs of the same command with `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (own passes, kernel-trace only)."]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print("\n".join(lines))
