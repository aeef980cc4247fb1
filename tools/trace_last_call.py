#!/usr/bin/env python3
"""Per-kernel totals of the LAST of `calls` identical calls in a rocprofv3 --kernel-trace csv (the trace is cut into `calls` equal
runs of launches after dropping the first `skip` launches): what one call costs, kernel by kernel, and its idle time."""
import csv, glob, os, sys
src, calls = sys.argv[1], int(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = max(glob.glob(src + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[skip:]
per = len(rows) // calls
rows = rows[-per:]
tot = {}
busy = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    tot.setdefault(k, [0, 0])
    tot[k][0] += d
    tot[k][1] += 1
    busy += d
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{per} launches per call: span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us")
for k, (t, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {t/1e3:9.1f} us  {n:5d} x {t/n/1e3:7.2f}  {k}")
