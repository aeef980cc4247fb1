#!/usr/bin/env python3
"""Gap analysis of a rocprofv3 --kernel-trace csv: for the last `count` kernel launches, total busy time, total idle time
between consecutive kernels, and the largest gaps with the kernels around them (where does the host fail to keep the GPU fed?)."""
import csv, glob, os, sys
src = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 700
f = max(glob.glob(src + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-count:]
busy = gap_total = 0
gaps = []
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if prev is not None:
        g = s - int(prev["End_Timestamp"])
        if g > 0:
            gap_total += g
            gaps.append((g, prev["Kernel_Name"].split("(")[0][-40:], r["Kernel_Name"].split("(")[0][-40:]))
    prev = r
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} launches: span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {gap_total/1e3:.1f} us")
hist = {}
for g, a, b in gaps:
    k = (a, b)
    hist.setdefault(k, [0, 0])
    hist[k][0] += g
    hist[k][1] += 1
for (a, b), (tot, n) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {tot/1e3:8.1f} us in {n:4d} gaps (avg {tot/n/1e3:6.2f})  after {a}  before {b}")
