#!/usr/bin/env python3
"""Print the launch timeline of the LAST prove call in a rocprofv3 --kernel-trace csv: per launch the kernel, its
duration and the idle gap since the previous kernel ended (shows where the serial round chain spends its time)."""
import csv, glob, sys
src = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
import os
f = max(glob.glob(src + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)   # newest run
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
prev_end = None
tot_k = tot_gap = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{name:62s} grid={r.get('Grid_Size','?'):>8s} dur={(e-s)/1e3:8.2f} us gap={gap/1e3:7.2f} us")
    tot_k += e - s; tot_gap += gap; prev_end = e
print(f"kernels {tot_k/1e3:.1f} us, gaps {tot_gap/1e3:.1f} us, span {(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e3:.1f} us")
