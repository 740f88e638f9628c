#!/usr/bin/env python3
"""Per-launch durations of one ADMM iteration's solve sweeps from a rocprofv3 kernel trace.
usage: level_trace.py <dir with *_kernel_trace.csv>   (prints the kernels between the last two rhs_gather launches)"""
import csv, glob, os, sys
paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)))
rows.sort()
rhs = [i for i, r in enumerate(rows) if "rhs_gather" in r[2]]
a, b = rhs[-2], rhs[-1]
prev_end = rows[a][1]
tot = {}
for s, e, name, grid, wg in rows[a:b]:
    short = name.split("(")[0].replace("void admm_dev::", "").replace("admm_dev::", "")
    print("%-34s grid %8d wg %5d  dur %8.2f us  gap %6.2f us" % (short[:34], grid // max(wg, 1), wg, (e - s) / 1e3, (s - prev_end) / 1e3))
    tot[short] = tot.get(short, 0) + (e - s) / 1e3
    prev_end = e
print("iteration span %.1f us" % ((rows[b][0] - rows[a][0]) / 1e3))
for k, v in tot.items():
    print("  %-34s %8.1f us" % (k[:34], v))
