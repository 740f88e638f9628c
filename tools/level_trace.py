#!/usr/bin/env python3
"""Per-launch picture of one ADMM iteration from a rocprofv3 kernel trace: workgroups, duration (End - Start timestamp of the
dispatch) and the gap to the previous kernel's end.
usage: level_trace.py <dir with *_kernel_trace.csv>   (prints the kernels between the last two rhs_gather launches)

rocprofv3 stamps a dispatch when the command processor picks its packet up and when its last wave retires: between two
DEPENDENT kernels of one stream the next packet is picked up the moment the previous one completes, so `gap` is ~0 by
construction and the 1.5-1.9 us until the first wave of the next kernel runs sit INSIDE its duration.  The in-kernel view --
first workgroup start, staging barrier, last workgroup end per launch -- is tools/sweep_timeline.py (variant build with
real-time-counter stamps); `launch+ramp` below is that overhead where both are available."""
import csv, glob, os, sys

def col(r, *names):
    for n in names:
        if n in r and r[n] not in ("", None):
            return int(float(r[n]))
    return 0

paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            wg = max(col(r, "Workgroup_Size_X", "Workgroup_Size"), 1) * max(col(r, "Workgroup_Size_Y"), 1) * max(col(r, "Workgroup_Size_Z"), 1)
            grid = max(col(r, "Grid_Size_X", "Grid_Size"), 1) * max(col(r, "Grid_Size_Y"), 1) * max(col(r, "Grid_Size_Z"), 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid, wg, col(r, "VGPR_Count"), col(r, "LDS_Block_Size")))
rows.sort()
rhs = [i for i, r in enumerate(rows) if "rhs_gather" in r[2]]
a, b = rhs[-2], rhs[-1]
prev_end = rows[a - 1][1] if a > 0 else rows[a][0]
tot = {}
print("%-34s %9s %6s %5s %6s  %10s  %8s" % ("kernel", "workgroups", "wg", "vgpr", "lds", "dur us", "gap us"))
for s, e, name, grid, wg, vg, lds in rows[a:b]:
    short = name.split("(")[0].replace("void admm_dev::", "").replace("admm_dev::", "")
    print("%-34s %9d %6d %5d %6d  %10.2f  %8.2f" % (short[:34], grid // max(wg, 1), wg, vg, lds, (e - s) / 1e3, (s - prev_end) / 1e3))
    tot[short] = tot.get(short, 0) + (e - s) / 1e3
    prev_end = e
print("iteration span %.1f us (rhs_gather start to the next rhs_gather start)" % ((rows[b][0] - rows[a][0]) / 1e3))
for k, v in tot.items():
    print("  %-34s %8.1f us" % (k[:34], v))
