#!/usr/bin/env python3
"""Timeline of the last ADMM iterations of a run from a rocprofv3 kernel trace, all queues: start offset, duration, queue,
kernel -- shows what overlaps what in the pipelined-groups mode.
usage: pipe_trace.py <dir with *_kernel_trace.csv> [iterations = 2]"""
import csv, glob, os, sys
paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0), r.get("Queue_Id", "?")))
rows.sort()
roots = [i for i, r in enumerate(rows) if "root_product" in r[2]]
a, b = roots[-1 - niter], roots[-1]
t0 = rows[a][1]
queues = {}
busy = {}
for s, e, name, grid, wg, q in rows[a + 1:b + 1]:
    short = name.split("(")[0].replace("void admm_dev::", "").replace("admm_dev::", "")
    qi = queues.setdefault(q, len(queues))
    print("%9.1f us  +%7.1f us  q%d  %-30s grid %7d" % ((s - t0) / 1e3, (e - s) / 1e3, qi, short[:30], grid // max(wg, 1)))
    busy[qi] = busy.get(qi, 0.0) + (e - s) / 1e3
print("span of %d iterations: %.1f us; kernel time per queue: %s" % (niter, (rows[b][1] - t0) / 1e3, {k: round(v, 1) for k, v in busy.items()}))
