#!/usr/bin/env python3
"""The class API's frame boundary from a rocprofv3 trace (kernel + memory-copy records): what happens between the last kernel of one
System::step() and the first ADMM kernel of the next, with the idle gaps between the pieces.

  cd /tmp && rocprofv3 --kernel-trace [--memory-copy-trace] --output-format csv -d /tmp/capi -- python3 $REPO/tools/probe/class_api_cost.py 32 32 163 trace
  (--memory-copy-trace crashed rocprofv3 on this pool in round 4: without it the DMAs show up as the idle time between the reordering kernels)
  python3 tools/class_api_timeline.py /tmp/capi
"""
import csv, glob, os, sys

d = sys.argv[1]
ev = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "").replace("admm_dev::", "")))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        nbytes = r.get("Bytes") or r.get("Size") or "0"
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %.2f MB" % (r.get("Direction", r.get("Name", "copy")), float(nbytes) / 1e6)))
ev.sort()
# frame boundaries: the epilogue kernel ends a frame's device work, the prologue kernel starts the next one's
epi = [i for i, e in enumerate(ev) if e[2].startswith("K epilogue_kernel")]
pro = [i for i, e in enumerate(ev) if e[2].startswith("K prologue_kernel")]
shown = 0
for a in epi:
    nxt = [p for p in pro if p > a]
    if not nxt:
        continue
    b = nxt[0]
    moves = [e for e in ev[a:b + 1] if e[2].startswith("C ") or "permute_" in e[2] or "state_" in e[2]]
    if len(moves) < 2:        # a boundary of the resident loop: nothing travels
        continue
    print("frame boundary %d: epilogue end -> next prologue start %.1f us" % (shown, (ev[b][0] - ev[a][1]) / 1e3))
    prev = ev[a][1]
    for s, e, name in ev[a + 1:b + 1]:
        print("   +%8.1f us idle | %-60s %8.1f us" % ((s - prev) / 1e3, name[:60], (e - s) / 1e3))
        prev = max(prev, e)
    busy = sum(e - s for s, e, _ in ev[a + 1:b])
    print("   device busy inside the boundary %.1f us, idle %.1f us" % (busy / 1e3, (ev[b][0] - ev[a][1] - busy) / 1e3))
    shown += 1
    if shown >= 3:
        break
if not shown:
    print("no class-API frame boundary found (%d kernel/copy records)" % len(ev))
