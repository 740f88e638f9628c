#!/bin/bash
# usage: tools/group_local_times.sh G   (GPU box) -> per-group tet kernel durations of the last frame, real physics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; G=${1:-8}
rm -rf /tmp/glt; (cd /tmp && ADMM_HIP_PIPE=$G timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/glt -- python3 $GRAFT_REPO_ROOT/tools/probe/group_local_times.py > /dev/null 2>&1)
python3 - $G <<'PY'
import csv, glob, sys
G=int(sys.argv[1])
rows=[]
for p in glob.glob("/tmp/glt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "project_tet_kernel" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r.get("Grid_Size",0) or 0)//64))
rows.sort()
last=rows[-20*G:]          # the last frame: 20 iterations x G group launches
per=[[] for _ in range(G)]
for i,(t,d,g) in enumerate(last): per[i%G].append((d,g))
print("G = %d groups; per group: blocks, tet kernel us (mean over the last frame's 20 iterations; min..max)" % G)
for g in range(G):
    ds=[d for d,_ in per[g]]
    print("  group %d: %6d blocks  %7.1f us  (%.1f .. %.1f)" % (g, per[g][0][1], sum(ds)/len(ds), min(ds), max(ds)))
print("  sum over the groups %.1f us" % sum(sum(d for d,_ in per[g])/len(per[g]) for g in range(G)))
PY
