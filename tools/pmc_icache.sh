#!/bin/bash
# GPU box: instruction-cache behaviour of the tet kernel (one frame of the 1M-tet bar)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for g in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/ic_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/ic_$i -- python3 $GRAFT_REPO_ROOT/tools/run_steps.py 32 32 163 1 > /tmp/ic_$i.log 2>&1) || { echo "pass $i failed"; tail -3 /tmp/ic_$i.log; }
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob
res = {}
for f in glob.glob("/tmp/ic_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "project_tet" not in r["Kernel_Name"]: continue
        res.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(res.items()):
    print("%-24s per launch %.4g" % (k, sum(v) / 20.0))
PY
