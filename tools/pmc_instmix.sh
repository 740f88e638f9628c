#!/bin/bash
# GPU box: dynamic instruction mix of the tet kernel at the 1M-tet bar (one frame, per-launch means) -> gpurun_out/tet_instmix.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
groups=("SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_INSTS_FLAT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC")
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/im_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/im_$i -- python3 $GRAFT_REPO_ROOT/tools/run_steps.py 32 32 163 1 > /tmp/im_$i.log 2>&1) || echo "pass $i failed"
  i=$((i+1))
done
python3 - <<'PY' | tee gpurun_out/tet_instmix.txt
import csv, glob
res = {}
for f in glob.glob("/tmp/im_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "project_tet" not in r["Kernel_Name"]: continue
        e = res.setdefault(r["Counter_Name"], {})
        d = r.get("Dispatch_Id", "0")
        e[d] = e.get(d, 0.0) + float(r["Counter_Value"])
tot = None
for k in sorted(res):
    v = sum(res[k].values()) / max(len(res[k]), 1)
    if k == "SQ_INSTS_VALU": tot = v
for k in sorted(res):
    v = sum(res[k].values()) / max(len(res[k]), 1)
    print("%-28s %14.0f per launch  %8.1f per wave (15648 waves)%s" % (k, v, v / 15648.0, ("   %.1f %% of VALU" % (100 * v / tot)) if tot and k.startswith("SQ_INSTS_VALU_") else ""))
PY
