#!/bin/bash
# usage: tools/pmc_variant.sh <name> "<extra hipcc flags>"   (GPU box) -- builds a variant and measures the tet kernel's time + FETCH/WRITE_SIZE at 1M tets
set -e
name=$1; flags=$2
cd $GRAFT_REPO_ROOT
python - <<PY
import sys; sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package()
pkg._build.build(extra_hip_flags="$flags".split(), out="admm-elastic-sca_amd/_build/libadmm_hip_$name.so", tag="_$name")
PY
export ADMM_HIP_LIB=$GRAFT_REPO_ROOT/admm-elastic-sca_amd/_build/libadmm_hip_$name.so
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pv_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pv_$c -- python3 $GRAFT_REPO_ROOT/tools/run_steps.py 32 32 163 1 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("/tmp/pv_%s/*/*counter_collection.csv"%c)[0]
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "project_tet" in r["Kernel_Name"] and r["Counter_Name"]==c]
    res[c]=sum(v)/len(v)/1024
f=glob.glob("/tmp/pv_WRITE_SIZE/*/*kernel_trace.csv")[0]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "project_tet" in r["Kernel_Name"]]
print("$name", "tet kernel us %.1f"%(sum(d)/len(d)), "FETCH MB %.0f WRITE MB %.0f"%(res["FETCH_SIZE"],res["WRITE_SIZE"]))
PY
