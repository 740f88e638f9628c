#!/bin/bash
# GPU box: why do the closed-form kinds (hinges, strain triangles) of the mixed scene take 16-25 us per launch for 28-55 MB?  Per-batch launches
# (ADMM_HIP_LOCAL_MULTI=0; every pass under its own 90 s timeout: some counter groups make the profiler crawl), one counter group per rocprofv3 pass -> gpurun_out/pmc_cheap_kinds.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp ADMM_HIP_LOCAL_MULTI=0
groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE")
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/ck_$i
  (cd /tmp && timeout 90 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/ck_$i -- python3 $GRAFT_REPO_ROOT/tools/run_mixed.py 1 > /tmp/ck_$i.log 2>&1) || { echo "pass $i ($g) failed"; tail -3 /tmp/ck_$i.log; }
  i=$((i+1))
done
python3 - <<'PY' | tee gpurun_out/pmc_cheap_kinds.txt
import csv, glob
res = {}; dur = {}
for f in glob.glob("/tmp/ck_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("admm_dev::", "")
        if not k.startswith("project_"): continue
        e = res.setdefault(k, {}).setdefault(r["Counter_Name"], {})
        d = r.get("Dispatch_Id", "0")
        e[d] = e.get(d, 0.0) + float(r["Counter_Value"])
for f in glob.glob("/tmp/ck_0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("admm_dev::", "")
        if k.startswith("project_"): dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(res):
    ds = dur.get(k, [0.0])
    print("%s: %d launches, %.1f us per launch (under the counter pass)" % (k, len(ds), sum(ds) / len(ds)))
    for c in sorted(res[k]):
        v = sum(res[k][c].values()) / max(len(res[k][c]), 1)
        print("    %-40s %16.1f per launch" % (c, v))
PY
