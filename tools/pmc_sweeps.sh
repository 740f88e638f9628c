#!/bin/bash
# GPU box: what the sweep kernels wait for before their bytes count -- instruction cache, address translation, scalar cache
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for g in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/sw_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/sw_$i -- python3 $GRAFT_REPO_ROOT/tools/run_steps.py 32 32 163 1 > /tmp/sw_$i.log 2>&1) || { echo "pass $i ($g) failed"; tail -3 /tmp/sw_$i.log; }
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/sw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("solve_fwd_small", "solve_fwd_big_kernel<true, 4>", "solve_fwd_big_kernel<true, 8>", "solve_fwd_big_kernel<true, 16>", "solve_bwd_kernel<1, 8>", "solve_bwd_kernel<2, 8>", "solve_bwd_kernel<4, 4>", "root_product", "rhs_gather"):
            if key in k:
                res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in res.items():
    print(key)
    for c, v in sorted(d.items()):
        print("    %-28s per launch %.4g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
