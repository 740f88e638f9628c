#!/bin/bash
# GPU box: per-kernel PMC counters of one frame (20 ADMM iterations) of the 1M-tet bar, one counter group per rocprofv3 pass
# (kernel-trace + pmc only, as the pool requires).  Writes gpurun_out/pmc_1M.json (copy to profiles/<round>/).
# usage: tools/pmc_collect.sh [nx ny nz]      (PMC_MIXED=1: the mixed scene of BASELINE configs[4] instead -> gpurun_out/pmc_mixed.json)
cd $GRAFT_REPO_ROOT
dims=${@:-32 32 163}
export TMPDIR=/tmp
if [ -n "$PMC_MIXED" ]; then prog="run_mixed.py"; else prog="run_steps.py $dims"; fi
groups=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64")
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pmc_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_$i -- python3 $GRAFT_REPO_ROOT/tools/$prog 1 > /tmp/pmc_$i.log 2>&1) || { echo "pass $i ($g) failed"; tail -5 /tmp/pmc_$i.log; }
  i=$((i+1))
done
python3 - "$dims" "${PMC_MIXED:-}" <<'PY'
import csv, glob, json, os, sys
res = {}
for d in sorted(glob.glob("/tmp/pmc_[0-9]*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            c = r["Counter_Name"]
            e = res.setdefault(k, {}).setdefault(c, [0.0, {}])
            disp = r.get("Dispatch_Id", r.get("Correlation_Id", "0"))
            e[1][disp] = e[1].get(disp, 0.0) + float(r["Counter_Value"])   # sum over XCDs / instances of one dispatch
out = {}
for k, cs in res.items():
    out[k] = {c: {"per_launch": sum(v[1].values()) / max(len(v[1]), 1), "launches": len(v[1])} for c, v in cs.items()}
def _src_hash():
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "admm-elastic-sca_amd", "build.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m.source_hash()
mixed = len(sys.argv) > 2 and sys.argv[2]
doc = {"workload": ("mixed scene of BASELINE configs[4] (26x26x123 bar, half NH half StVK tets, 158x158 cloth), 1 frame x 20 ADMM iterations, 1 MI355X" if mixed else "NH bar %s cubes, 1 frame x 20 ADMM iterations, 1 MI355X" % sys.argv[1]),
       "method": "rocprofv3 --kernel-trace --pmc <group>, one counter group per pass, command: python3 tools/run_steps.py <dims> 1 (tools/pmc_collect.sh); values are per launch (mean over launches, summed over a dispatch's instances). FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3 reports them; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, both the raw and the x2 figure are quoted in DESIGN.md.",
       "csrc_sha256": _src_hash(),      # the sources the counters were collected on (bench.py compares it with the tree it runs)
       "kernels": out}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(doc, open("gpurun_out/pmc_mixed.json" if mixed else "gpurun_out/pmc_1M.json", "w"), indent=1)
for k, v in out.items():
    print(k[:48], {c: round(x["per_launch"], 1) for c, x in v.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES")})
PY
