#!/bin/bash
# round 4, GPU session E: leaner log, region counters, EPL A/B, class-API timeline, PMC refresh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python tools/tet_phase_profile.py frames=3 > $O/tet_phase_profile.txt 2>&1; tail -8 $O/tet_phase_profile.txt
timeout 900 python tools/probe/lib_ab.py scene=mixed reps=3 "epl2=" "epl1=-DADMM_MULTI_EPL=1" > $O/epl_ab.txt 2>&1; cat $O/epl_ab.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python3 tools/bench_summary.py r4e < $O/bench.json
for z in 1 0; do
  rm -rf /tmp/capi$z
  (cd /tmp && ADMM_HIP_STATE_ZEROCOPY=$z rocprofv3 --kernel-trace --output-format csv -d /tmp/capi$z -- python3 $GRAFT_REPO_ROOT/tools/probe/class_api_cost.py 32 32 163 trace > /tmp/capi$z.log 2>&1)
  echo "== ADMM_HIP_STATE_ZEROCOPY=$z"; python3 tools/class_api_timeline.py /tmp/capi$z
done > $O/class_api_timeline.txt 2>&1; head -40 $O/class_api_timeline.txt
bash tools/pmc_collect.sh > $O/pmc_collect.log 2>&1; cp gpurun_out/pmc_1M.json $O/pmc_1M.json; tail -3 $O/pmc_collect.log
bash tools/pmc_instmix.sh > $O/tet_instmix.txt 2>&1; tail -30 $O/tet_instmix.txt
