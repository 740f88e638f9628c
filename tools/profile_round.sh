#!/bin/bash
# GPU box: the round's judged artefacts in one go -> gpurun_out/{pmc_1M.json, bench_1M.json, bench_1M_under_rocprof.json, bench_1M_kernel_stats.csv,
# level_trace_1M.txt, bench_mixed.json}; copy them into profiles/rNN/ afterwards.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# 1. PMC passes first: bench.py quotes roofline.traffic / roofline.valu from profiles/r01/pmc_1M.json, so refresh that file (on this box) before the bench line is made
bash tools/pmc_collect.sh
cp gpurun_out/pmc_1M.json profiles/r01/pmc_1M.json
# 2. the bench line
python bench.py > gpurun_out/bench_1M.json 2> gpurun_out/bench_1M.err
tail -1 gpurun_out/bench_1M.json | cut -c1-400
# 3. the same command under rocprofv3 --kernel-trace --stats
rm -rf /tmp/prof_stats
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_1M_under_rocprof.json 2>/dev/null)
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/bench_1M_kernel_stats.csv
head -12 gpurun_out/bench_1M_kernel_stats.csv | cut -c1-160
# 4. per-level picture of one ADMM iteration + the mixed scene
ADMM_HIP_VERBOSE=1 python tools/run_steps.py 32 32 163 1 2>&1 | grep "admm_hip: level" > gpurun_out/level_trace_1M.txt
python tools/level_trace.py /tmp/prof_stats >> gpurun_out/level_trace_1M.txt 2>&1
tail -12 gpurun_out/level_trace_1M.txt
python bench.py --config mixed > gpurun_out/bench_mixed.json 2>/dev/null
tail -1 gpurun_out/bench_mixed.json | cut -c1-200
# 5. where a tet-kernel wave spends its life (instrumented variant build)
python tools/tet_phase_profile.py frames=2 > gpurun_out/tet_phase_profile.txt 2>&1
tail -11 gpurun_out/tet_phase_profile.txt
