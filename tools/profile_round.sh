#!/bin/bash
# GPU box: the round's judged artefacts in one go -> gpurun_out/prof/{pmc_1M.json, bench_1M.json, bench_1M_under_rocprof.json,
# bench_1M_kernel_stats.csv, level_trace_1M.txt, ranks_one_gpu.txt, ranks_one_gpu_contiguous.txt, bench_mixed.json};
# copy them into profiles/rNN/ afterwards.   usage: tools/profile_round.sh [rNN]
cd $GRAFT_REPO_ROOT
R=${1:-r06}
O=gpurun_out/prof
mkdir -p $O
export TMPDIR=/tmp
# 1. PMC passes first: bench.py quotes roofline.traffic / roofline.valu from profiles/$R/pmc_1M.json (labelled as such in the
#    line), so refresh that file on this box before the bench line is made
bash tools/pmc_collect.sh > $O/pmc_collect.log 2>&1
mkdir -p profiles/$R && cp gpurun_out/pmc_1M.json profiles/$R/pmc_1M.json && cp gpurun_out/pmc_1M.json $O/pmc_1M.json
# 2. the bench line
ADMM_BENCH_PMC=$R/pmc_1M.json python bench.py > $O/bench_1M.json 2> $O/bench_1M.err
tail -1 $O/bench_1M.json | cut -c1-300
# 3. the same command under rocprofv3 --kernel-trace --stats
rm -rf /tmp/prof_stats
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_1M_under_rocprof.json 2>/dev/null)
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/bench_1M_kernel_stats.csv
head -14 $O/bench_1M_kernel_stats.csv | cut -c1-160
# 4. per-level picture of one ADMM iteration (eager launches), the mixed scene
ADMM_HIP_VERBOSE=1 python tools/run_steps.py 32 32 163 1 2>&1 | grep "admm_hip: level" > $O/level_trace_1M.txt
rm -rf /tmp/prof_trace
(cd /tmp && ADMM_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -- python3 $GRAFT_REPO_ROOT/tools/run_steps.py 32 32 163 1 > /dev/null 2>&1)
python tools/level_trace.py /tmp/prof_trace >> $O/level_trace_1M.txt 2>&1
tail -12 $O/level_trace_1M.txt
PMC_MIXED=1 bash tools/pmc_collect.sh >> $O/pmc_collect.log 2>&1; cp gpurun_out/pmc_mixed.json profiles/$R/pmc_mixed.json; cp gpurun_out/pmc_mixed.json $O/pmc_mixed.json
python bench.py --config mixed --no-extras > $O/bench_mixed.json 2>/dev/null
tail -1 $O/bench_mixed.json | cut -c1-200
# 5. what every RANK of a 2 / 4 / 8-rank run computes per iteration, real physics, one rank on the GPU at a time (tools/ranks_one_gpu.py)
for w in 2 4 8; do timeout 300 python tools/ranks_one_gpu.py --world $w --warm 2 --frames 1 2>&1 | grep -v amdgpu.ids; done > $O/ranks_one_gpu.txt
for w in 2 4 8; do timeout 300 python tools/ranks_one_gpu.py --world $w --warm 2 --frames 1 --mode contiguous 2>&1 | grep -v amdgpu.ids; done > $O/ranks_one_gpu_contiguous.txt
cat $O/ranks_one_gpu.txt $O/ranks_one_gpu_contiguous.txt
