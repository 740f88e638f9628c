#!/bin/bash
# GPU box: the round's judged artefacts in one go -> gpurun_out/{bench_1M.json, bench_1M_under_rocprof.json, bench_1M_kernel_stats.csv, pmc_1M.json}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_1M.json 2> gpurun_out/bench_1M.err
tail -1 gpurun_out/bench_1M.json | cut -c1-400
export TMPDIR=/tmp
rm -rf /tmp/prof_stats
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_1M_under_rocprof.json 2>/dev/null)
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/bench_1M_kernel_stats.csv
head -12 gpurun_out/bench_1M_kernel_stats.csv | cut -c1-160
bash tools/pmc_collect.sh
