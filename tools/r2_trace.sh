#!/bin/bash
# GPU box: per-launch picture of one ADMM iteration (rocprofv3 kernel trace of 1 frame) -> gpurun_out/r2_level_trace_<tag>.txt
cd $GRAFT_REPO_ROOT
tag=${1:-cur}
export TMPDIR=/tmp
rm -rf /tmp/prof_trace
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -- python3 $GRAFT_REPO_ROOT/tools/run_steps.py 32 32 163 1 > /dev/null 2>&1)
python tools/level_trace.py /tmp/prof_trace > gpurun_out/r2_level_trace_$tag.txt 2>&1
cat gpurun_out/r2_level_trace_$tag.txt
