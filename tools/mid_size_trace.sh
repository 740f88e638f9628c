#!/bin/bash
# per-launch durations of one ADMM iteration on a mid-size bar (kernel trace under rocprofv3)
# usage: bash tools/mid_size_trace.sh nx ny nz [VAR=value ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
nx=$1; ny=$2; nz=$3; shift 3
for v in "$@"; do export "$v"; done
rm -rf /tmp/walk_trace
timeout 300 rocprofv3 --kernel-trace -d /tmp/walk_trace -o t --output-format csv -- python3 tools/run_steps.py $nx $ny $nz 3 > /dev/null 2>&1
python3 tools/level_trace.py /tmp/walk_trace
