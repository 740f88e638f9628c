#!/bin/bash
# round 4, GPU session D: two elements per lane in the one-launch local step; the class API's frame boundary
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for i in 1 2; do python bench.py --config mixed --no-extras 2>/dev/null | python3 tools/bench_summary.py "mixed EPL2 run $i"; done | tee $O/mixed.txt
for z in 0 1 0 1; do echo "ADMM_HIP_STATE_ZEROCOPY=$z"; ADMM_HIP_STATE_ZEROCOPY=$z python tools/probe/class_api_cost.py 32 32 163 ab; done > $O/class_api_ab.txt 2>&1; cat $O/class_api_ab.txt
for z in 0 1; do
  rm -rf /tmp/capi$z
  (cd /tmp && ADMM_HIP_STATE_ZEROCOPY=$z rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/capi$z -- python3 $GRAFT_REPO_ROOT/tools/probe/class_api_cost.py 32 32 163 trace > /tmp/capi$z.log 2>&1)
  echo "== ADMM_HIP_STATE_ZEROCOPY=$z"; python3 tools/class_api_timeline.py /tmp/capi$z
done > $O/class_api_timeline.txt 2>&1; head -60 $O/class_api_timeline.txt
