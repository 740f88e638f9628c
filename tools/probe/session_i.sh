#!/bin/bash
# GPU box: per-level durations of the forward sweep for load-group depths of the 16-wave tiles (variant builds, kernel trace of one iteration each, twice)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
import os
for d in (2, 3, 4, 5, 8):
    pkg._build.build(force=False, extra_hip_flags=["-DADMM_FWD_DEPTH16=%d" % d], out=os.path.join("admm-elastic-sca_amd", "_build", "libadmm_hip_d%d.so" % d), tag="_d%d" % d)
PY
for rep in 1 2; do
for d in 8 2 3 4 5; do
  rm -rf /tmp/tr_$d
  (cd /tmp && ADMM_HIP_LIB=$OLDPWD/admm-elastic-sca_amd/_build/libadmm_hip_d$d.so ADMM_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$d -- python3 $OLDPWD/tools/run_steps.py 32 32 163 2 > /dev/null 2>&1)
  echo "== depth16 $d (rep $rep)"; python3 tools/level_trace.py /tmp/tr_$d | grep "big_kernel<true, 16>\|solve_fwd_big_kernel  \|root_product" | head -6
done
done > gpurun_out/i_depth_levels.txt 2>&1
cat gpurun_out/i_depth_levels.txt
