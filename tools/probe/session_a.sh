#!/bin/bash
# round 4, GPU session A: what bounds under-filled tet launches?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
./tools/probe/_build/valu_latency > $O/valu_latency.txt 2>&1
ADMM_HIP_TPB=16 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_tpb16.log 2>&1; tail -3 $O/pytest_tpb16.log
timeout 600 python tools/probe/underfilled.py timeline dims=13x13x50 kind=TET_STVK frames=14 > $O/timeline_c2.txt 2>&1
timeout 600 python tools/probe/underfilled.py timeline dims=13x13x50 kind=TET_STVK frames=14 tpb=16 >> $O/timeline_c2.txt 2>&1
timeout 600 python tools/probe/underfilled.py timeline dims=16x16x81 kind=TET_NH frames=3 > $O/timeline_125k.txt 2>&1
timeout 600 python tools/probe/underfilled.py timeline dims=10x10x9 kind=TET_NH frames=5 > $O/timeline_5k.txt 2>&1
timeout 900 python tools/probe/underfilled.py tpb dims=13x13x50 kind=TET_STVK frames=14 > $O/tpb.txt 2>&1
timeout 900 python tools/probe/underfilled.py tpb dims=10x10x9 kind=TET_NH frames=8 >> $O/tpb.txt 2>&1
timeout 900 python tools/probe/underfilled.py tpb dims=16x16x81 kind=TET_NH frames=6 >> $O/tpb.txt 2>&1
# the same with the compiler scheduling for ILP (a lone wave cannot hide its own dependent-issue latency)
python - <<'PY' > $O/build_ilp.log 2>&1
import os, sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
out = os.path.abspath("admm-elastic-sca_amd/_build/libadmm_hip_ilp.so")
pkg._build.build(force=False, extra_hip_flags=["-mllvm", "-amdgpu-sched-strategy=max-ilp"], out=out, tag="_ilp")
PY
timeout 900 python tools/probe/underfilled.py tpb dims=13x13x50 kind=TET_STVK frames=14 lib=$PWD/admm-elastic-sca_amd/_build/libadmm_hip_ilp.so > $O/tpb_ilp.txt 2>&1
cat $O/valu_latency.txt $O/timeline_c2.txt $O/timeline_125k.txt $O/timeline_5k.txt $O/tpb.txt $O/tpb_ilp.txt
