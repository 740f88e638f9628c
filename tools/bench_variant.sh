#!/bin/bash
# usage: tools/bench_variant.sh <name> "<extra hipcc flags>" [bench args]  (GPU box)
name=$1; flags=$2; shift 2
cd $GRAFT_REPO_ROOT
python - <<PY
import sys; sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package()
pkg._build.build(extra_hip_flags="$flags".split(), out="admm-elastic-sca_amd/_build/libadmm_hip_$name.so", tag="_$name")
PY
ADMM_HIP_LIB=$GRAFT_REPO_ROOT/admm-elastic-sca_amd/_build/libadmm_hip_$name.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python tools/bench_summary.py $name
