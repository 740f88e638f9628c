#!/bin/bash
# usage (GPU box): tools/walk_variants.sh "<scenes>" "name:flags" ...   -- builds library variants and runs tools/probe/tree_policy_ab.py with each
cd $GRAFT_REPO_ROOT
scenes=$1; shift
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python - <<PY > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package()
pkg._build.build(extra_hip_flags="$flags".split(), out="admm-elastic-sca_amd/_build/libadmm_hip_$name.so", tag="_$name")
PY
  echo "== $name ($flags)"
  WALK_AB_SCENES=$scenes ADMM_HIP_LIB=$GRAFT_REPO_ROOT/admm-elastic-sca_amd/_build/libadmm_hip_$name.so python tools/probe/tree_policy_ab.py 2>&1 | grep -v "^admm_hip\|amdgpu.ids" | grep " on "
done
