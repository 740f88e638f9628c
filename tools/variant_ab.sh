#!/bin/bash
# usage (GPU box): tools/variant_ab.sh "<scene indices of tools/probe/tree_policy_ab.py>" "name:flags" ...   -- builds library variants (extra hipcc flags) and
# prints wall / forward / backward / local per ADMM iteration for each, the shipped library first
cd $GRAFT_REPO_ROOT
scenes=$1; shift
echo "== shipped"
WALK_AB_SCENES=$scenes python tools/probe/tree_policy_ab.py 2>&1 | grep " off "
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python - <<PY > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package()
pkg._build.build(extra_hip_flags="$flags".split(), out="admm-elastic-sca_amd/_build/libadmm_hip_$name.so", tag="_$name")
PY
  echo "== $name ($flags)"
  WALK_AB_SCENES=$scenes ADMM_HIP_LIB=$GRAFT_REPO_ROOT/admm-elastic-sca_amd/_build/libadmm_hip_$name.so python tools/probe/tree_policy_ab.py 2>&1 | grep " off "
done
