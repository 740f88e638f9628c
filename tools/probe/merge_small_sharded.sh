#!/bin/bash
# GPU box: ADMM_HIP_MERGE_SMALL (four-way tree nodes inside a rank's subtrees) under subtree sharding at sizes where the sweeps are byte-bound.
# The default (regions up to 4/3 of a rank's share become four-way nodes) was tuned on the 1M-tet bar, where a level's latency is what counts.
#   usage: tools/probe/merge_small_sharded.sh nx ny nz "<worlds>" "<merge_small values; 'default' = unset>"
cd ${GRAFT_REPO_ROOT:-.}
NX=$1; NY=$2; NZ=$3
for w in $4; do for m in $5; do
  if [ "$m" = default ]; then unset ADMM_HIP_MERGE_SMALL; else export ADMM_HIP_MERGE_SMALL=$m; fi
  timeout 1500 python tools/ranks_one_gpu.py --world $w --dims $NX $NY $NZ --warm 2 --frames 1 --json 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=o['per_frame'][0]
print('bar ${NX}x${NY}x${NZ} ranks %d merge_small %-8s factor whole %.3f GB resident/rank %.3f-%.3f  fwd %.4f bwd %.4f local %.4f  critical path %.4f ms' % (o['world'], '$m', o['factor_gb_whole'], min(o['factor_gb_resident']), max(o['factor_gb_resident']), max(r['solve_fwd_ms']), max(r['solve_bwd_ms']), max(r['local_ms']), r['critical_path_ms']))"
done; done
