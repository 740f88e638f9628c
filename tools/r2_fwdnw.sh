#!/bin/bash
# GPU box: waves per tile in the forward sweep's block kernel by level width (ADMM_HIP_FWD_NW4 / _NW8 = widest supernode for 4 / 8 waves)
cd $GRAFT_REPO_ROOT
for cfg in "0 0" "128 0" "128 400" "200 400" "100 200" "400 0" "0 400" "0 1200" "128 1200"; do
  set -- $cfg
  ADMM_HIP_FWD_NW4=$1 ADMM_HIP_FWD_NW8=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_summary.py "nw4<=$1,nw8<=$2"
done
