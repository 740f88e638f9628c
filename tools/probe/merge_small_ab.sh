#!/bin/bash
# GPU box: ADMM_HIP_MERGE_SMALL at the 1M-tet bar on one GPU (regions of at most that many nodes become four-way tree nodes: one level less near the leaves, more fill)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 150 300 600 1200 2500; do
  ADMM_HIP_MERGE_SMALL=$v python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python3 -c "
import sys, json
o = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p = o['roofline']['phases_ms_per_iter']
print('MERGE_SMALL=%-5s rep $rep: %.4f ms/iter (local %.4f fwd %.4f bwd %.4f) levels %d nnz_L %.1fM' % ('$v', o['ms_per_step']/20, p['local_ms'], p['solve_fwd_ms'], p['solve_bwd_ms'], o['config']['levels'], o['config']['nnz_L']/1e6))"
done; done
