#!/bin/bash
# GPU box: the per-rank tables of DESIGN section 6 at sizes where sharding pays (round 6): what every rank of a 2 / 4 / 8-rank run computes per
# ADMM iteration (tools/ranks_one_gpu.py: N contexts on ONE GPU, real physics, one rank's kernels on the GPU at a time), what it factors and
# keeps (rank-local factorization), next to the one-GPU run of the same bar (bench.py --dims).
#   usage: tools/ranks_tables.sh <tag> <nx> <ny> <nz> [frames]      -> gpurun_out/ranks/ranks_one_gpu_<tag>.txt
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; NX=$2; NY=$3; NZ=$4; FR=${5:-1}
O=gpurun_out/ranks; mkdir -p $O
OUT=$O/ranks_one_gpu_$TAG.txt
echo "== one GPU, bar ${NX}x${NY}x${NZ}: python bench.py --dims $NX $NY $NZ --steps 3 --warmup 2 --no-cpu-baseline --no-extras" > $OUT
timeout 1500 python bench.py --dims $NX $NY $NZ --steps 3 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
L=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=L['config']; p=L['roofline']['phases_ms_per_iter']
print('tets+anchors per GPU: all; nodes %s; nnz(L) %.4g; levels %d; initialize %.2f s (numeric factorization %.3f s on %s)' % (c['workload'].split(' tets, ')[1].split(' nodes')[0], c['nnz_L'], c['levels'], c['initialize_s'], c['factor_numeric_s'], c['factor_numeric_on']))
print('ms per ADMM iteration: total %.4f = local %.4f + rhs %.4f + forward %.4f + backward %.4f;  value %.4g iters/s x tets' % (p['total_ms'], p['local_ms'], p['rhs_ms'], p['solve_fwd_ms'], p['solve_bwd_ms'], L['value']))
" >> $OUT 2>&1
for w in 2 4 8; do
  echo "== $w ranks (subtree shards, rank-local factorization; ADMM_HIP_DIST_TOP=${ADMM_HIP_DIST_TOP:-auto})" >> $OUT
  timeout 2400 python tools/ranks_one_gpu.py --world $w --dims $NX $NY $NZ --warm 2 --frames $FR 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
