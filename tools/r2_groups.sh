#!/bin/bash
# GPU box: concurrent subtree groups on one GPU (ADMM_HIP_GROUPS) -- parity subset, A/B bench, timeline
cd $GRAFT_REPO_ROOT
for g in 2 3; do ADMM_HIP_GROUPS=$g python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "solve or full_size or traj" 2>&1 | tail -1; done
for r in 1 2; do for g in 1 2 3 4; do ADMM_HIP_GROUPS=$g python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_summary.py groups$g; done; done
ADMM_HIP_GROUPS=2 python tools/sweep_timeline.py > gpurun_out/sweep_timeline_g2.txt 2>&1
python tools/sweep_timeline.py > gpurun_out/sweep_timeline_g1.txt 2>&1
tail -3 gpurun_out/sweep_timeline_g1.txt gpurun_out/sweep_timeline_g2.txt
