#!/bin/bash
# GPU box: equal-area forward tiles + root rows that fill the CUs once: global-step parity, then A/B against the plain tiling (run-time knobs)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_sharding.py -x -q -m gpu > gpurun_out/h_parity.txt 2>&1; tail -3 gpurun_out/h_parity.txt
for sc in bar:32x32x163 bar:24x24x120 bar:16x16x82 mixed; do
  echo "== $sc"
  timeout 900 python tools/probe/lib_ab.py scene=$sc reps=3 "even=" "plain=ADMM_HIP_FWD_EVEN_MAX=0 ADMM_HIP_ROOT_FILL=0" "tiles_only=ADMM_HIP_ROOT_FILL=0"
done > gpurun_out/h_ab.txt 2>&1
cat gpurun_out/h_ab.txt
