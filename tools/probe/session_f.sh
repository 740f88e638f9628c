#!/bin/bash
# GPU box: does the one-case step selection pay?  parity first, then A/B of the two builds on the headline bar, a shard-sized bar and StVK
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "local or tet or project or hyper or fewer" > gpurun_out/f_parity.txt 2>&1; tail -3 gpurun_out/f_parity.txt
for sc in bar:32x32x163 bar:16x16x82 bar:32x32x163:TET_STVK; do
  echo "== $sc"
  timeout 900 python tools/probe/lib_ab.py scene=$sc reps=3 "uniform=" "select=-DADMM_CSTEP_UNIFORM=0"
done > gpurun_out/f_ab.txt 2>&1
cat gpurun_out/f_ab.txt
