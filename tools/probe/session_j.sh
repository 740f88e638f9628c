#!/bin/bash
# GPU box: depth of the 16-wave forward tiles on other scenes (variant builds alternated)
cd "$(dirname "$0")/../.."
for sc in bar:24x24x120 bar:16x16x82 bar:13x13x50:TET_STVK mixed bar:32x32x163; do
  echo "== $sc"
  timeout 900 python tools/probe/lib_ab.py scene=$sc reps=2 "d16_4=" "d16_8=-DADMM_FWD_DEPTH16=8" "d16_2=-DADMM_FWD_DEPTH16=2" "d16_3=-DADMM_FWD_DEPTH16=3"
done > gpurun_out/j_depth16.txt 2>&1
cat gpurun_out/j_depth16.txt
