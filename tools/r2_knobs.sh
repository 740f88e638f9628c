#!/bin/bash
# GPU box: A/B of the sweep kernels' compile-time knobs (3 timed frames each, phases in ms per ADMM iteration)
cd $GRAFT_REPO_ROOT
bash tools/bench_variant.sh base ""
bash tools/bench_variant.sh bwdU8 "-DADMM_BWD_UNROLL=8"
bash tools/bench_variant.sh bwdU2 "-DADMM_BWD_UNROLL=2"
bash tools/bench_variant.sh fwdD4 "-DADMM_FWD_DEPTH=4"
bash tools/bench_variant.sh fwdD12 "-DADMM_FWD_DEPTH=12"
bash tools/bench_variant.sh base2 ""
