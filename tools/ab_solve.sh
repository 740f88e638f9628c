cd $GRAFT_REPO_ROOT
BENCH_TIMING_EXPERIMENT=1 bash tools/bench_variant.sh fake "-DADMM_BWD_FAKE_REDUCE=1" --steps 1 --warmup 0
