cd $GRAFT_REPO_ROOT
bash tools/bench_variant.sh w1 "-DADMM_FWD_SMALL_WAVES=1"
bash tools/bench_variant.sh w4 ""
bash tools/bench_variant.sh w2 "-DADMM_FWD_SMALL_WAVES=2"
bash tools/bench_variant.sh w1 "-DADMM_FWD_SMALL_WAVES=1"
