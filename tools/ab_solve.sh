cd $GRAFT_REPO_ROOT
bash tools/bench_variant.sh base "" 
bash tools/bench_variant.sh fd16 "-DADMM_FWD_DEPTH=16"
bash tools/bench_variant.sh fd4 "-DADMM_FWD_DEPTH=4"
bash tools/bench_variant.sh bu8 "-DADMM_BWD_UNROLL=8"
bash tools/bench_variant.sh bu2 "-DADMM_BWD_UNROLL=2"
bash tools/bench_variant.sh cw2 "-DADMM_BWD_BIG_CW=2"
bash tools/bench_variant.sh cw4 "-DADMM_BWD_BIG_CW=4"
