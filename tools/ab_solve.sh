cd $GRAFT_REPO_ROOT
bash tools/bench_variant.sh b64 "-DADMM_LOCAL_BLOCK=64"
bash tools/bench_variant.sh b256 ""
bash tools/bench_variant.sh b64 "-DADMM_LOCAL_BLOCK=64"
bash tools/bench_variant.sh b256 ""
