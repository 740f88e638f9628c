# GPU box: per-rank kernel time of an N-rank subtree-sharded run (no-op all-reduce) under elimination-tree policy variants
# usage: bash tools/fake_world_policy.sh "<world sizes>" "VAR=value,VAR=value" ...   (first variant: library defaults)
cd $GRAFT_REPO_ROOT
worlds=$1; shift
for w in $worlds; do for v in "" "$@"; do
  ( for kv in ${v//,/ }; do export $kv; done
    BENCH_TIMING_EXPERIMENT=1 ADMM_BENCH_FAKE_WORLD=$w python bench.py --no-cpu-baseline --no-extras --shard subtree --steps 3 --warmup 1 2>/dev/null | python3 tools/bench_summary.py "world$w-${v:-default}" )
done; done
