# GPU box: per-rank kernel time of an N-rank run (no-op all-reduce; communication excluded), both sharding modes
cd $GRAFT_REPO_ROOT
for w in 2 4 8; do for m in subtree contiguous; do
  BENCH_TIMING_EXPERIMENT=1 ADMM_BENCH_FAKE_WORLD=$w python bench.py --no-cpu-baseline --no-extras --shard $m --steps 3 --warmup 1 2>/dev/null | python3 tools/bench_summary.py "world$w-$m-rank0"
done; done
BENCH_TIMING_EXPERIMENT=1 ADMM_BENCH_FAKE_WORLD=8 ADMM_BENCH_FAKE_RANK=5 python bench.py --no-cpu-baseline --no-extras --shard subtree --steps 3 --warmup 1 2>/dev/null | python3 tools/bench_summary.py "world8-subtree-rank5"
