#!/bin/bash
# GPU box: does dealing MORE, smaller subtrees to every rank (ADMM_HIP_SUBTREES_PER_RANK) balance the ranks' local steps, and what does the
# larger replicated top cost?  Per setting: each rank's real-physics local step (tools/group_local_times.sh) and one rank's launch sequence
# with a no-op all-reduce (bench.py fake world) at 4 and 8 ranks.
cd $GRAFT_REPO_ROOT
for k in 1 2 4; do
  export ADMM_HIP_SUBTREES_PER_RANK=$k
  echo "=== ADMM_HIP_SUBTREES_PER_RANK=$k"
  for w in 4 8; do
    bash tools/group_local_times.sh $w
    ADMM_HIP_VERBOSE=1 BENCH_TIMING_EXPERIMENT=1 ADMM_BENCH_FAKE_WORLD=$w python bench.py --no-cpu-baseline --no-extras --shard subtree --steps 3 --warmup 1 2> /tmp/sub_${k}_${w}.err | python3 tools/bench_summary.py "world$w-subtree-rank0"
    grep "subtree sharding" /tmp/sub_${k}_${w}.err | head -1
  done
done
