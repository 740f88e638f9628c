#!/bin/bash
# GPU box: backward sweep over the replicated top only where a rank reads it -- sharding parity, then one rank's kernel times of 2 / 4 / 8-rank runs both ways
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_sharding.py tests/test_cpp_host.py -x -q -m gpu > gpurun_out/k_parity.txt 2>&1; tail -3 gpurun_out/k_parity.txt
for all in 0 1; do
  echo "== ADMM_HIP_TOP_BWD_ALL=$all"
  for w in 2 4 8; do for rk in 0 $((w-1)) $((w/2)); do
    ADMM_HIP_TOP_BWD_ALL=$all BENCH_TIMING_EXPERIMENT=1 ADMM_BENCH_FAKE_WORLD=$w ADMM_BENCH_FAKE_RANK=$rk timeout 300 python bench.py --no-cpu-baseline --no-extras --shard subtree --steps 3 --warmup 1 2>/dev/null | python3 tools/bench_summary.py "world$w-subtree-rank$rk"
  done; done
done > gpurun_out/k_top_bwd.txt 2>&1
cat gpurun_out/k_top_bwd.txt
