#!/bin/bash
# Multi-GPU node: the headline bench at 1 / 2 / 4 / 8 ranks, exactly as the driver launches it (torch.distributed.run, one
# process per GPU, launched BEFORE anything touches a GPU), one JSON line per N into gpurun_out/scale_N<k>.json and a
# per-rank phase table (ms per ADMM iteration) on stdout.
#   tools/run_scale.sh [max_ranks] [extra bench args...]     e.g. tools/run_scale.sh 8 --shard subtree
# Expected from single-GPU measurements of one rank's launch sequence and each rank's real-physics local step (profiles/r04/
# per_rank_kernel_time_fake_world.txt, group_local_times.txt): 0.468 / 0.374 / 0.264 ms per ADMM iteration at 2 / 4 / 8
# ranks + one 0.55 MB all-reduce, i.e. ~1.47 / 1.83 / 2.60 x one GPU (0.686 ms) before communication; contiguous sharding ~1.3 x at 8.
cd "$(dirname "$0")/.."
max=${1:-8}; shift
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  [ $n -gt $max ] && break
  out=gpurun_out/scale_N$n.json
  if [ $n -eq 1 ]; then
    python bench.py --gpus 1 --steps 5 --warmup 1 --no-extras "$@" > $out 2> gpurun_out/scale_N$n.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) \
      bench.py --gpus $n --steps 5 --warmup 1 "$@" > $out 2> gpurun_out/scale_N$n.err
  fi
  python3 - $out $n <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print("N=%s: no result (%r); see the .err file" % (sys.argv[2], e)); sys.exit(0)
print("N=%d  value %.4g  ms/iter %.3f  all-reduce: %s" % (d["n_gpus"], d["value"], d["ms_per_step"] / d["config"]["admm_iters_per_step"], d["config"].get("allreduce")))
for k, v in (d.get("per_rank") or {}).items():
    print("   %-14s %s" % (k, v))
PY
done
