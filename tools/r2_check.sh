#!/bin/bash
# GPU box: round-2 bring-up of bench.py -- the default line (with class_api / other_configs / cpu_baseline), then the 8-rank
# launch sequence of ONE rank with a real (1-rank) RCCL communicator inside the library: eager, and captured as a HIP graph.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r2_bench.json; tail -5 gpurun_out/r2_bench.err
for g in 0 1; do
  BENCH_TIMING_EXPERIMENT=1 ADMM_HIP_GRAPH_COMM=$g ADMM_BENCH_FAKE_WORLD=8 ADMM_BENCH_FAKE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/r2_fake8_graph$g.json 2> gpurun_out/r2_fake8_graph$g.err
  echo "fake8 graph_comm=$g rc=$?"; tail -c 1200 gpurun_out/r2_fake8_graph$g.json; grep -v "^$" gpurun_out/r2_fake8_graph$g.err | tail -5
done
