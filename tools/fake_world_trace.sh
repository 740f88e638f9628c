# GPU box: per-launch picture of ONE rank of an 8-rank subtree-sharded run (no-op all-reduce) -> gpurun_out/level_trace_fake8.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BENCH_TIMING_EXPERIMENT=1 ADMM_BENCH_FAKE_WORLD=${1:-8}
rm -rf /tmp/pt_fake
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_fake -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1)
python tools/level_trace.py /tmp/pt_fake > gpurun_out/level_trace_fake${ADMM_BENCH_FAKE_WORLD}.txt 2>&1
tail -50 gpurun_out/level_trace_fake${ADMM_BENCH_FAKE_WORLD}.txt
