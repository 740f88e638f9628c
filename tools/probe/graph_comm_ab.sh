cd $GRAFT_REPO_ROOT
for w in 4 8; do for g in "0 0" "1 1"; do set -- $g
  ADMM_HIP_GRAPH=$1 ADMM_HIP_GRAPH_COMM=$2 BENCH_TIMING_EXPERIMENT=1 ADMM_BENCH_FAKE_WORLD=$w ADMM_BENCH_FAKE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$w \
    bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --timing-stride 1000 2>/dev/null | python3 -c "
import sys, json
o = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('fake world $w, rank 0, ADMM_HIP_GRAPH=$1 GRAPH_COMM=$2: ms_per_step %.3f -> %.4f ms per iteration; graph_state %s' % (o['ms_per_step'], o['ms_per_step'] / 20, o.get('graph_state')))"
done; done
