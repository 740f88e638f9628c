# GPU box: bench.py end to end with N ranks sharing the one GPU (gloo instead of RCCL), checksums against the single-rank run
cd $GRAFT_REPO_ROOT
ARGS="--steps 2 --warmup 1 --dims 16 16 65 --no-cpu-baseline --no-extras"
python bench.py --gpus 1 $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1', d['config']['x_checksum'])"
for N in 2 4 8; do for M in subtree contiguous; do
ADMM_BENCH_SHARE_GPU=1 ADMM_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --shard $M $ARGS 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$N $M', d['config']['x_checksum'], d['n_gpus'], d['config']['parallelism'][:40])"
done; done
