"""Multi-GPU path: elements shard across ranks, the partial
right-hand sides are summed with one all-reduce per ADMM iteration, the solve is
replicated.

CPU (gloo, world_size 2): the shard ranges the C library computes, and that the
sum over ranks of the per-rank partial RHS (M x_bar on rank 0 only) is the full
RHS -- computed from the oracle's D, W, z, u and reduced through
torch.distributed.  GPU: the same scene run as two shards on ONE GPU (two
contexts, the all-reduce hook rendezvouses two host threads) reproduces the
unsharded result."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from __graft_entry__ import load_package
from checkers import KIND, Oracle
pkg = load_package()
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dims = (3, 3, 7)
mg = pkg.meshgen
x, t = mg.bar(*dims); m = mg.lumped_tet_mass(x, t, 1000.0)
anchors = mg.bar_anchor_nodes(dims[0], dims[1])
s = pkg.make_bar_system(*dims, device_id=-1, rank=rank, world=world)
s.initialize()
# (1) shard ranges: contiguous, disjoint, covering, reference order preserved
rng = [s.local_range(b) for b in range(2)]
allr = [None] * world
dist.all_gather_object(allr, rng)
for b, n in enumerate((t.shape[0], anchors.size)):
    edges = [r[b] for r in allr]
    assert edges[0][0] == 0 and edges[-1][1] == n and all(edges[i][1] == edges[i + 1][0] for i in range(world - 1)), edges
inf = s.info()
assert inf["rank"] == rank and inf["world"] == world and inf["n_elems_local"] == sum(b - a for a, b in rng)
# (2) partial RHS sums to the full RHS
o = Oracle(); o.settings(0.04, 1)
o.add_nodes(x.ravel(), np.repeat(m, 3)); o.add_forces(KIND["TET_NH"], t, [1e5, 1e5, 5]); o.add_forces(KIND["ANCHOR"], anchors, [-1.0, 1.0]); o.add_gravity([0, -9.8, 0])
assert o.initialize()
o.step()                                  # leaves u, z of the (single) ADMM iteration
rr, cc, vv = o.D_triplets(); W = o.wdiag; q = o.z - o.u
gi = o.global_idx()
rows_of = np.zeros(o.rows, dtype=np.int64)        # force index of every row
for i in range(o.n_forces):
    rows_of[gi[i]:gi[i] + (9 if i < t.shape[0] else 3)] = i
first = [0, t.shape[0]]
mine = np.zeros(o.n_forces, dtype=bool)
for b, (a, e) in enumerate(rng):
    mine[first[b] + a:first[b] + e] = True
sel = mine[rows_of[rr]]
part = np.zeros(3 * x.shape[0])
np.add.at(part, cc[sel], 0.04 ** 2 * vv[sel] * W[rr[sel]] ** 2 * q[rr[sel]])
base = np.repeat(m, 3) * 0.0
full = np.zeros(3 * x.shape[0]); np.add.at(full, cc, 0.04 ** 2 * vv * W[rr] ** 2 * q[rr])
tt = torch.from_numpy(part.copy())
dist.all_reduce(tt, op=dist.ReduceOp.SUM)
assert np.abs(tt.numpy() - full).max() < 1e-9 * max(1.0, np.abs(full).max())
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_sharding_gloo_world2(tmp_path, pkg):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-2000:]
        assert "rank %d ok" % r in out


def test_world_without_hook_is_an_error(pkg):
    s = pkg.System(device_id=-1)
    with pytest.raises(pkg.AdmmHipError):
        s.set_shard(2, 2)          # rank out of range
    s.set_shard(1, 2)
    s.add_nodes(np.zeros(9), np.ones(9))
    s.initialize()
    with pytest.raises(pkg.AdmmHipError):
        s.set_shard(0, 2)          # too late


@pytest.mark.gpu
def test_two_shards_on_one_gpu(pkg):
    import torch
    dims = (5, 4, 11)
    ref = pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"])
    ref.initialize()
    shards = [pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"], rank=r, world=2) for r in range(2)]
    bar = threading.Barrier(2)
    bufs = {}

    class _Ptr:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def make_hook(r):
        def hook(ptr, count, stream):
            torch.cuda.synchronize()
            bufs[r] = torch.as_tensor(_Ptr(ptr, count), device="cuda:0")
            bar.wait()
            if r == 0:                       # "all-reduce": both buffers end up with the sum
                tot = bufs[0] + bufs[1]
                bufs[0].copy_(tot); bufs[1].copy_(tot)
                torch.cuda.synchronize()
            bar.wait()
            return 0
        return hook
    for r, s in enumerate(shards):
        s.set_allreduce(make_hook(r))
        s.initialize()
    assert shards[0].info()["n_elems_local"] + shards[1].info()["n_elems_local"] == ref.info()["n_elems_total"]
    errs = []

    def run(s):
        try:
            s.step(1); s.sync()
        except Exception as e:    # surface failures of a worker thread
            errs.append(e)
    for frame in range(3):
        ref.step(1)
        th = [threading.Thread(target=run, args=(s,)) for s in shards]
        [t.start() for t in th]; [t.join() for t in th]
        assert not errs, errs
        x0, x1, xr = shards[0].m_x, shards[1].m_x, ref.m_x
        assert np.array_equal(x0, x1)                      # replicated solve: identical on every rank
        # = unsharded, up to the order of the fp64 partial sums (first frame: rounding only;
        # later frames: amplified by the truncated prox)
        assert np.abs(x0 - xr).max() < (1e-11 if frame == 0 else 1e-6)


@pytest.mark.gpu
def test_bench_two_ranks_end_to_end(tmp_path):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank, all-reduce hook
    inside the C step loop, barrier + MAX-over-ranks timing, one JSON line from rank 0) -- on a 1-GPU box both ranks share
    cuda:0 and gloo stands in for RCCL (test hooks of bench.py).  Same final positions as the single-rank run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--steps", "2", "--warmup", "1", "--dims", "6", "6", "20", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    env = dict(os.environ, ADMM_BENCH_SHARE_GPU="1", ADMM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                         os.path.join(root, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    lines = [l for l in r2.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["value"] > 0 and "cpu_baseline" not in two
    assert "sharded x2" in two["config"]["parallelism"]
    # the partial sums meet in a different order; the NH bar amplifies that to ~1e-7 over three frames
    assert abs(two["config"]["x_checksum"] - one["config"]["x_checksum"]) < 1e-6 * one["config"]["x_checksum"]
