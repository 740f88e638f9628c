"""The in-library RCCL path (csrc/comm.cpp, launch.inc's subtree exchange, the capturable multi-GPU iteration) under pytest.

A 1-GPU box cannot host a 2-rank RCCL communicator (RCCL refuses two ranks on one device), so these tests drive the REAL
code path -- admm_hip_rccl_unique_id -> admm_hip_rccl_init -> ncclAllReduce on the solver's stream from the C step loop --
through a communicator of ONE rank, whose all-reduce is the identity:

  * the communicator's life cycle (create, checked all-reduce, replace, destroy, again; no device memory left behind);
  * one rank's share of an 8-rank partition ("fake world": shard = (rank r, world 8), all-reduce through the 1-rank
    communicator) -- the launch sequence, buffers and counts of an 8-GPU run with the sum replaced by the identity.  A no-op
    hook is the same identity, so the frames must be BITWISE those of the hook path, for both sharding modes, with eager
    launches and with the iteration captured as a HIP graph around ncclAllReduce (ADMM_HIP_GRAPH_COMM=1);
  * bench.py exactly as tools/profile_round.sh ran it by hand: a fresh `torch.distributed.run` child, 1-rank NCCL process
    group, the ncclUniqueId broadcast through it, the library's own communicator, both sharding modes, eager vs captured.

Every GPU piece runs in a child process under a time-out: a hung collective fails the test instead of the session.
What the reference does here: nothing (single process); the loop being distributed is System.cpp:51-67.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch
from __graft_entry__ import load_package
pkg = load_package()
assert torch.cuda.is_available()
torch.cuda.set_device(0)
'''

LIFECYCLE = PRELUDE + r'''
def cycle():
    s = pkg.make_bar_system(4, 4, 8)
    assert s.rccl_async_error() == 0                      # no communicator: healthy by definition
    uid = s.rccl_unique_id()
    assert uid.any()
    s.rccl_init(uid, 0, 1)
    t = torch.arange(4096, dtype=torch.float64, device="cuda") * 0.37 - 5.0
    want = t.clone()
    torch.cuda.synchronize()
    for _ in range(3):
        s.debug_allreduce(t.data_ptr(), t.numel())         # one rank: the sum is the value itself, bit for bit
    assert torch.equal(t, want)
    assert s.rccl_async_error() == 0
    s.rccl_init(s.rccl_unique_id(), 0, 1)                  # a second communicator replaces (and destroys) the first
    s.debug_allreduce(t.data_ptr(), t.numel())
    assert torch.equal(t, want)
    s.set_rccl_comm(None)                                  # back to "no transport": the all-reduce must now refuse
    try:
        s.debug_allreduce(t.data_ptr(), t.numel())
        raise SystemExit("all-reduce without a transport did not fail")
    except pkg.AdmmHipError as e:
        assert "neither an RCCL communicator nor an all-reduce hook" in str(e), e
    s.rccl_init(s.rccl_unique_id(), 0, 1)
    s.initialize(); s.step(3); s.sync()                    # world 1: no collective in the loop, but the per-frame poll runs
    assert np.isfinite(s.m_x).all() and s.rccl_async_error() == 0
    del s                                                  # admm_hip_destroy -> comm_release -> ncclCommDestroy

cycle(); cycle()      # (the process-wide pools of RCCL / the HIP runtime reach their size in the first two cycles: +176 MB once, measured)
held = []
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for _ in range(4):
    cycle()
    torch.cuda.synchronize()
    held.append(free0 - torch.cuda.mem_get_info()[0])
print("LIFECYCLE ok; device memory held after 1..4 more cycles: %%s bytes" %% held)
assert max(held) < (16 << 20) and held[-1] <= held[0] + (1 << 20), held      # nothing piles up per context / communicator
'''

STEPLOOP = PRELUDE + r'''
mode, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dims = tuple(int(v) for v in sys.argv[4:7])
frames, iters = 3, 10
os.environ["ADMM_HIP_DENSE_MAX"] = "0"        # the panel sweeps (and the subtree partition), not the small-system inverse
os.environ["ADMM_HIP_LEAF"] = "16"            # a deep elimination tree on a small mesh
res = {}
for variant in ("hook", "rccl-eager", "rccl-graph"):
    os.environ["ADMM_HIP_GRAPH_COMM"] = "1" if variant == "rccl-graph" else "0"
    s = pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"], rank=rank, world=world, shard_mode=mode)
    s.set_factor_local(False)      # one rank's share behind a 1-rank communicator: nobody delivers the other subtrees' update matrices
    calls = [0]
    if variant == "hook":
        def noop(ptr, count, strm):
            calls[0] += 1
            return 0
        s.set_allreduce(noop)
    else:
        s.rccl_init(s.rccl_unique_id(), 0, 1)
    s.initialize()
    inf = s.info()
    assert inf["world"] == world and inf["rank"] == rank and 0 < inf["n_elems_local"] < inf["n_elems_total"]
    xs = []
    for f in range(frames):
        s.step(iters)
        xs.append(s.m_x.copy())
    g_before = s.graph_state()
    if variant != "hook":          # a NEW communicator replaces the one the iteration was captured with: the graphs must be captured again, not replayed
        s.rccl_init(s.rccl_unique_id(), 0, 1)
    s.step(iters); xs.append(s.m_x.copy())
    vs = s.m_v.copy()
    if variant == "rccl-graph":
        assert g_before["iter_graph"] and s.graph_state()["iter_graph"] and s.graph_state()["graph_launches"] > g_before["graph_launches"]
    if variant != "hook":
        assert s.rccl_async_error() == 0
        host = np.array([1.5, -2.25, 1e300])
        s.allreduce_host(host)                     # the class mirror's short host vectors go through the same communicator
        assert host.tolist() == [1.5, -2.25, 1e300]
    g = s.graph_state()
    res[variant] = dict(x=xs, v=vs, g=g, calls=calls[0])
    if variant == "hook":
        assert calls[0] >= (frames + 1) * iters and not g["iter_graph"], (calls, g)     # a host hook cannot be captured
    if variant == "rccl-eager":
        assert not g["iter_graph"] and g["graph_launches"] == 0, g
    if variant == "rccl-graph":
        assert g["iter_graph"] and g["graph_launches"] >= frames, g               # ncclAllReduce inside the captured iteration
    del s
for variant in ("rccl-eager", "rccl-graph"):
    for f in range(frames + 1):
        assert np.array_equal(res[variant]["x"][f], res["hook"]["x"][f]), (variant, f, np.abs(res[variant]["x"][f] - res["hook"]["x"][f]).max())
    assert np.array_equal(res[variant]["v"], res["hook"]["v"]), variant
fin = all(np.isfinite(x).all() for x in res["hook"]["x"])
print("STEPLOOP ok mode=%%s rank=%%d/%%d finite=%%s graph=%%s" %% (mode, rank, world, fin, res["rccl-graph"]["g"]))
assert fin
'''


def _run(script_text, tmp_path, args=(), timeout=600, env=None):
    script = tmp_path / "worker.py"
    script.write_text(script_text % {"root": ROOT})
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ADMM_HIP_GRAPH_COMM"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, str(script)] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.gpu
def test_rccl_one_rank_communicator_lifecycle(tmp_path):
    """admm_hip_rccl_unique_id -> admm_hip_rccl_init(uid, 0, 1) -> admm_hip_debug_allreduce (values unchanged, bit for bit) ->
    replace -> remove (the all-reduce then refuses) -> install again -> destroy with the context; six contexts in a row, the last
    four leave no device memory behind; ncclCommGetAsyncError reads 0 throughout."""
    out = _run(LIFECYCLE, tmp_path)
    assert "LIFECYCLE ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("mode,rank,world", [("subtree", 0, 8), ("subtree", 5, 8), ("contiguous", 3, 8), ("subtree", 1, 2)])
def test_rccl_step_loop_one_rank_of_a_partition(tmp_path, mode, rank, world):
    """The real step loop over a real ncclAllReduce: rank `rank`'s share of a `world`-rank partition with a 1-rank
    communicator in the context (identity sums).  Frames are bitwise those of the no-op hook, with eager launches
    (ADMM_HIP_GRAPH_COMM=0) and with the iteration -- ncclAllReduce included -- replayed as a HIP graph (=1)."""
    out = _run(STEPLOOP, tmp_path, args=(mode, rank, world, 6, 6, 60))
    assert "STEPLOOP ok" in out and "finite=True" in out


def _bench_fake(tmp_path, shard, graph_comm, dims, port, extra_env=None):
    env = dict(os.environ, ADMM_BENCH_FAKE_WORLD="8", ADMM_BENCH_FAKE_DIST="1", BENCH_TIMING_EXPERIMENT="1", ADMM_HIP_GRAPH_COMM=str(graph_comm),
               MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(extra_env or {})
    from test_sharding import _free_port
    for attempt in range(3):      # (a port that was free a moment ago can be taken by the time the launcher binds it -- other tests' ranks come and go: try another one)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", "1", "--shard", shard, "--no-cpu-baseline", "--no-extras", "--steps", "3", "--warmup", "1", "--dims"] + [str(d) for d in dims]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
        port = _free_port()
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["subtree", "contiguous"])
def test_bench_fake_world_over_the_library_communicator(tmp_path, shard):
    """bench.py as a fresh torch.distributed.run child (started before anything touches the GPU): 1-rank NCCL process group,
    the ncclUniqueId through it, admm_hip_rccl_init, then rank 0's share of an 8-rank partition stepping over ncclAllReduce.
    Eager and captured iterations give the same frames (x_checksum is a sum of |x|: equal only if the frames are)."""
    from test_sharding import _free_port
    dims = (12, 12, 60)          # 10 309 nodes: the sweeps, sharded by subtree; below the graph threshold
    lines = {}
    for g in (0, 1):
        o = _bench_fake(tmp_path, shard, g, dims, _free_port())
        assert o["config"]["allreduce"] == "ncclAllReduce inside libadmm_hip.so", o["config"]["allreduce"]
        assert o["rccl_ranks_seen"] == 1 and o["fake_world"] == 8
        assert np.isfinite(o["config"]["x_checksum"]) and o["value"] > 0
        assert o["rccl_async_error"] == 0
        lines[g] = o
    assert lines[0]["config"]["x_checksum"] == lines[1]["config"]["x_checksum"]
    assert lines[1]["graph_state"]["iter_graph"] and not lines[0]["graph_state"]["iter_graph"]


@pytest.mark.gpu
def test_bench_fake_world_full_size_eager(tmp_path):
    """The driver's 8-GPU command at the headline size, one rank of it: 1 001 472 tets, subtree shards, eager launches with
    ncclAllReduce between the kernels (no graph above 100k nodes) -- what every rank of the first real 8-GPU run executes."""
    from test_sharding import _free_port
    o = _bench_fake(tmp_path, "subtree", 0, (32, 32, 163), _free_port())
    assert o["config"]["allreduce"] == "ncclAllReduce inside libadmm_hip.so" and o["rccl_ranks_seen"] == 1
    assert np.isfinite(o["config"]["x_checksum"]) and o["rccl_async_error"] == 0
    assert not o["graph_state"]["iter_graph"]
