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


@pytest.mark.parametrize("world", [2, 3, 8])
def test_subtree_partition_properties(pkg, monkeypatch, world):
    """Subtree sharding, host side: the ranks' element sets partition every batch; every node an element touches is
    owned by that element's rank or lies in the replicated top; the top is a small part of the nodes; the loads balance."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")          # (small systems are solved with the explicit inverse and never sharded)
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")              # a deep elimination tree on a small mesh
    dims = (6, 6, 40)
    mg = pkg.meshgen
    x, t = mg.bar(*dims)
    systems = []
    for r in range(world):
        s = pkg.make_bar_system(*dims, device_id=-1, rank=r, world=world)
        s.set_shard_mode("subtree")
        s.initialize()
        systems.append(s)
    owner = systems[0].node_owner()
    for s in systems[1:]:
        assert np.array_equal(s.node_owner(), owner)                     # every rank computes the same partition
    assert owner.min() == -1 and owner.max() == world - 1
    assert (owner == -1).mean() < 0.35
    anchors = mg.bar_anchor_nodes(dims[0], dims[1])
    for b, idx in enumerate((t, anchors.reshape(-1, 1))):
        seen = np.zeros(idx.shape[0], np.int32)
        for r, s in enumerate(systems):
            ids = s.local_elements(b)
            assert np.all(np.diff(ids) > 0)                              # reference order inside a rank
            seen[ids] += 1
            o = owner[idx[ids]]
            assert np.all((o == r) | (o == -1)), (b, r)
        assert np.all(seen == 1)
    loads = np.array([s.info()["n_elems_local"] for s in systems], dtype=float)
    assert loads.sum() == t.shape[0] + anchors.size and loads.max() < 1.6 * loads.mean()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_distributed_top_partition(pkg, monkeypatch, world):
    """The distributed top's plan, host side (a host-only context planning as a context with a device and a transport would:
    ADMM_HIP_PLAN_AS_IF_DEVICE): ONE top supernode = the separators of the first log2(world) bisection levels, one subtree per rank; the ranks'
    row slices of the root's explicit inverse tile it exactly; every rank keeps its own subtrees' panels + its slice, nobody the whole factor;
    the elements partition; every rank computes the same plan."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")
    monkeypatch.setenv("ADMM_HIP_PLAN_AS_IF_DEVICE", "1")
    monkeypatch.setenv("ADMM_HIP_DIST_TOP", "1")           # (by default only from 300k nodes on)
    dims = (6, 6, 48)
    systems = []
    for r in range(world):
        s = pkg.make_bar_system(*dims, device_id=-1, rank=r, world=world, shard_mode="subtree")
        s.initialize()
        systems.append(s)
    infos = [s.info() for s in systems]
    owner = systems[0].node_owner()
    assert all(np.array_equal(s.node_owner(), owner) for s in systems[1:])
    assert all(i["dist_top"] == 1 and i["factor_local"] == 1 for i in infos)
    k = infos[0]["nodes_top"]
    assert k == int((owner == -1).sum()) and 0 < k < 0.4 * owner.size and sorted(set(owner.tolist())) == [-1] + list(range(world))
    assert sum(i["sweep_entries_top"] for i in infos) == k * k and all(i["sweep_entries_top_bwd"] == 0 for i in infos)
    assert sum(i["nodes_own"] for i in infos) + k == owner.size
    whole = infos[0]["panel_bytes"] // 8
    assert all(i["factor_doubles_resident"] < whole for i in infos)
    assert sum(i["n_elems_local"] for i in infos) == infos[0]["n_elems_total"]
    assert all(i["comm_doubles_iter"] == infos[0]["comm_doubles_iter"] for i in infos) and infos[0]["comm_doubles_iter"] >= 6 * k
    # the default rule: distributed from ADMM_HIP_DIST_TOP_MIN_NODES nodes on (300k; this bar has 2.4k), rank-local factorization either way
    monkeypatch.delenv("ADMM_HIP_DIST_TOP")
    s = pkg.make_bar_system(*dims, device_id=-1, rank=0, world=world, shard_mode="subtree"); s.initialize()
    assert s.info()["dist_top"] == 0 and s.info()["factor_local"] == 1
    monkeypatch.setenv("ADMM_HIP_DIST_TOP_MIN_NODES", "2000")
    s = pkg.make_bar_system(*dims, device_id=-1, rank=0, world=world, shard_mode="subtree"); s.initialize()
    assert s.info()["dist_top"] == 1
    monkeypatch.delenv("ADMM_HIP_DIST_TOP_MIN_NODES")
    # without the knob a host-only context (no device, no transport) plans the replicated top of rounds 2-5
    monkeypatch.delenv("ADMM_HIP_PLAN_AS_IF_DEVICE")
    s = pkg.make_bar_system(*dims, device_id=-1, rank=0, world=world, shard_mode="subtree"); s.initialize()
    assert s.info()["dist_top"] == 0 and s.info()["factor_local"] == 0


@pytest.mark.parametrize("world", [2, 5])
def test_subtree_partition_unstructured(pkg, monkeypatch, world):
    """The same properties on an unstructured mesh (Delaunay tets of random points: 5-60 tets per node)."""
    from scipy.spatial import Delaunay
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 1, size=(3000, 3)) * np.array([1.0, 1.0, 3.0])
    t = Delaunay(pts).simplices.astype(np.int32)
    vol = np.abs(np.einsum("ij,ij->i", pts[t[:, 1]] - pts[t[:, 0]], np.cross(pts[t[:, 2]] - pts[t[:, 0]], pts[t[:, 3]] - pts[t[:, 0]]))) / 6.0
    t = t[vol > 1e-3 * vol.mean()]
    systems = []
    for r in range(world):
        s = pkg.System(device_id=-1); s.set_timestep(0.02)
        s.add_nodes(pts.ravel(), np.full(3 * pts.shape[0], 1e-3))
        s.add_forces(pkg.KIND["TET_LINEAR"], t, [50.0])
        s.set_shard(r, world); s.set_shard_mode("subtree")
        s.initialize()
        systems.append(s)
    owner = systems[0].node_owner()
    assert all(np.array_equal(s.node_owner(), owner) for s in systems[1:])
    assert owner.min() == -1 and owner.max() == world - 1 and (owner == -1).mean() < 0.5
    seen = np.zeros(t.shape[0], np.int32)
    for r, s in enumerate(systems):
        ids = s.local_elements(0)
        seen[ids] += 1
        o = owner[t[ids]]
        assert np.all((o == r) | (o == -1))
    assert np.all(seen == 1)
    loads = np.array([s.info()["n_elems_local"] for s in systems], dtype=float)
    assert loads.max() < 1.6 * loads.mean()


def _thread_allreduce_hooks(world):
    """all-reduce between `world` contexts living in ONE process on ONE GPU (threads meeting at barriers)"""
    import torch
    bar = threading.Barrier(world)
    bufs = {}

    class _Ptr:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def make_hook(r):
        def hook(ptr, count, stream):
            torch.cuda.synchronize()
            bufs[r] = torch.as_tensor(_Ptr(ptr, count), device="cuda:0")
            bar.wait()
            if r == 0:
                tot = bufs[0].clone()
                for q in range(1, world):
                    tot += bufs[q]
                for q in range(world):
                    bufs[q].copy_(tot)
                torch.cuda.synchronize()
            bar.wait()
            return 0
        return hook
    return [make_hook(r) for r in range(world)]


def _run_sharded(shards, frames, iters, b):
    world = len(shards)
    out = [None] * world
    errs = []

    def run(r):
        try:
            sol = shards[r].solve_only(b)
            xs = []
            for _ in range(frames):
                shards[r].step(iters)
                xs.append(shards[r].m_x.copy())
            out[r] = (sol, xs, shards[r].m_v.copy())
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))
            raise
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "subtree"), (4, "subtree"), (3, "subtree"), (8, "subtree"), (4, "contiguous")])
def test_sharded_step_matches_single_rank(pkg, monkeypatch, world, mode):
    """`world` shards of one scene on ONE GPU (one context and one host thread per rank, meeting in the all-reduce hook)
    against the unsharded step, plus the solve-only entry point.  Subtree sharding exchanges only the top separators' rows
    per iteration and rebuilds x once per frame.  Cloth (triangle strain + bend + anchors: no truncated minimiser): tight.
    StVK bar: the sums meet in another order and the truncated L-BFGS amplifies that: 1e-5 over 3 frames."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")          # the panel sweeps, not the small-system inverse
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")              # a deep elimination tree on a small mesh
    from conftest import golden
    g = golden("traj_cloth.npz")
    n = g["x"].shape[0]

    def cloth(rank, w):
        s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
        s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
        s.add_forces(pkg.KIND["TRI_STRAIN"], g["tris"], [float(g["k_tri"]), g["lim"][0], g["lim"][1], 1.0])
        s.add_forces(pkg.KIND["BEND"], g["hinges"], [float(g["k_bend"])])
        s.add_forces(pkg.KIND["ANCHOR"], g["anchors"], [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        if w > 1:
            s.set_shard(rank, w)
        return s
    dims = (5, 4, 30)
    def mixed(rank, w):     # two disconnected bodies (two elimination-tree roots), five force kinds
        return pkg.make_mixed_system(4, 3, 12, 10, 8, rank=rank, world=w)[0]
    from scipy.spatial import Delaunay                      # an unstructured mesh: corotational tets (no truncated minimiser -> tight)
    prng = np.random.default_rng(7)
    pts = prng.uniform(0, 1, size=(900, 3)) * np.array([1.0, 1.0, 3.0])
    dt_tets = Delaunay(pts).simplices.astype(np.int32)
    vol = np.abs(np.einsum("ij,ij->i", pts[dt_tets[:, 1]] - pts[dt_tets[:, 0]], np.cross(pts[dt_tets[:, 2]] - pts[dt_tets[:, 0]], pts[dt_tets[:, 3]] - pts[dt_tets[:, 0]]))) / 6.0
    dt_tets = dt_tets[vol > 1e-3 * vol.mean()]

    def delaunay(rank, w):
        s = pkg.System(device_id=0); s.set_timestep(0.02)
        s.add_nodes(pts.ravel(), np.full(3 * pts.shape[0], 1.0 / pts.shape[0]))
        s.add_forces(pkg.KIND["TET_LINEAR"], dt_tets, [50.0])
        s.add_forces(pkg.KIND["ANCHOR"], np.nonzero(pts[:, 2] < 0.2)[0].astype(np.int32), [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        if w > 1:
            s.set_shard(rank, w)
        return s
    for name, make, tol in (("cloth", cloth, 1e-9), ("bar", lambda r, w: pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"], rank=r, world=w), 1e-5),
                            ("mixed", mixed, 1e-5), ("delaunay", delaunay, 1e-9)):
        ref = make(0, 1)
        ref.initialize()
        shards = [make(r, world) for r in range(world)]
        hooks = _thread_allreduce_hooks(world)
        for r, s in enumerate(shards):
            s.set_shard_mode(mode)
            s.set_allreduce(hooks[r])
        pkg.initialize_together(shards)          # (collective under rank-local factorization: one thread per rank)
        assert sum(s.info()["n_elems_local"] for s in shards) == ref.info()["n_elems_total"]
        b = np.random.default_rng(2).normal(size=3 * ref.n_nodes)
        xref = ref.solve_only(b)
        out = _run_sharded(shards, 3, 10, b)
        refx = []
        for _ in range(3):
            ref.step(10); refx.append(ref.m_x.copy())
        for r in range(world):
            sol, xs, vs = out[r]
            assert np.abs(sol - xref).max() < 1e-10 * np.abs(xref).max(), (name, r, "solve")
            for f in range(3):
                assert np.abs(xs[f] - refx[f]).max() < tol, (name, r, f, np.abs(xs[f] - refx[f]).max())
            assert np.array_equal(xs[-1], out[0][1][-1]) and np.array_equal(vs, out[0][2])      # all ranks end bitwise identical


@pytest.mark.gpu
def test_backward_over_the_top_only_where_a_rank_reads_it(pkg, monkeypatch):
    """Under subtree sharding a rank runs the backward sweep of the replicated top only for the separators it reads (front rows of its
    subtrees, nodes of its elements, their ancestors); the full x of a frame takes every top node from the lowest rank that computed it.
    Eight shards of a deep tree: the frames are bitwise those of the runs that sweep the whole top on every rank
    (ADMM_HIP_TOP_BWD_ALL=1), and so is the solve-only entry point."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")
    monkeypatch.setenv("ADMM_HIP_DIST_TOP", "0")           # the REPLICATED top (what 3 / 5 / 6 ... ranks always run; 8 ranks by default split ONE root's product by rows instead)
    # long thin bars: the ranks' subtrees are slabs, a rank reads 2 of the 5 (world 8) resp. 2 - 5 of the 8 (world 6) top separators
    # (tools/probe/top_needed_verbose.py prints the counts)
    for world, dims in ((8, (4, 4, 120)), (6, (5, 5, 90))):
        ref = pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"]); ref.initialize()
        b = np.random.default_rng(3).normal(size=3 * ref.n_nodes)
        xref = ref.solve_only(b)
        res = {}
        for knob in ("0", "1"):
            monkeypatch.setenv("ADMM_HIP_TOP_BWD_ALL", knob)
            shards = [pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"], rank=r, world=world) for r in range(world)]
            hooks = _thread_allreduce_hooks(world)
            for r, s in enumerate(shards):
                s.set_shard_mode("subtree"); s.set_allreduce(hooks[r])
            pkg.initialize_together(shards)
            res[knob] = _run_sharded(shards, 2, 10, b)
        for r in range(world):
            assert np.abs(res["0"][r][0] - xref).max() < 1e-10 * np.abs(xref).max(), (world, r, "solve vs one rank")
            assert np.array_equal(res["0"][r][0], res["1"][r][0]), (world, r, "solve")
            for f in range(2):
                assert np.array_equal(res["0"][r][1][f], res["1"][r][1][f]), (world, r, f)
            assert np.array_equal(res["0"][r][2], res["1"][r][2])
            assert np.array_equal(res["0"][r][1][-1], res["0"][0][1][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_distributed_top_vs_replicated_top(pkg, monkeypatch, world):
    """Subtree shards of 2 / 4 / 8 ranks, the two designs of the top of the elimination tree side by side: DISTRIBUTED (round 6, the default: the
    separators of the first log2(world) bisection levels are ONE root supernode; every rank keeps and streams only its rows of the root's explicit
    inverse, the slices of x meet in a second small all-reduce) and REPLICATED (ADMM_HIP_DIST_TOP=0: every rank sweeps a multi-level top).  Both
    solve the single-rank system to 1e-10, both end a frame bitwise equal on all ranks; the distributed top leaves a rank less factor than the
    replicated one and nothing to sweep backward above its own subtrees."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")
    dims = (6, 6, 64)
    ref = pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"]); ref.initialize()
    b = np.random.default_rng(9).normal(size=3 * ref.n_nodes)
    xref = ref.solve_only(b)
    infos = {}
    # "1" / "0": the two tops; "host": the distributed top with the numeric factorization on the HOST (ADMM_HIP_FACTOR=host: the whole factor there, every rank
    # uploads its own panels and its rows of the root's inverse); "whole": ADMM_HIP_FACTOR_LOCAL=0, the round-5 scheme (every rank factors and keeps everything)
    for knob in ("1", "0", "host", "whole"):
        monkeypatch.setenv("ADMM_HIP_DIST_TOP", "0" if knob in ("0", "whole") else "1")
        if knob == "host": monkeypatch.setenv("ADMM_HIP_FACTOR", "host")
        else: monkeypatch.delenv("ADMM_HIP_FACTOR", raising=False)
        if knob == "whole": monkeypatch.setenv("ADMM_HIP_FACTOR_LOCAL", "0")
        else: monkeypatch.delenv("ADMM_HIP_FACTOR_LOCAL", raising=False)
        shards = [pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"], rank=r, world=world, shard_mode="subtree") for r in range(world)]
        hooks = _thread_allreduce_hooks(world)
        for r, s in enumerate(shards):
            s.set_allreduce(hooks[r])
        pkg.initialize_together(shards)
        out = _run_sharded(shards, 2, 10, b)
        infos[knob] = [s.info() for s in shards]
        for r in range(world):
            assert np.abs(out[r][0] - xref).max() < 1e-10 * np.abs(xref).max(), (knob, r)
            assert np.array_equal(out[r][1][-1], out[0][1][-1]) and np.array_equal(out[r][2], out[0][2]), (knob, r)
        assert sum(i["n_elems_local"] for i in infos[knob]) == ref.info()["n_elems_total"]
    d, rp = infos["1"], infos["0"]
    assert all(i["dist_top"] == 1 and i["factor_local"] == 1 for i in d) and all(i["dist_top"] == 0 and i["factor_local"] == 1 for i in rp)
    assert all(i["dist_top"] == 1 and i["device_factor"] == 0 and i["factor_doubles_resident"] == j["factor_doubles_resident"] for i, j in zip(infos["host"], d))
    assert all(i["factor_local"] == 0 and i["dist_top"] == 0 and 8 * i["factor_doubles_resident"] == i["panel_bytes"] and i["factor_exchange_doubles"] == 0 for i in infos["whole"])
    assert all(8 * i["factor_doubles_resident"] < i["panel_bytes"] and i["factor_exchange_doubles"] > 0 and i["device_factor"] == 1 for i in d + rp)
    assert all(i["sweep_entries_top_bwd"] == 0 for i in d)
    k = d[0]["nodes_top"]
    assert sum(i["sweep_entries_top"] for i in d) == k * k                      # the ranks' row slices tile the root's inverse exactly
    # recompute_weights is collective too (every rank re-factors its share, the subtree roots' update matrices meet again): the anchors' weights
    # down to 0.5, every third tet's doubled -- the sharded solve of the NEW system equals the single-rank solve of the new system
    monkeypatch.setenv("ADMM_HIP_DIST_TOP", "1")
    shards = [pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"], rank=r, world=world, shard_mode="subtree") for r in range(world)]
    hooks = _thread_allreduce_hooks(world)
    for r, s in enumerate(shards):
        s.set_allreduce(hooks[r])
    pkg.initialize_together(shards)
    wt = ref.read_rest(0)["weight"].copy(); wt[::3] *= 2.0
    wa = np.full(ref.read_rest(1)["weight"].size, 0.5)
    sol = [None] * world; errs = []

    def rew(r, s):
        try:
            s.set_weights(0, wt); s.set_weights(1, wa); s.recompute_weights()
            sol[r] = s.solve_only(b)
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
    ref.set_weights(0, wt); ref.set_weights(1, wa); ref.recompute_weights()
    xnew = ref.solve_only(b)
    th = [threading.Thread(target=rew, args=(r, s)) for r, s in enumerate(shards)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert not errs and all(x is not None for x in sol), errs
    assert np.abs(xnew - xref).max() > 1e-6 * np.abs(xref).max()                 # (the weights did change the system)
    for r in range(world):
        assert np.abs(sol[r] - xnew).max() < 1e-10 * np.abs(xnew).max(), r
        assert np.array_equal(sol[r], sol[0])
    assert sum(i["nodes_own"] for i in d) + k == ref.n_nodes
    assert d[0]["comm_doubles_iter"] > 3 * k                                     # two collectives: [top RHS | contribution rows] and the top's x


@pytest.mark.gpu
@pytest.mark.parametrize("world,dist", [(4, "1"), (3, "0"), (2, "0")])
def test_sharded_residuals_and_early_exit(pkg, monkeypatch, world, dist):
    """Residual tracking under subtree sharding (SURVEY 8(f) rank 4 x 8(e)): |r|^2 is additive over the ranks' elements, s = D^T W^T W (z - z_prev) is a sum over
    all ranks' elements -- both go through the all-reduce -- so every rank reports the single-rank norms (cloth: no truncated minimiser, 1e-9 relative), and a
    tolerance ends the ADMM loop at the same iteration on every rank (distributed top at 4 ranks, replicated at 3 and 2)."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")
    monkeypatch.setenv("ADMM_HIP_DIST_TOP", dist)
    from conftest import golden
    g = golden("traj_cloth.npz")
    n = g["x"].shape[0]

    def cloth(rank, w):
        s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
        s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
        s.add_forces(pkg.KIND["TRI_STRAIN"], g["tris"], [float(g["k_tri"]), g["lim"][0], g["lim"][1], 1.0])
        s.add_forces(pkg.KIND["BEND"], g["hinges"], [float(g["k_bend"])])
        s.add_forces(pkg.KIND["ANCHOR"], g["anchors"], [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        if w > 1:
            s.set_shard(rank, w); s.set_shard_mode("subtree")
        return s
    iters = 16
    ref = cloth(0, 1); ref.initialize(); ref.enable_residuals(True)
    ref.step(iters)
    r1, s1, n1 = ref.residuals()
    assert n1 == iters and r1[-1] < r1[0]
    tol_r, tol_s = 1.2 * r1[9], 1.2 * s1[9]
    ref2 = cloth(0, 1); ref2.initialize(); ref2.set_tolerance(tol_r, tol_s, 1); ref2.step(iters)
    stop1 = ref2.residuals()[2]
    assert 1 < stop1 < iters
    for tolerances in (None, (tol_r, tol_s)):
        shards = [cloth(r, world) for r in range(world)]
        hooks = _thread_allreduce_hooks(world)
        for r, sh in enumerate(shards):
            sh.set_allreduce(hooks[r])
        pkg.initialize_together(shards)
        assert shards[0].info()["dist_top"] == int(dist)
        out = [None] * world; errs = []

        def run(r):
            try:
                if tolerances: shards[r].set_tolerance(tolerances[0], tolerances[1], 1)
                else: shards[r].enable_residuals(True)
                shards[r].step(iters)
                out[r] = shards[r].residuals() + (shards[r].m_x.copy(),)
            except Exception as e:  # noqa: BLE001
                errs.append((r, repr(e)))
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join(timeout=300) for t in th]
        assert not errs and all(o is not None for o in out), errs
        for r in range(world):
            rr, ss, nn, xx = out[r]
            if tolerances:
                assert nn == stop1, (r, nn, stop1)                                   # the same iteration ends the loop on every rank, and it is the single-rank one
            else:
                assert nn == iters and np.allclose(rr, r1, rtol=1e-9, atol=1e-14) and np.allclose(ss, s1, rtol=1e-9, atol=1e-14), r
            assert np.array_equal(xx, out[0][3])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_four_way_subtrees_match_single_rank(pkg, world):
    """Shares of >= 4096 nodes: regions up to 4/3 of a rank's share become four-way tree nodes inside the ranks' subtrees (host_factor); the
    sharded solve -- own subtrees, one exchange, replicated top -- equals the single-rank solve of the same system, and a frame of ADMM
    iterations ends bitwise identical on every rank and within the truncated minimiser's sensitivity of the single-rank frame."""
    dims = (16, 16, 60)      # 17 629 nodes
    ref = pkg.make_bar_system(*dims); ref.initialize()
    shards = [pkg.make_bar_system(*dims, rank=r, world=world) for r in range(world)]
    hooks = _thread_allreduce_hooks(world)
    for r, s in enumerate(shards):
        s.set_shard_mode("subtree"); s.set_allreduce(hooks[r])
    pkg.initialize_together(shards)
    assert shards[0].info()["n_levels"] < ref.info()["n_levels"] or shards[0].info()["n_levels"] <= 6      # merged subtrees: a shallow tree
    assert sum(s.info()["n_elems_local"] for s in shards) == ref.info()["n_elems_total"]
    b = np.random.default_rng(5).normal(size=3 * ref.n_nodes)
    xref = ref.solve_only(b)
    out = _run_sharded(shards, 1, 10, b)
    ref.step(10)
    for r in range(world):
        sol, xs, vs = out[r]
        assert np.abs(sol - xref).max() < 1e-10 * np.abs(xref).max(), (r, "solve")
        assert np.abs(xs[0] - ref.m_x).max() < 2e-5, (r, np.abs(xs[0] - ref.m_x).max())      # (the sums meet in another order; the truncated L-BFGS amplifies that: DESIGN section 4)
        assert np.array_equal(xs[0], out[0][1][0]) and np.array_equal(vs, out[0][2])


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(8, "subtree"), (8, "subtree-distributed-top"), (4, "contiguous")])
def test_full_size_shards_vs_compiled_reference(pkg, monkeypatch, world, mode):
    """The headline workload as the multi-GPU runs cut it -- the 1,001,472-tet Neo-Hookean bar in 8 subtree shards (4 contiguous shards) -- on ONE
    GPU (one context and host thread per rank, the hook sums the ranks' buffers): after one frame of 20 iterations every rank holds the same bits,
    and they agree with the COMPILED REFERENCE's frame (tests/golden/traj_bar_1M.npz) within the bound every trajectory fixture uses, 20 x the
    reference's own sensitivity to a 1-ulp perturbation of its start -- the sharded sums meet in another order than the single-GPU run's, the
    local steps are bit-identical.  The partition is exact and the exchange is the small one (subtree: top rows only)."""
    from conftest import golden
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "traj_bar_1M.npz")):
        pytest.skip("full-size fixture not generated")
    g = golden("traj_bar_1M.npz")
    dims = [int(v) for v in g["dims"]]
    if mode == "subtree-distributed-top":       # (178 596 nodes: replicated by default; the distributed top forced)
        monkeypatch.setenv("ADMM_HIP_DIST_TOP", "1"); mode = "subtree"
    shards = [pkg.make_bar_system(*dims, rank=r, world=world, shard_mode=mode) for r in range(world)]
    hooks = _thread_allreduce_hooks(world)
    for r, s in enumerate(shards):
        s.set_allreduce(hooks[r]); s.keep_z(False)
    pkg.initialize_together(shards)
    infos = [s.info() for s in shards]
    assert sum(i["n_elems_local"] for i in infos) == infos[0]["n_elems_total"] and infos[0]["n_nodes"] == int(g["n_nodes"])
    if mode == "subtree":
        assert sum(i["nodes_own"] for i in infos) + infos[0]["nodes_top"] == infos[0]["n_nodes"]
        assert all(i["factor_local"] == 1 and i["factor_doubles_resident"] < 0.6 * infos[0]["panel_bytes"] / 8 for i in infos)      # nobody holds the whole factor
        assert infos[0]["dist_top"] == (1 if os.environ.get("ADMM_HIP_DIST_TOP") == "1" else 0)
        assert 8 * infos[0]["comm_doubles_iter"] < 1 << 20                   # well under a megabyte per iteration (the whole RHS is 4.29 MB)
    out = [None] * world
    errs = []

    def run(r):
        try:
            shards[r].step(int(g["iters"]))
            out[r] = shards[r].m_x
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=600) for t in th]
    assert not errs and all(o is not None for o in out), errs
    for r in range(1, world):
        assert np.array_equal(out[r], out[0]), r
    x = out[0].reshape(-1, 3)
    bound = max(1e-9, 20.0 * float(g["ulp_sensitivity"]))
    err = np.abs(x[::int(g["stride"])] - g["x_sample"]).max()
    assert err < bound, (err, bound)
    assert abs(np.abs(x).sum() - float(g["sum_abs"])) < bound * x.size


@pytest.mark.gpu
def test_4M_tet_bar_in_8_shards_matches_one_gpu(pkg):
    """The configuration of the per-rank tables at the size where sharding pays (64x64x163 bar, 4 005 888 tets, 692 900 nodes, 8 ranks): above
    300k nodes the top is DISTRIBUTED by default and a rank's subtrees are rebuilt with near-leaf four-way nodes only (its levels stream > 100 MB).  Against the
    same bar on one GPU: the solve of a random right-hand side to 1e-10, one ADMM iteration from a deformed start to 1e-10 (local steps bit-identical, two
    different eliminations), all ranks bitwise equal, nobody holds more than a quarter of what the one-GPU run holds."""
    from checkers import deformed_start
    dims, world = (64, 64, 163), 8
    ref = pkg.make_bar_system(*dims); ref.initialize()
    b = np.random.default_rng(4).normal(size=3 * ref.n_nodes)
    xref = ref.solve_only(b)
    x0 = deformed_start(ref.m_x)
    ref.m_x = x0; ref.step(1)
    x1 = ref.m_x.copy()
    whole = ref.info()["panel_bytes"]
    del ref
    shards = [pkg.make_bar_system(*dims, rank=r, world=world, shard_mode="subtree") for r in range(world)]
    hooks = _thread_allreduce_hooks(world)
    for r, s in enumerate(shards):
        s.set_allreduce(hooks[r]); s.keep_z(False)
    pkg.initialize_together(shards)
    infos = [s.info() for s in shards]
    assert all(i["dist_top"] == 1 and i["factor_local"] == 1 for i in infos)
    assert max(8 * i["factor_doubles_resident"] for i in infos) < 0.3 * whole
    assert sum(i["n_elems_local"] for i in infos) == infos[0]["n_elems_total"]
    out = [None] * world; errs = []

    def run(r):
        try:
            sol = shards[r].solve_only(b)
            shards[r].m_x = x0; shards[r].step(1)
            out[r] = (sol, shards[r].m_x.copy())
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=900) for t in th]
    assert not errs and all(o is not None for o in out), errs
    for r in range(world):
        assert np.array_equal(out[r][0], out[0][0]) and np.array_equal(out[r][1], out[0][1]), r
    assert np.abs(out[0][0] - xref).max() < 1e-10 * np.abs(xref).max()
    assert np.abs(out[0][1] - x1).max() < 1e-10, np.abs(out[0][1] - x1).max()


@pytest.mark.gpu
def test_full_size_mixed_scene_in_8_shards_vs_compiled_reference(pkg):
    """BASELINE.json configs[4] as it is meant to run -- 498,888 NH + StVK tets, 99,856 cloth triangles, 149k hinges, anchors on "8 GPUs" -- as 8 subtree
    shards on ONE GPU: every force kernel of the scene under sharding (one launch per rank for its whole local step), two bodies = two elimination
    trees whose tops are replicated; all ranks bitwise equal after one 20-iteration frame and inside the compiled reference's envelope
    (tests/golden/traj_mixed_full.npz)."""
    from conftest import golden
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "traj_mixed_full.npz")):
        pytest.skip("full-size fixture not generated")
    g = golden("traj_mixed_full.npz")
    world = 8
    shards = [pkg.make_mixed_system(*[int(v) for v in g["bar_dims"]], *[int(v) for v in g["cloth"]], rank=r, world=world, shard_mode="subtree")[0] for r in range(world)]
    hooks = _thread_allreduce_hooks(world)
    for r, s in enumerate(shards):
        s.set_allreduce(hooks[r]); s.keep_z(False)
    pkg.initialize_together(shards)
    infos = [s.info() for s in shards]
    assert infos[0]["n_nodes"] == int(g["n_nodes"]) and sum(i["n_elems_local"] for i in infos) == infos[0]["n_elems_total"]
    out = [None] * world
    errs = []

    def run(r):
        try:
            shards[r].step(int(g["iters"]))
            out[r] = shards[r].m_x
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=600) for t in th]
    assert not errs and all(o is not None for o in out), errs
    for r in range(1, world):
        assert np.array_equal(out[r], out[0]), r
    x = out[0].reshape(-1, 3)
    bound = max(1e-9, 20.0 * float(g["ulp_sensitivity"]))
    err = np.abs(x[::int(g["stride"])] - g["x_sample"]).max()
    assert err < bound, (err, bound)
    assert abs(np.abs(x).sum() - float(g["sum_abs"])) < bound * x.size


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
    pkg.initialize_together(shards)
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
    args = ["--steps", "2", "--warmup", "1", "--dims", "8", "8", "40", "--no-cpu-baseline"]      # 3321 nodes: the panel sweeps, sharded by subtree
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    env = dict(os.environ, ADMM_BENCH_SHARE_GPU="1", ADMM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    # PLAIN `python bench.py --gpus 2`: bench.py starts its two ranks itself (a fresh torch.distributed.run child, before
    # anything touches the GPU) and relays rank 0's line and the return code
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    lines = [l for l in r2.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["value"] > 0 and "cpu_baseline" not in two
    assert two["rccl_ranks_seen"] == 2 and two["ranks_ok"] is True and "gloo" in two["config"]["allreduce"]
    # ... and as the driver launches it (torch.distributed.run around bench.py): same line
    for attempt in range(3):      # (a port that was free a moment ago can be taken by the time the launcher binds it: try another one)
        r3 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                             os.path.join(root, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True, timeout=900, env=env)
        if r3.returncode == 0 or "EADDRINUSE" not in r3.stderr:
            break
    assert r3.returncode == 0, r3.stderr[-3000:]
    three = json.loads([l for l in r3.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert three["n_gpus"] == 2 and three["config"]["x_checksum"] == two["config"]["x_checksum"]
    assert two["config"]["parallelism"].startswith("x2: elements and elimination subtrees")
    pr = two["per_rank"]                                       # per-rank phase times and element counts travel with the line
    assert len(pr["local_ms"]) == 2 and min(pr["total_ms"]) > 0 and sum(pr["elements"]) == 8 * 8 * 40 * 6 + 9 * 9
    # ---- the N > 1 line's schema: every rank's times, bytes and counts; the exchange; the sweeps' roofline ----
    for k in ("local_ms", "rhs_ms", "allreduce_ms", "solve_fwd_ms", "solve_bwd_ms", "total_ms", "elements", "tets", "fwd_bytes", "bwd_bytes", "nodes_own"):
        assert len(pr[k]) == 2, k
    assert sum(pr["tets"]) == 8 * 8 * 40 * 6
    for k, v in two["per_rank_summary"].items():
        assert v["slowest"] >= v["fastest"] >= 0 and v["slowest"] == max(pr[k])
    cm, sh, inf1 = two["comm"], two["shard"], one["config"]
    assert cm["collectives_per_iter"] == 1 and cm["collectives_per_frame_extra"] == 1 and cm["allreduce_ms_per_iter"] > 0           # (3 321 nodes: the replicated top, one collective; N = 4 / 8 below: distributed)
    assert cm["bytes_per_frame_extra"] == 8 * 3 * 9 * 9 * 41 and 0 < cm["bytes_per_iter"] < cm["bytes_per_frame_extra"]      # the top rows only, not the whole RHS
    assert cm["bytes_per_collective"] == cm["bytes_per_iter"] and cm["top"].startswith("replicated") and two["factor"]["rank_local"] is True
    assert sh["mode"] == "subtree" and sh["nodes_top"] > 0 and sh["nodes_own"] == pr["nodes_own"][0] and sum(pr["nodes_own"]) + sh["nodes_top"] == 9 * 9 * 41
    assert sh["sweep_entries_top_bwd"] <= sh["sweep_entries_top"] and 0 < sh["replicated_top_share_of_fwd_bytes"] < 1
    assert two["rccl_async_error"] == 0 and "graph_state" in two
    rf = two["roofline"]
    assert set(rf["all"]) >= {"solve_fwd (gather+panel kernels, all levels)", "solve_bwd_kernel (all levels)"}       # the sweeps keep their entries at N > 1
    assert rf["all"]["solve_fwd (gather+panel kernels, all levels)"]["GB/s"] > 0 and rf["iteration"]["bytes"] > 0 and rf["iteration"]["frac"] > 0
    # a rank streams less than the whole factor, and the two ranks together the whole factor + the top once more
    whole = one["roofline"]["all"]["solve_fwd (gather+panel kernels, all levels)"]
    assert max(pr["fwd_bytes"]) < whole["GB/s"] * whole["ms"] * 1e6 * 1.0001
    # the N = 1 line from the same code: same schema minus the rank objects, value = frames * iters / wall * tets
    assert "per_rank" not in one and "comm" not in one and one["n_gpus"] == 1
    assert abs(one["value"] - 2 * 20 / (one["ms_per_step"] * 2e-3) * 8 * 8 * 40 * 6) < 1e-6 * one["value"]
    # the partial sums meet in a different order; the NH bar amplifies that to ~1e-7 over three frames
    assert abs(two["config"]["x_checksum"] - one["config"]["x_checksum"]) < 1e-6 * one["config"]["x_checksum"]


_BENCH_NRANK_ARGS = ["--steps", "2", "--warmup", "1", "--dims", "12", "12", "60", "--no-cpu-baseline", "--no-extras"]      # 10 309 nodes: panel sweeps, a top of several levels at 8 ranks
_bench_one_rank_cache = {}


def _bench_one_rank_line():
    import json
    if "line" not in _bench_one_rank_cache:
        r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + _BENCH_NRANK_ARGS, capture_output=True, text=True, timeout=900)
        assert r1.returncode == 0, r1.stderr[-2000:]
        _bench_one_rank_cache["line"] = json.loads(r1.stdout.strip().splitlines()[-1])
    return _bench_one_rank_cache["line"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 8])
def test_bench_n_ranks_end_to_end(n):
    """`python bench.py --gpus 4` / `--gpus 8` exactly as typed -- what the driver's first multi-GPU commands are: bench.py starts its N ranks
    itself (fresh torch.distributed.run child), every rank a REAL process with its own context; on this one-GPU box they share cuda:0 and gloo
    stands in for RCCL (bench.py's test hooks).  N processes through the consensus bootstrap, the rank-local factorization (initialize is a
    collective: the subtree roots' update matrices meet in one all-reduce), the subtree partition with a replicated top of several levels,
    the N-way per_rank gather: rc 0, ONE JSON line, a head count of N through the very all-reduce path the iterations use, every per_rank
    list N long, the ranks' elements a partition, every rank factored only its share, the final positions those of the one-rank run."""
    import json
    one = _bench_one_rank_line()
    env = dict(os.environ, ADMM_BENCH_SHARE_GPU="1", ADMM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", ADMM_HIP_DIST_TOP="1")      # (the distributed top: by default only from 300k nodes on)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + _BENCH_NRANK_ARGS, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    L = json.loads(lines[0])
    assert L["n_gpus"] == n and L["steps"] == 2 and L["value"] > 0 and L["scaling"] == "strong"
    assert L["rccl_ranks_seen"] == n and L["ranks_ok"] is True and "gloo" in L["config"]["allreduce"]
    pr = L["per_rank"]
    for k in ("local_ms", "rhs_ms", "allreduce_ms", "solve_fwd_ms", "solve_bwd_ms", "total_ms", "elements", "tets", "fwd_bytes", "bwd_bytes", "nodes_own",
              "factor_mb_resident", "factor_numeric_s"):
        assert len(pr[k]) == n, k
    assert sum(pr["elements"]) == 12 * 12 * 60 * 6 + 13 * 13 and sum(pr["tets"]) == 12 * 12 * 60 * 6 and min(pr["elements"]) > 0
    sh = L["shard"]
    assert sh["mode"] == "subtree" and sum(pr["nodes_own"]) + sh["nodes_top"] == 13 * 13 * 61 and sh["nodes_top"] > 0
    fl = L["factor"]
    assert fl["rank_local"] is True and fl["exchange_bytes_once"] > 0
    assert max(pr["factor_mb_resident"]) < fl["whole_mb"] and sum(pr["factor_mb_resident"]) < fl["whole_mb"] * (1.0 + n * sh["replicated_top_share_of_all_entries"] + 0.3)
    assert L["rccl_async_error"] == 0 and L["comm"]["collectives_per_iter"] == 2 and L["comm"]["top"].startswith("distributed")
    assert abs(L["config"]["x_checksum"] - one["config"]["x_checksum"]) < 1e-6 * one["config"]["x_checksum"]


@pytest.mark.gpu
def test_bench_eight_ranks_a_hung_collective_exits_nonzero():
    """Eight real processes, rank 5 stops reaching the collective in a timed frame: every watchdog ends its rank within the per-frame limit, the
    launcher relays a non-zero code, no JSON line."""
    import time
    env = dict(os.environ, ADMM_BENCH_SHARE_GPU="1", ADMM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", ADMM_BENCH_FRAME_TIMEOUT_MIN="5", ADMM_BENCH_FRAME_TIMEOUT_MAX="10",
               ADMM_BENCH_TEST_HANG_RANK="5", ADMM_BENCH_TEST_HANG_AFTER="13")      # the 14th collective of the timed region: the 7th iteration of timed frame 0 (two per iteration under the distributed top)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + _BENCH_NRANK_ARGS, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "no progress for" in r.stderr and "timed frame" in r.stderr and "exiting with code 3" in r.stderr
    assert time.time() - t < 600


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    """`--gpus N` must describe the launch: a launcher that started another number of ranks is an error, not a silently
    mislabelled run (no torch import, no GPU needed for the check)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    env = dict(os.environ, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_bench_self_launch_relays_the_childs_failure():
    """`python bench.py --gpus 2` without a launcher starts its own ranks; in the GPU-less build container both ranks refuse
    to run ("needs an MI355X", no CPU fallback) and the parent must hand that failure on as a non-zero exit code."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("build-container check (on a GPU box the ranks would run the full-size bench)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "starting -m torch.distributed.run" in r.stderr and "needs an MI355X" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_watchdog_ends_a_stalled_process():
    """bench.Watchdog (N > 1 runs): a phase that outlasts its limit ends the process with code 3 and a message naming rank and phase --
    a plain exit from a daemon thread, no re-exec; a process that keeps beating, or that stopped the watchdog, is left alone."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "w = bench.Watchdog(3, 8, limit_s=30.0, poll_s=0.05)\n"
            "for i in range(5): w.beat('frame %%d' %% i, 20.0); time.sleep(0.05)\n"      # (generous while it beats: a loaded build container must not trip it)
            "mode = sys.argv[1]\n"
            "if mode == 'stop': w.stop()\n"
            "w.beat('timed frame 7: waiting for its events', 0.3) if mode != 'stop' else None\n"
            "time.sleep(3.0); print('survived')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code, "hang"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "survived" not in r.stdout
    assert "rank 3 of 8" in r.stderr and "timed frame 7: waiting for its events" in r.stderr and "exiting with code 3" in r.stderr
    r = subprocess.run([sys.executable, "-c", code, "stop"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "survived" in r.stdout


@pytest.mark.gpu
def test_bench_two_ranks_a_hung_collective_exits_nonzero():
    """One rank of a 2-rank run never reaches the collective of a timed frame (test hook in bench.py's all-reduce hook): the watchdogs
    end the ranks within the per-frame limit, bench.py relays a non-zero code, no JSON line, and stderr says which rank stalled where."""
    import time
    args = ["--steps", "2", "--warmup", "1", "--dims", "8", "8", "40", "--no-cpu-baseline"]
    env = dict(os.environ, ADMM_BENCH_SHARE_GPU="1", ADMM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", ADMM_BENCH_FRAME_TIMEOUT_MIN="5", ADMM_BENCH_FRAME_TIMEOUT_MAX="10",
               ADMM_BENCH_TEST_HANG_RANK="1", ADMM_BENCH_TEST_HANG_AFTER="13")      # the 14th collective of the timed region: the 7th iteration of timed frame 0 (two per iteration under the distributed top)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "no progress for" in r.stderr and "timed frame" in r.stderr and "exiting with code 3" in r.stderr
    assert time.time() - t < 300


def _spring_net(n=9, h=0.1):
    """The net of tests/cpp/user_force.cpp: n x n nodes, springs to the +x, +z and diagonal neighbours."""
    x = np.zeros((n * n, 3))
    for j in range(n):
        for i in range(n):
            x[i + n * j] = (h * i, 0.02 * np.sin(0.7 * i + 0.3 * j), h * j)
    pairs = []
    for j in range(n):
        for i in range(n):
            q = i + n * j
            for nb in (q + 1 if i + 1 < n else -1, q + n if j + 1 < n else -1, q + n + 1 if (i + 1 < n and j + 1 < n) else -1):
                if nb >= 0:
                    pairs.append((q, nb))
    return x, np.array(pairs, np.int32)


def _user_spring_system(pkg, x, pairs, k, rank=0, world=1, mode=None, generic=True):
    """Springs as USER forces through the generic batch (selector rows as triplets, project() = a numpy hook doing Spring's
    arithmetic, Force.cpp:52-71, on this rank's elements), or as the built-in kind."""
    s = pkg.System(device_id=0); s.set_timestep(0.04)
    s.add_nodes(x.ravel(), np.full(x.size, 0.05))
    ne = pairs.shape[0]
    w = np.sqrt(k)
    if generic:
        rows = np.repeat(np.arange(ne) * 3, 3) + np.tile(np.arange(3), ne)          # element e owns rows 3e..3e+2
        tr = np.concatenate([rows, rows]).astype(np.int32)
        tc = np.concatenate([3 * np.repeat(pairs[:, 0], 3) + np.tile(np.arange(3), ne), 3 * np.repeat(pairs[:, 1], 3) + np.tile(np.arange(3), ne)]).astype(np.int32)
        tv = np.concatenate([np.ones(3 * ne), -np.ones(3 * ne)])
        b = s.add_generic(np.arange(ne + 1) * 3, tr, tc, tv, np.full(3 * ne, w))
        d0 = x[pairs[:, 0]] - x[pairs[:, 1]]
        rest = np.sqrt(d0[:, 0] * d0[:, 0] + (d0[:, 1] * d0[:, 1] + d0[:, 2] * d0[:, 2]))
        state = {}

        def project(dt, Dx, u, z):
            ids = state.setdefault("ids", s.local_elements(b))        # this rank's user forces
            r = (3 * ids[:, None] + np.arange(3)[None, :])
            dx = Dx[r]; d = dx + u[r]
            nrm = np.sqrt(d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]))
            c = 1.0 / (w * w + k)
            with np.errstate(invalid="ignore", divide="ignore"):
                dn = np.where(nrm[:, None] <= 0.0, 0.0, d / nrm[:, None])
            zi = c * (k * (rest[ids][:, None] * dn) + (w * w) * d)
            u[r] += (dx - zi)
            z[r] = zi
        s.set_project_hook(project)
    else:
        s.add_forces(pkg.KIND["SPRING"], pairs, [k])
    n = int(round(np.sqrt(x.shape[0])))
    s.add_forces(pkg.KIND["ANCHOR"], np.array([0, n - 1], np.int32), [-1.0, 1.0])
    s.add_gravity([0, -9.8, 0])
    if world > 1:
        s.set_shard(rank, world)
        s.set_shard_mode(mode)
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "subtree"), (3, "subtree"), (2, "contiguous")])
def test_user_forces_sharded(pkg, monkeypatch, world, mode):
    """User-defined forces (generic batch + project hook) under sharding: every rank projects its own user forces, the device
    evaluates D_i x for their rows and adds their shares of the right-hand side.  Unsharded the user springs equal the built-in
    spring kernel bit for bit; sharded they agree with the unsharded run to rounding and all ranks end bitwise identical."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "8")
    x, pairs = _spring_net(11)
    k = 400.0
    builtin = _user_spring_system(pkg, x, pairs, k, generic=False); builtin.initialize()
    ref = _user_spring_system(pkg, x, pairs, k); ref.initialize()
    refx = []
    for _ in range(3):
        builtin.step(10); ref.step(10)
        assert np.array_equal(builtin.m_x, ref.m_x)
        refx.append(ref.m_x.copy())
    shards = [_user_spring_system(pkg, x, pairs, k, rank=r, world=world, mode=mode) for r in range(world)]
    hooks = _thread_allreduce_hooks(world)
    for r, s in enumerate(shards):
        s.set_allreduce(hooks[r])
    pkg.initialize_together(shards)
    assert sorted(np.concatenate([s.local_elements(0) for s in shards]).tolist()) == list(range(pairs.shape[0]))     # every user force on exactly one rank
    out = _run_sharded(shards, 3, 10, np.zeros(3 * ref.n_nodes))
    for r in range(world):
        _, xs, vs = out[r]
        for f in range(3):
            assert np.abs(xs[f] - refx[f]).max() < 1e-9, (r, f)
        assert np.array_equal(xs[-1], out[0][1][-1]) and np.array_equal(vs, out[0][2])


@pytest.mark.gpu
def test_user_forces_residuals_match_builtin(pkg, monkeypatch):
    """Residual tracking with user-defined forces: |r| and |s| per ADMM iteration of the user-spring net equal those of the built-in
    spring kernel (the user rows' |r|^2 is summed on the host, their share of s goes through the same slots), and the early exit
    stops at the same iteration."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    x, pairs = _spring_net(11)
    k = 400.0
    res = []
    for generic in (False, True):
        s = _user_spring_system(pkg, x, pairs, k, generic=generic)
        s.initialize(); s.enable_residuals(True)
        s.step(12)
        r, sd, n = s.residuals()
        assert n == 12 and r.size == 12
        res.append((r.copy(), sd.copy(), s.m_x.copy()))
    assert np.array_equal(res[0][2], res[1][2])
    assert np.allclose(res[0][0], res[1][0], rtol=1e-10, atol=1e-14) and np.allclose(res[0][1], res[1][1], rtol=1e-10, atol=1e-14)
    assert res[0][0][-1] < res[0][0][0]                       # the primal residual falls over the frame
    tol_r, tol_s = 1.5 * res[0][0][6], 1.5 * res[0][1][6]
    stops = []
    for generic in (False, True):
        s = _user_spring_system(pkg, x, pairs, k, generic=generic)
        s.initialize(); s.set_tolerance(tol_r, tol_s, 1)
        s.step(12)
        stops.append(s.residuals()[2])
    assert stops[0] == stops[1] and stops[0] < 12
