"""Every environment knob of the library keeps the results right (GPU).

README.md's knob table lists ~45 ADMM_HIP_* variables; tests/test_abi.py checks that the table is complete.  This file
is the other half: no knob selects a path that nothing runs.  Each case builds the same scene with ONE knob moved off
its default and compares with the default build:

  * knobs that only change WHEN or WHERE something is computed (launch order, fused launches, storing z, frame-boundary
    transport, slot layout of the separate residual passes) must reproduce the default run BIT FOR BIT;
  * knobs that change the elimination tree or the sweep kernels' shapes change the order of fp64 sums: the solve agrees to
    1e-10 x max|x| with the default AND with the library's own assembled A (||A x - b||), and a few frames of a scene without a
    truncated minimiser (corotational tets + anchors) agree to 1e-9.

The knobs that other tests already move (LEAF, DENSE_MAX, GRAPH, FRAME_GRAPH, LOCAL_MULTI, PRERED, TPB, STATE_DIRECT,
TREE_SEARCH, MERGE*, BWD_NW*, BWD_CW2*, TOP_BWD_ALL, GRAPH_COMM, SLOTS_NODE_SORTED, FACTOR, VERBOSE, RCCL_ID_FILE) are
not repeated.  Not covered anywhere: ADMM_HIP_RCCL_LIB (the path only matters in a process that has no librccl loaded,
and the Python plumbing always has PyTorch's).
"""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS = (12, 12, 30)          # 5239 nodes: leaves of 64, separators of 169 (block items, roots with an explicit inverse)


def _bar(pkg, kind, params, dims=DIMS, rank=0, world=1, mode=None):
    mg = pkg.meshgen
    x, t = mg.bar(*dims)
    m = mg.lumped_tet_mass(x, t, 1000.0)
    s = pkg.System(device_id=0); s.set_timestep(0.04)
    s.add_nodes(x.ravel(), np.repeat(m, 3))
    s.add_forces(pkg.KIND[kind], t, params)
    s.add_forces(pkg.KIND["ANCHOR"], mg.bar_anchor_nodes(dims[0], dims[1]), [-1.0, 1.0])
    s.add_gravity([0.0, -9.8, 0.0])
    if world > 1:
        s.set_shard(rank, world)
        if mode:
            s.set_shard_mode(mode)
    return s


def _run(s, frames=3, iters=8):
    s.initialize()
    b = np.random.default_rng(5).normal(size=3 * s.n_nodes)
    sol = s.solve_only(b)
    res = np.abs(s.apply_A(sol) - b).max() / np.abs(b).max()
    xs = []
    for _ in range(frames):
        s.step(iters); xs.append(s.m_x.copy())
    return dict(sol=sol, res=res, xs=xs, v=s.m_v.copy(), u=s.read_local(0)["u"].copy(), info=s.info())


SWEEP_KNOBS = [
    ("ADMM_HIP_FWD_SMALL_K", "16"), ("ADMM_HIP_BWD_SMALL_K", "16"), ("ADMM_HIP_FWD_SMALL_K", "0"),
    ("ADMM_HIP_FWD_NW4", "0"), ("ADMM_HIP_FWD_NW4", "100000"), ("ADMM_HIP_FWD_NW8", "100000"), ("ADMM_HIP_FWD_NW16_TILES", "0"), ("ADMM_HIP_FWD_NW16_TILES", "1000000"),
    ("ADMM_HIP_XCD", "0"), ("ADMM_HIP_XCD", "1"),
    ("ADMM_HIP_ROOT_FUSE_K", "0"), ("ADMM_HIP_ROOT_FUSE_K", "64"), ("ADMM_HIP_ROOT_INVERSE", "0"), ("ADMM_HIP_ROOT_DEPTH", "3"),
    ("ADMM_HIP_THREADS", "1"), ("ADMM_HIP_THREADS", "3"),
]


@pytest.mark.parametrize("knob,value", SWEEP_KNOBS)
def test_tree_and_sweep_knobs_keep_the_solve(pkg, monkeypatch, knob, value):
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "64")
    ref = _run(_bar(pkg, "TET_LINEAR", [4000.0]))
    monkeypatch.setenv(knob, value)
    if knob == "ADMM_HIP_THREADS":
        monkeypatch.setenv("ADMM_HIP_FACTOR", "host")          # the host factorization is what the thread count belongs to
    got = _run(_bar(pkg, "TET_LINEAR", [4000.0]))
    assert got["res"] < 1e-11 and ref["res"] < 1e-11
    assert np.abs(got["sol"] - ref["sol"]).max() < 1e-10 * np.abs(ref["sol"]).max(), (knob, value)
    for f in range(3):
        assert np.isfinite(got["xs"][f]).all()
        assert np.abs(got["xs"][f] - ref["xs"][f]).max() < 1e-9, (knob, value, f, np.abs(got["xs"][f] - ref["xs"][f]).max())
    if knob == "ADMM_HIP_THREADS":
        assert got["info"]["host_threads"] == int(value) and got["info"]["device_factor"] == 0
    if knob == "ADMM_HIP_ROOT_DEPTH":
        assert got["info"]["n_levels"] <= ref["info"]["n_levels"]      # the root spans more bisection levels: a shallower tree


BITWISE_KNOBS = [("ADMM_HIP_FUSE_ANCHORS", "0"), ("ADMM_HIP_TET_ORDER", "0"), ("ADMM_HIP_TET_ORDER_MIN", "1000000"), ("ADMM_HIP_KEEP_Z", "0"), ("ADMM_HIP_KEEP_Z", "1")]


@pytest.mark.parametrize("knob,value", BITWISE_KNOBS)
def test_launch_knobs_are_bitwise_neutral(pkg, monkeypatch, knob, value):
    """anchors in their own launch instead of the tet launch's tail; tet blocks in mesh order instead of costliest first (the
    default here runs WITH the cost order: its block threshold is lowered for this scene); z stored or not whatever the caller said"""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_TET_ORDER_MIN", "8")          # 25 920 tets = 405 blocks: the launch order by cost is active
    ref = _run(_bar(pkg, "TET_STVK", [1e5, 1e5, 5]), frames=4)
    monkeypatch.setenv(knob, value)
    got = _run(_bar(pkg, "TET_STVK", [1e5, 1e5, 5]), frames=4)
    for f in range(4):
        assert np.array_equal(got["xs"][f], ref["xs"][f]), (knob, value, f)
    assert np.array_equal(got["v"], ref["v"]) and np.array_equal(got["u"], ref["u"]) and np.array_equal(got["sol"], ref["sol"])


def test_state_zerocopy_off_is_bitwise_the_same(pkg, monkeypatch):
    """the class API's frame boundary through one DMA per vector + reordering kernels instead of one zero-copy kernel each way"""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_STATE_DIRECT", "0")
    outs = []
    for zc in ("1", "0"):
        monkeypatch.setenv("ADMM_HIP_STATE_ZEROCOPY", zc)
        s = _bar(pkg, "TET_STVK", [1e5, 1e5, 5], dims=(6, 5, 14)); s.initialize()
        hx = s.m_x.copy(); hv = s.m_v.copy()
        s.pin_host(hx); s.pin_host(hv)
        try:
            for _ in range(3):
                s.upload_state(hx, hv); s.step(6); s.download_state(hx, hv)
        finally:
            s.pin_host(hx, False); s.pin_host(hv, False)
        outs.append((hx.copy(), hv.copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_unfused_residual_passes_agree_with_the_fused_ones(pkg, monkeypatch):
    """ADMM_HIP_RES_UNFUSED=1: |r|, |s| per iteration from the separate snapshot / primal / dual passes (per-corner slots) against the
    ones the tet and anchor kernels produce themselves: same definitions, other summation order -> 1e-10 relative; same trajectory bits."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_PRERED", "0")               # (the separate passes need per-corner slots: both runs use that layout)
    res = []
    for unf in (None, "1"):
        if unf:
            monkeypatch.setenv("ADMM_HIP_RES_UNFUSED", unf)
        s = _bar(pkg, "TET_STVK", [1e5, 1e5, 5], dims=(5, 4, 11)); s.initialize(); s.enable_residuals(True)
        s.step(10)
        r, sd, n = s.residuals()
        assert n == 10
        res.append((r.copy(), sd.copy(), s.m_x.copy()))
    assert np.array_equal(res[0][2], res[1][2])              # tracking never changes the iterates, whichever passes compute the norms
    assert np.allclose(res[0][0], res[1][0], rtol=1e-9, atol=1e-13) and np.allclose(res[0][1], res[1][1], rtol=1e-9, atol=1e-13)


def _hooks(world):
    import torch
    bar = threading.Barrier(world); bufs = {}

    class _Ptr:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def make(r):
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
    return [make(r) for r in range(world)]


@pytest.mark.parametrize("knob,value", [("ADMM_HIP_SHARD", "contiguous"), ("ADMM_HIP_SHARD", "subtree"), ("ADMM_HIP_SUBTREES_PER_RANK", "2"), ("ADMM_HIP_SUBTREES_PER_RANK", "4")])
def test_sharding_knobs(pkg, monkeypatch, knob, value):
    """ADMM_HIP_SHARD overrides the mode the caller set (read in admm_hip_finalize); ADMM_HIP_SUBTREES_PER_RANK deals every rank
    several smaller subtrees (larger replicated top).  Two ranks on one GPU: all ranks bitwise equal, the single-rank run to rounding."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "16")
    dims = (6, 6, 40)
    ref = _bar(pkg, "TET_LINEAR", [4000.0], dims=dims); ref.initialize()
    refx = []
    for _ in range(2):
        ref.step(8); refx.append(ref.m_x.copy())
    world = 2
    monkeypatch.setenv(knob, value)
    if knob == "ADMM_HIP_SUBTREES_PER_RANK":
        monkeypatch.setenv("ADMM_HIP_DIST_TOP", "0")      # (a knob of the replicated top's partition; the distributed top deals exactly one subtree per rank)
    caller_mode = "subtree" if value == "contiguous" else ("contiguous" if knob == "ADMM_HIP_SHARD" else "subtree")      # the environment must win over the caller
    shards = [_bar(pkg, "TET_LINEAR", [4000.0], dims=dims, rank=r, world=world, mode=caller_mode) for r in range(world)]
    hooks = _hooks(world)
    for r, s in enumerate(shards):
        s.set_allreduce(hooks[r])
    pkg.initialize_together(shards)
    out = [None] * world

    def run(r):
        xs = []
        for _ in range(2):
            shards[r].step(8); xs.append(shards[r].m_x.copy())
        out[r] = xs
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert all(o is not None for o in out)
    for f in range(2):
        assert np.array_equal(out[0][f], out[1][f])
        assert np.abs(out[0][f] - refx[f]).max() < 1e-9
    inf = shards[0].info()
    if knob == "ADMM_HIP_SHARD":
        assert (inf["nodes_top"] > 0) == (value == "subtree") and (inf["comm_doubles_frame"] > 0) == (value == "subtree")
        if value == "contiguous":
            a, e = shards[0].local_range(0); a1, e1 = shards[1].local_range(0)
            assert a == 0 and e == a1 and e1 == 6 * 6 * 40 * 6
    else:
        monkeypatch.delenv(knob)
        base = _bar(pkg, "TET_LINEAR", [4000.0], dims=dims, rank=0, world=world, mode="subtree"); base.initialize()
        assert inf["nodes_top"] >= base.info()["nodes_top"]      # more, smaller subtrees: a larger (or equal) replicated top
