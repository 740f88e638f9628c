#!/usr/bin/env python3
"""GPU box: sanity workloads beyond the benchmark -- the 4M-tet bar (64x64x163 cubes: 5 GB of panels), on request the 16M-tet bar
(80x80x417 cubes, `bar16m`: 64-bit offsets everywhere) and an unstructured Delaunay mesh of ~1M NH tets (150,000 random points in a
1x1x4 box) -- ms per ADMM iteration, phases, levels, and A solve(b) = b on the bars."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()


def run(s, n_el, label):
    t0 = time.time(); s.initialize(); ti = time.time() - t0
    s.step(20); s.sync()
    s.enable_timing(4)
    ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0)
    t = time.perf_counter()
    for _ in range(3):
        s.step(20)
        tm = s.timing()
        for k in ph:
            ph[k] += tm[k] / 60
    s.sync()
    t = (time.perf_counter() - t) / 60
    assert np.isfinite(s.m_x).all()
    inf = s.info()
    print(label, "elements %d nodes %d nnzL %.3g levels %d initialize %.1fs  ms/iter %.3f  iters/s x el %.4g " % (n_el, inf["n_nodes"], inf["nnz_L"], inf["n_levels"], ti, 1e3 * t, n_el / t),
          {k: round(v, 3) for k, v in ph.items()}, flush=True)


which = sys.argv[1:] or ["delaunay", "bar4m"]
if "delaunay" in which:
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 1, size=(150000, 3)) * np.array([1.0, 1.0, 4.0])
    tets = Delaunay(pts).simplices.astype(np.int32)
    e = pts[tets]
    vol = np.einsum("ij,ij->i", e[:, 1] - e[:, 0], np.cross(e[:, 2] - e[:, 0], e[:, 3] - e[:, 0])) / 6.0
    tets[vol < 0] = tets[vol < 0][:, [0, 1, 3, 2]]
    tets = tets[np.abs(vol) > 1e-4 * np.abs(vol).mean()]
    if os.environ.get("DELAUNAY_SORT"):      # experiment: the elements in Morton order of their centroids instead of scipy's order
        c = pts[tets].mean(axis=1); q = ((c - c.min(0)) / (c.max(0) - c.min(0)) * 1023).astype(np.uint64)
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
            return v
        key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        tets = tets[np.argsort(key, kind="stable")]
    m = pkg.meshgen.lumped_tet_mass(pts, tets, 1000.0)
    s = pkg.System(device_id=0); s.set_timestep(0.04)
    s.add_nodes(pts.ravel(), np.repeat(m, 3))
    s.add_forces(pkg.KIND["TET_NH"], tets, [1e5, 1e5, 5])
    s.add_forces(pkg.KIND["ANCHOR"], np.nonzero(pts[:, 2] < 0.05)[0].astype(np.int32), [-1.0, 1.0])
    s.add_gravity((0.0, -9.8, 0.0))
    run(s, tets.shape[0], "delaunay 150k points:")
    del s
def solve_check(s, label):
    """|A solve(b) - b| / |b| on this system's factor (the panels' offsets pass 2^31 entries at 16M tets)"""
    n = s.info()["n_nodes"]
    b = np.random.default_rng(1).normal(size=(n, 3))
    x = s.solve_only(b)
    r = s.apply_A(x) - b.ravel()
    print(label, "A solve(b) = b to %.2e (relative, 2-norm)" % (np.linalg.norm(r) / np.linalg.norm(b)), flush=True)


if "bar4m" in which:
    s = pkg.make_bar_system(64, 64, 163)
    run(s, s.n_tets, "bar 64x64x163:")
    solve_check(s, "bar 64x64x163:")
    del s
if "bar16m" in which:      # 16.0M tets, 2.74M nodes: more than 2^31 factor entries
    s = pkg.make_bar_system(80, 80, 417)
    run(s, s.n_tets, "bar 80x80x417:")
    solve_check(s, "bar 80x80x417:")
