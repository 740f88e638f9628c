#!/usr/bin/env python3
"""Golden vectors that pin the GLOBAL SOLVE directly (build container only: needs oracle/_ref/libadmm_ref.so, i.e. the
reference compiled from /root/reference by oracle/Makefile):

    x = solver.solve(b)      System.cpp:62, Eigen SimplicialLDLT, SimplicialCholesky.h:153-177

evaluated by the reference's OWN solver object (derived-class accessor in oracle/ref_shim.cpp, ref_solve) on three
right-hand sides (checkers.solve_rhs: white noise, an M x_bar-like vector, a unit spike) for
  * dillo   -- shipped armadillo, 2761 NH tets + anchors (scene arrays: traj_dillo_nh.npz), and once more after
               System::recompute_weights with the anchors' weights set to 0 (poordillo's H / F keys) and every third
               tet's weight doubled (System.cpp:159-179);
  * bunny   -- shipped bunny, 2510 StVK tets, no anchors (traj_bunny_stvk.npz);
  * cloth   -- triangle strain + bend + anchors (traj_cloth.npz);
  * bar100k -- the 16x16x65-cube NH bar (99 840 tets, 19 074 nodes; the solution is stored at every 8th dof + norms).
-> tests/golden/solve_<scene>.npz.  usage: python tests/golden/make_golden_solve.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from checkers import KIND, Ref, have_ref, solve_rhs  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
assert have_ref(), "the compiled reference (oracle/_ref/libadmm_ref.so) is needed"
pkg = load_package()


def scene(name):
    """-> (dt, x [n][3], m3 [3n], forces = [(kind name, idx, params)])"""
    if name == "dillo":
        g = np.load(os.path.join(GOLD, "traj_dillo_nh.npz")); n = g["x"].shape[0]
        return float(g["dt"]), g["x"], np.full(3 * n, float(g["mass"])), [("TET_NH", g["tets"], [float(g["mu"]), float(g["lam"]), int(g["max_iter"])]), ("ANCHOR", g["anchors"], [-1.0, 1.0])]
    if name == "bunny":
        g = np.load(os.path.join(GOLD, "traj_bunny_stvk.npz")); n = g["x"].shape[0]
        return float(g["dt"]), g["x"], np.full(3 * n, float(g["mass"])), [("TET_STVK", g["tets"], [float(g["mu"]), float(g["lam"]), int(g["max_iter"])])]
    if name == "cloth":
        g = np.load(os.path.join(GOLD, "traj_cloth.npz")); n = g["x"].shape[0]
        return float(g["dt"]), g["x"], np.full(3 * n, float(g["mass"])), [("TRI_STRAIN", g["tris"], [float(g["k_tri"]), g["lim"][0], g["lim"][1], 1.0]),
                                                                        ("BEND", g["hinges"], [float(g["k_bend"])]), ("ANCHOR", g["anchors"], [-1.0, 1.0])]
    if name == "bar100k":
        dims = (16, 16, 65)
        x, t = pkg.meshgen.bar(*dims); m = pkg.meshgen.lumped_tet_mass(x, t, 1000.0)
        return 0.04, x, np.repeat(m, 3), [("TET_NH", t, [1e5, 1e5, 5]), ("ANCHOR", pkg.meshgen.bar_anchor_nodes(dims[0], dims[1]), [-1.0, 1.0])]
    raise KeyError(name)


def main():
    for name, seed, stride in (("dillo", 11, 1), ("bunny", 12, 1), ("cloth", 13, 1), ("bar100k", 14, 8)):
        dt, x, m3, forces = scene(name)
        r = Ref(); r.settings(dt, 1)
        r.add_nodes(x.ravel(), m3)
        for kind, idx, par in forces:
            r.add_forces(KIND[kind], idx, par)
        assert r.initialize()
        B = solve_rhs(seed, x, m3)
        X = np.stack([r.solve(b) for b in B])
        out = dict(seed=seed, stride=stride, dt=dt, dof=X.shape[1], x=X[:, ::stride].copy(), x_max=np.abs(X).max(axis=1), x_sum=X.sum(axis=1), x_sumsq=(X * X).sum(axis=1),
                   what="x = solver.solve(b) of the compiled reference (System.cpp:62) on checkers.solve_rhs(seed, x0, m3)")
        if name == "dillo":      # System::recompute_weights (System.cpp:159-179) with edited Force::weight members
            nt, na = forces[0][1].shape[0], len(forces[1][1])
            w = np.array([r.lib.ref_force_weight(r.h, i) for i in range(nt + na)])
            w_new = w.copy(); w_new[:nt:3] *= 2.0; w_new[nt:] = 0.0
            for i in range(nt + na):
                r.set_force_weight(i, float(w_new[i]))
            r.recompute_weights()
            X2 = np.stack([r.solve(b) for b in B])
            out.update(w_before=w, w_after=w_new, x_after=X2, n_tets=nt)
            assert np.abs(X2 - X).max() > 1e-3 * np.abs(X).max()
        np.savez_compressed(os.path.join(GOLD, "solve_%s.npz" % name), **out)
        print("solve_%s.npz: dof %d, |x|max per rhs %s" % (name, X.shape[1], out["x_max"]))


if __name__ == "__main__":
    main()
