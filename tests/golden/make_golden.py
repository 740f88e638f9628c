#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/ from the COMPILED REFERENCE
(oracle/_ref/libadmm_ref.so, built by oracle/Makefile from /root/reference).

Run in the build container only (the reference does not travel):

    python tests/golden/make_golden.py

Every .npz holds inputs and the reference's outputs -- data only.  The two mesh
files (dillo919, bunny_1124) are the reference's own sample data
(samples/poordillo, samples/bunnyexpand), stored as arrays.

Fixtures
  project_<KIND>.npz   per-Force::project tuples: 256 elements x 4 consecutive
                       calls (u and warm-start state carried like System::step does)
  known_answers.npz    the reference's own two printed answers (singletet, singlenode)
  traj_bar_<nh|stvk>.npz  4x4x12-cube bar, 3 frames x 20 iters, + the reference's
                       sensitivity to a 1-ulp input perturbation (the parity envelope)
  traj_dillo_nh.npz    poordillo mesh (2761 NH tets), gravity + anchored hand/foot, 3 frames (the reference bifurcates at frame 4 under 1-ulp perturbations)
  traj_bunny_stvk.npz  bunnyexpand mesh (2510 StVK tets), x scaled x1.3 after initialize, 2 frames (chaotic afterwards: 1-ulp sensitivity > 1e-3)
  traj_cloth.npz       30x20 sym-plane cloth, TriangleStrain + Bend + 2 anchors, 3 frames x 30 iters
  traj_bar_nh_5400.npz, traj_bar_stvk_50700.npz   the throughput sizes of BASELINE.json configs[1] / [2] (make_baseline_bars)
  assembly_bar.npz     global_idx, W diagonal and D triplets in the reference's own row layout
  traj_collision.npz   plinkopony-like: corotational tets falling on cylinders / sphere / floor (CollisionForce)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from checkers import KIND, KIND_ROWS, Ref  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
meshgen = pkg.meshgen
REF_SAMPLES = "/root/reference/samples"


def rand_tet(rng):
    while True:
        x = rng.normal(size=(4, 3)) * rng.uniform(0.05, 2)
        vol = np.linalg.det(np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]]))
        if abs(vol) > 1e-3 * np.abs(x).max() ** 3:
            return x


def gen_Dx(rng, kind, ncalls, amp):
    rows = KIND_ROWS[kind]
    out = []
    for _ in range(ncalls):
        if rows == 9 and kind != KIND["BEND"]:
            A = np.eye(3) + amp * rng.normal(size=(3, 3))
            if rng.uniform() < 0.1:
                A[:, 2] *= -1
            out.append(A.ravel(order="F"))
        elif rows == 6:
            A = np.eye(3)[:, :2] + amp * rng.normal(size=(3, 2))
            out.append(A.ravel(order="F"))
        else:
            out.append(rng.normal(size=rows) * (1 + amp))
    return np.array(out)


PROJECT_CASES = {
    "TET_NH": [1e5, 1e5, 5], "TET_STVK": [100.0, 100.0, 5], "TET_LINEAR": [1e3], "TET_VOLUME": [100.0, 0.9, 1.1],
    "TRI_STRAIN": [100.0, 0.95, 1.05, 1], "BEND": [20.0], "SPRING": [50.0], "ANCHOR": [-1.0, 1.0],
    "TRI_AREA": [100.0, 4, 0.9, 1.1], "TRI_FUNG": [50.0, 0.5, 2.0],
}


def make_projects(only=None):
    for name, params in PROJECT_CASES.items():
        if only and name not in only:
            continue
        kind = KIND[name]
        rng = np.random.default_rng(1000 + kind)
        N, Cc = 256, 4
        rows = KIND_ROWS[kind]
        X = np.zeros((N, 4, 3)); DX = np.zeros((N, Cc, rows)); U0 = np.zeros((N, rows))
        Z = np.zeros((N, Cc, rows)); U = np.zeros((N, Cc, rows)); ST = np.zeros((N, 4)); IT = np.zeros((N, Cc), np.int32); INIT = np.zeros((N, 16))
        for e in range(N):
            x = rand_tet(rng)
            amp = rng.choice([0.0, 1e-8, 0.01, 0.1, 0.3, 0.6])
            Dx = gen_Dx(rng, kind, Cc, amp)
            u0 = rng.normal(size=rows) * rng.choice([0, 0.01, 0.1])
            r = Ref.project_single(kind, x, params, Dx, u0)
            X[e] = x; DX[e] = Dx; U0[e] = u0; Z[e] = r["z"]; U[e] = r["u"]; ST[e] = r["state"]; IT[e] = r["n_iters"]; INIT[e] = r["init"]
        np.savez_compressed(os.path.join(HERE, "project_%s.npz" % name), kind=kind, params=np.array(params, dtype=np.float64), x_rest=X, Dx=DX, u0=U0,
                            z=Z, u=U, state=ST, n_iters=IT, init=INIT)
        print("project", name, "iters hist", np.bincount(IT.ravel())[:12] if name in ("TET_NH", "TET_STVK", "TRI_FUNG") else "", "non-finite z:", int((~np.isfinite(Z)).any(axis=(1, 2)).sum()))


def make_known_answers():
    # singletet (deps/admm-elastic-sca/samples/singletet.cpp:27-111)
    r = Ref(); r.settings(1.0, 20)
    x = np.zeros(12); x[0 * 3 + 1] = 1; x[2 * 3 + 2] = 1; x[3 * 3 + 0] = 1
    r.add_nodes(x, np.ones(12))
    r.add_forces(KIND["ANCHOR"], [0, 1, 2], [-1.0, 1.0])
    r.add_forces(KIND["TET_LINEAR"], [[0, 1, 2, 3]], [1.0])
    assert r.initialize()
    xx = r.x; xx[9] = 200.0; r.x = xx
    r.step()
    tet_x = r.x
    # singlenode (deps/admm-elastic-sca/samples/singlenode.cpp:25-73)
    r = Ref(); r.settings(1.0, 20)
    r.add_nodes(np.zeros(3), np.ones(3))
    r.add_gravity([0.0, np.float32(-9.8), 0.0])
    assert r.initialize()
    ys = []
    for _ in range(4):
        r.step(); ys.append(r.x.copy())
    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), singletet_x=tet_x, singlenode_x=np.array(ys),
                        singletet_printed=171.57142857142716, singlenode_printed=np.array([-9.8, -29.4, -58.8, -98.0]))
    print("singletet node4 x", repr(tet_x[9]), "singlenode y", [y[1] for y in ys])


def perturb_x(s, seed, eps=2e-16):
    xx = s.x
    s.x = xx * (1 + eps * np.random.default_rng(seed).choice([-1.0, 1.0], size=xx.size))


def envelope(build, frames, seeds=(1, 2, 3, 4, 5)):
    """max over seeds of |x_ref(perturbed by 1 ulp) - x_ref| per frame: the
    reference's own sensitivity, i.e. the resolution at which a trajectory can
    be compared at all (its truncated L-BFGS + line search is discontinuous)."""
    base = build(); pert = []
    for sd in seeds:
        p = build(); perturb_x(p, sd); pert.append(p)
    env = []
    for _ in range(frames):
        base.step()
        xb = base.x
        e = 0.0
        for p in pert:
            p.step(); e = max(e, np.abs(p.x - xb).max())
        env.append(e)
    return np.array(env)


def bar_system(S, kind, dims, mu, lam, iters, dt=0.04, perturb=0.0):
    x, t = meshgen.bar(*dims)
    m = meshgen.lumped_tet_mass(x, t, 1000.0)
    s = S(); s.settings(dt, iters)
    s.add_nodes(x.ravel(), np.repeat(m, 3))
    s.add_forces(kind, t, [mu, lam, 5])
    s.add_forces(KIND["ANCHOR"], meshgen.bar_anchor_nodes(dims[0], dims[1]), [-1.0, 1.0])
    s.add_gravity([0, -9.8, 0])
    assert s.initialize()
    if perturb:
        xx = s.x
        sg = np.random.default_rng(5).choice([-1.0, 1.0], size=xx.size)
        s.x = xx * (1 + perturb * sg)
    return s


def make_bars():
    dims = (4, 4, 12)
    for name, kind in (("nh", KIND["TET_NH"]), ("stvk", KIND["TET_STVK"])):
        r = bar_system(Ref, kind, dims, 1e5, 1e5, 20)
        X = []
        for _ in range(3):
            r.step()
            X.append(r.x.copy())
        ENV = envelope(lambda: bar_system(Ref, kind, dims, 1e5, 1e5, 20), 3)
        # one-iteration frame: only solve rounding separates implementations
        r1 = bar_system(Ref, kind, dims, 1e5, 1e5, 1)
        r1.step()
        np.savez_compressed(os.path.join(HERE, "traj_bar_%s.npz" % name), dims=np.array(dims), kind=kind, mu=1e5, lam=1e5, max_iter=5, dt=0.04,
                            iters=20, x_frames=np.array(X), ulp_sensitivity=np.array(ENV), x_one_iter=r1.x, u_one_iter=r1.u, z_one_iter=r1.z,
                            global_idx=r1.global_idx())
        print("bar", name, "1-ulp sensitivity per frame", ENV)
    # assembly in the reference's row layout
    r = bar_system(Ref, KIND["TET_NH"], dims, 1e5, 1e5, 1)
    rr, rc, rv = r.D_triplets()
    k = np.lexsort((rr, rc))
    np.savez_compressed(os.path.join(HERE, "assembly_bar.npz"), dims=np.array(dims), global_idx=r.global_idx(), wdiag=r.wdiag, weights=r.weights(),
                        D_rows=rr[k], D_cols=rc[k], D_vals=rv[k], rows=r.rows)


def make_baseline_bars():
    """The two THROUGHPUT sizes of BASELINE.json configs[1] / configs[2] as SURVEY 8(d) fixes them (bench.py other_configs times them):
    Neo-Hookean bar 10x10x9 cubes = 5 400 tets, StVK bar 13x13x50 = 50 700 tets, mu = lambda = 1e5, 20 iterations per frame.
      x_frames / ulp_sensitivity   3 frames from rest + the reference's own 1-ulp sensitivity per frame (5 seeds)
      *_one_iter                   ONE ADMM iteration from checkers.deformed_start (+ and * only: the same bits everywhere), scaled so that the
                                   deformation gradients of these short bars sit 5-20 % off the identity: x, v after the frame; u, z (the 9 real
                                   rows of the reference's 36 per tet), warm-start state and L-BFGS iteration count of every `tet_stride`-th tet,
                                   u / z of every anchor -- the local step sees bit-identical input on every implementation."""
    from checkers import deformed_start
    for name, kind, dims, tstride, scale in (("nh_5400", KIND["TET_NH"], (10, 10, 9), 1, 10.0), ("stvk_50700", KIND["TET_STVK"], (13, 13, 50), 8, 3.0)):
        r = bar_system(Ref, kind, dims, 1e5, 1e5, 20)
        X = []
        for _ in range(3):
            r.step(); X.append(r.x.copy())
        ENV = envelope(lambda: bar_system(Ref, kind, dims, 1e5, 1e5, 20), 3)
        r1 = bar_system(Ref, kind, dims, 1e5, 1e5, 1)
        x0 = r1.x.copy()
        r1.x = deformed_start(x0 * scale) / scale      # (the polynomial terms act on the scaled coordinates: short bars deform like the 8 m one)
        r1.step()
        nt = 6 * dims[0] * dims[1] * dims[2]; na = (dims[0] + 1) * (dims[1] + 1)
        u = r1.u; z = r1.z
        assert u.size == 36 * nt + 3 * na
        ut = u[:36 * nt].reshape(nt, 36); zt = z[:36 * nt].reshape(nt, 36)
        assert not ut[:, 9:].any() and not zt[:, 9:].any()
        tets = np.arange(0, nt, tstride)
        st = np.zeros((tets.size, 4)); it = np.zeros(tets.size, np.int32)
        for k, e in enumerate(tets):
            st[k], it[k] = r1.hyper_state(int(e))
        np.savez_compressed(os.path.join(HERE, "traj_bar_%s.npz" % name), dims=np.array(dims), kind=kind, mu=1e5, lam=1e5, max_iter=5, dt=0.04, iters=20,
                            x_frames=np.array(X), ulp_sensitivity=np.array(ENV), start_scale=scale, tet_stride=tstride,
                            x_one_iter=r1.x, v_one_iter=r1.v, u_tets=ut[tets, :9], z_tets=zt[tets, :9], state_tets=st, n_iters_tets=it,
                            u_anchors=u[36 * nt:].reshape(na, 3), z_anchors=z[36 * nt:].reshape(na, 3))
        print("baseline bar", name, "1-ulp sensitivity per frame", ENV, "max |u| after the deformed iteration %.3e" % np.abs(ut).max(), "L-BFGS iterations", np.bincount(it))


def load_tetmesh(path_base, scale):
    """TetGen .node/.ele (first column = index).  Vertices pass through float32
    like trimesh2's Vec<3,float> (reference src/ForceBuilder.hpp:132-135)."""
    nodes = np.loadtxt(path_base + ".node", skiprows=1)
    eles = np.loadtxt(path_base + ".ele", skiprows=1, dtype=np.int64)
    x = (nodes[:, 1:4].astype(np.float32) * np.float32(scale)).astype(np.float32).astype(np.float64)
    return x, eles[:, 1:5].astype(np.int32)


def make_meshes():
    # poordillo: NH mu=lambda=1e5, maxIter 5, uniform mass 140/919, dt .06, 10 iters (samples/poordillo/poordillo.xml)
    x, t = load_tetmesh(os.path.join(REF_SAMPLES, "poordillo", "dillo919"), 0.01)
    n = x.shape[0]
    hand = np.where(np.linalg.norm(x - np.array([.6, .8, .5]), axis=1) < 0.2)[0]
    foot = np.where(np.linalg.norm(x - np.array([-.25, -.6, -.1]), axis=1) < 0.2)[0]
    anchors = np.concatenate([hand, foot]).astype(np.int32)
    def build():
        r = Ref(); r.settings(0.06, 10)
        r.add_nodes(x.ravel(), np.full(3 * n, 140.0 / n))
        r.add_forces(KIND["TET_NH"], t, [1e5, 1e5, 5])
        r.add_forces(KIND["ANCHOR"], anchors, [-1.0, 1.0])
        r.add_gravity([0, -9.8, 0])
        assert r.initialize()
        return r
    r = build()
    X = []
    for _ in range(3):
        r.step(); X.append(r.x.copy())
    env = envelope(build, 3)
    np.savez_compressed(os.path.join(HERE, "traj_dillo_nh.npz"), x=x, tets=t, anchors=anchors, mass=140.0 / n, mu=1e5, lam=1e5, max_iter=5, dt=0.06, iters=10,
                        x_frames=np.array(X), ulp_sensitivity=env)
    print("dillo envelope", env)
    print("dillo: nodes", n, "tets", t.shape[0], "anchors", anchors.size)
    # bunnyexpand: StVK mu=lambda=100, mass 1, dt .04, 10 iters; x scaled 1.3 about the origin after initialize
    x, t = load_tetmesh(os.path.join(REF_SAMPLES, "bunnyexpand", "bunny_1124"), 10.0)
    n = x.shape[0]
    def build2():
        r = Ref(); r.settings(0.04, 10)
        r.add_nodes(x.ravel(), np.full(3 * n, 1.0 / n))
        r.add_forces(KIND["TET_STVK"], t, [100.0, 100.0, 5])
        assert r.initialize()
        r.x = r.x * 1.3
        return r
    r = build2()
    X = []
    for _ in range(2):
        r.step(); X.append(r.x.copy())
    env = envelope(build2, 2)
    np.savez_compressed(os.path.join(HERE, "traj_bunny_stvk.npz"), x=x, tets=t, mass=1.0 / n, mu=100.0, lam=100.0, max_iter=5, dt=0.04, iters=10, scale=1.3,
                        x_frames=np.array(X), ulp_sensitivity=env)
    print("bunny envelope", env)
    print("bunny: nodes", n, "tets", t.shape[0])


def make_cloth():
    # windyflag-like: 30x20 sym plane, TriangleStrain k=100 limits .95/1.05, Bend k=20, 2 corner anchors, gravity, 30 iters (samples/windyflag/cloth.xml)
    w, l = 30, 20
    x, tris = meshgen.sym_plane(w, l, size=1.5)
    x = x.astype(np.float32).astype(np.float64)
    hinges = meshgen.bend_hinges(tris)
    n = x.shape[0]
    anchors = np.array([0, w], dtype=np.int32)

    def build():
        r = Ref(); r.settings(0.04, 30)
        r.add_nodes(x.ravel(), np.full(3 * n, 0.5 / n))
        r.add_forces(KIND["TRI_STRAIN"], tris, [100.0, 0.95, 1.05, 1.0])
        r.add_forces(KIND["BEND"], hinges, [20.0])
        r.add_forces(KIND["ANCHOR"], anchors, [-1.0, 1.0])
        r.add_gravity([0, -9.8, 0])
        assert r.initialize()
        return r
    r = build()
    X = []
    for _ in range(3):
        r.step(); X.append(r.x.copy())
    env = envelope(build, 3)
    np.savez_compressed(os.path.join(HERE, "traj_cloth.npz"), x=x, tris=tris, hinges=hinges, anchors=anchors, mass=0.5 / n, k_tri=100.0, lim=np.array([0.95, 1.05]),
                        k_bend=20.0, dt=0.04, iters=30, x_frames=np.array(X), ulp_sensitivity=env)
    print("cloth envelope", env)
    print("cloth: nodes", n, "tris", tris.shape[0], "hinges", hinges.shape[0])


def make_skin():
    """Two small membranes for the triangle kinds no sample instantiates:
    TriArea (area preservation) + Bend, and FungTriangle skin, both pinned on one edge under gravity."""
    w, l = 8, 6
    x, tris = meshgen.sym_plane(w, l, size=0.4)
    x = x.astype(np.float32).astype(np.float64)
    hinges = meshgen.bend_hinges(tris)
    n = x.shape[0]
    anchors = np.arange(0, w + 1, dtype=np.int32)
    for name, kind, par, extra, frames in (("triarea", "TRI_AREA", [100.0, 4, 0.95, 1.05], True, 4), ("fung", "TRI_FUNG", [40.0, 0.5, 2.0], False, 3)):
        def build():
            r = Ref(); r.settings(0.02, 15)
            r.add_nodes(x.ravel(), np.full(3 * n, 0.2 / n))
            r.add_forces(KIND[kind], tris, par)
            if extra:
                r.add_forces(KIND["BEND"], hinges, [5.0])
            r.add_forces(KIND["ANCHOR"], anchors, [-1.0, 1.0])
            r.add_gravity([0, -9.8, 0])
            assert r.initialize()
            return r
        r = build()
        X = []
        for _ in range(frames):
            r.step(); X.append(r.x.copy())
        env = envelope(build, frames)
        np.savez_compressed(os.path.join(HERE, "traj_skin_%s.npz" % name), x=x, tris=tris, hinges=hinges, anchors=anchors, mass=0.2 / n, kind=KIND[kind], params=np.array(par),
                            with_bend=extra, k_bend=5.0, dt=0.02, iters=15, x_frames=np.array(X), ulp_sensitivity=env, global_idx=r.global_idx(), wdiag_head=r.wdiag[:12])
        print("skin", name, "finite", np.isfinite(np.array(X)).all(), "sag", np.array(X)[-1].reshape(-1, 3)[:, 1].min(), "envelope", env)


def make_collision():
    """plinkopony-like (samples/plinkopony/plinkopony.cpp:53-96): a LinearTetStrain body falls on
    z-axis cylinders, a sphere and a floor; one CollisionForce over all nodes, weight 32."""
    dims = (3, 3, 5)
    x, t = meshgen.bar(*dims, h=0.1)
    x = x + np.array([0.0, 1.0, 0.0])
    m = meshgen.lumped_tet_mass(x, t, 100.0)
    types = np.array([2, 2, 1, 0], dtype=np.int32)
    params = np.array([[0.05, 0.6, 0, 0.12], [0.35, 0.45, 0, 0.1], [0.15, 0.2, 0.25, 0.15], [0, -0.2, 0, 0]], dtype=np.float64)

    def build():
        r = Ref(); r.settings(0.02, 15)
        r.add_nodes(x.ravel(), np.repeat(m, 3))
        r.add_forces(KIND["TET_LINEAR"], t, [1000.0])
        r.add_collision(types, params, 32.0)
        r.add_gravity([0, -9.8, 0])
        assert r.initialize()
        return r
    r = build()
    X = []
    for _ in range(40):
        r.step(); X.append(r.x.copy())
    X = np.array(X)
    keep = [0, 4, 9, 19, 29, 39]
    env = envelope(build, 40)
    np.savez_compressed(os.path.join(HERE, "traj_collision.npz"), x=x, tets=t, mass=m, types=types, params=params, weight=32.0, k=1000.0, dt=0.02, iters=15,
                        frames=np.array(keep), x_frames=X[keep], ulp_sensitivity=env[keep], global_idx=r.global_idx(), wdiag_tail=r.wdiag[-6:])
    print("collision: min y", X[-1].reshape(-1, 3)[:, 1].min(), "envelope", env[keep])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "skin":
        make_skin()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "baseline_bars":
        make_baseline_bars()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "projects":      # e.g.  make_golden.py projects TRI_AREA TRI_FUNG
        make_projects(sys.argv[2:])
        sys.exit(0)
    make_collision()
    make_projects()
    make_known_answers()
    make_bars()
    make_baseline_bars()
    make_meshes()
    make_cloth()
    make_skin()
