"""CPU, build container only: the oracle against the COMPILED REFERENCE on
random inputs (skipped where oracle/_ref/libadmm_ref.so does not exist).
This is how the oracle is pinned beyond the committed fixtures."""
import numpy as np
import pytest

import checkers
from checkers import KIND, KIND_ROWS, Oracle, Ref

pytestmark = pytest.mark.skipif(not checkers.have_ref(), reason="compiled reference not present (oracle/_ref)")


def test_svd_bit_exact():
    rng = np.random.default_rng(0)
    for t in range(3000):
        F = rng.normal(size=9) * 10 ** rng.uniform(-3, 3)
        if t % 5 == 0:
            F = (np.eye(3) + 0.3 * rng.normal(size=(3, 3))).ravel()
        if t % 17 == 0:
            F[3:6] = F[0:3]
        if t % 501 == 0:
            F[:] = 0
        for a, b in zip(Ref.svd3(F), Oracle.svd3(F)):
            assert np.array_equal(a, b)
        G = rng.normal(size=6) * 10 ** rng.uniform(-3, 3)
        for a, b in zip(Ref.svd32(G), Oracle.svd32(G)):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("name,params", [("TET_LINEAR", [10.0]), ("TET_VOLUME", [100.0, 0.9, 1.1]), ("TET_NH", [100.0, 150.0, 5]), ("TET_STVK", [3e3, 1e3, 9]),
                                         ("TRI_STRAIN", [100.0, 0.95, 1.05, 1.0]), ("TRI_AREA", [100.0, 4, 0.9, 1.1]), ("TRI_FUNG", [50.0, 0.5, 2.0])])
def test_project_corner_cases_bit_exact(name, params):
    """The oracle against the compiled reference on inputs at the edges of the format (checkers.extreme_matrices: every decade of scale,
    entries 300 decades apart, rank 0 / 1 / 2, 2x2 blocks at the Jacobi rotation threshold incl. t == 0, infinities, NaNs): the same
    matrices the GPU kernels are checked on against the oracle (test_gpu_parity.py::test_svd_and_prox_corner_cases_bit_exact)."""
    import warnings
    kind = KIND[name]
    rng = np.random.default_rng(77 + kind)
    Dx = np.ascontiguousarray(checkers.extreme_matrices(rng, 256)[:, :KIND_ROWS[kind]])
    x_rest = np.array([0.0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1])[:3 * checkers.KIND_NODES[kind]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for e in range(Dx.shape[0]):
            X = x_rest.reshape(-1, 3) + e * np.array([3.0, 0, 0])
            a = Ref.project_single(kind, X.ravel(), params, Dx[e]); b = Oracle.project_single(kind, X.ravel(), params, Dx[e])
            assert np.array_equal(a["z"], b["z"], equal_nan=True), (name, e, Dx[e], a["z"], b["z"])
            assert np.array_equal(a["u"], b["u"], equal_nan=True), (name, e)
            if name in ("TET_NH", "TET_STVK"):
                assert np.array_equal(a["state"], b["state"], equal_nan=True), (name, e)
            if name in ("TET_NH", "TET_STVK", "TRI_FUNG"):
                assert np.array_equal(a["n_iters"], b["n_iters"]), (name, e)


CASES = [("TET_NH", [1e5, 1e5, 5]), ("TET_NH", [50, 80, 20]), ("TET_STVK", [100, 100, 5]), ("TET_STVK", [3e3, 1e3, 12]), ("TET_LINEAR", [1.0]),
         ("TET_VOLUME", [100, 0.9, 1.1]), ("TRI_STRAIN", [100, .95, 1.05, 1]), ("TRI_STRAIN", [10, .5, 2, 0]), ("BEND", [20.]), ("SPRING", [50.]),
         ("ANCHOR", [55., 1]), ("TRI_AREA", [100., 4, .9, 1.1]), ("TRI_AREA", [7., 1, 1.0, 1.0]), ("TRI_FUNG", [50., 0.5, 2.0]), ("TRI_FUNG", [2e3, 0, 0])]


@pytest.mark.parametrize("name,params", CASES)
def test_project_bit_exact(name, params):
    kind = KIND[name]
    rng = np.random.default_rng(kind * 7 + len(params))
    rows = KIND_ROWS[kind]
    for t in range(300):
        while True:
            x = rng.normal(size=(4, 3)) * rng.uniform(0.05, 2)
            if abs(np.linalg.det(np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]]))) > 1e-3 * np.abs(x).max() ** 3:
                break
        amp = rng.choice([0.0, 1e-8, 0.01, 0.1, 0.3, 0.6])
        Dx = []
        for c in range(5):
            if rows == 9 and name != "BEND":
                A = np.eye(3) + amp * rng.normal(size=(3, 3))
                if rng.uniform() < 0.1:
                    A[:, 2] *= -1
                Dx.append(A.ravel(order="F"))
            elif rows == 6:
                Dx.append((np.eye(3)[:, :2] + amp * rng.normal(size=(3, 2))).ravel(order="F"))
            else:
                Dx.append(rng.normal(size=rows) * (1 + amp))
        u0 = rng.normal(size=rows) * rng.choice([0, 0.01, 0.1])
        a = Ref.project_single(kind, x, params, np.array(Dx), u0)
        b = Oracle.project_single(kind, x, params, np.array(Dx), u0)
        keys = ("z", "u", "init") + (("state", "n_iters") if name in ("TET_NH", "TET_STVK", "TRI_FUNG") else ())
        for k in keys:
            if k == "state" and name == "TRI_FUNG":   # only the solver's Hessian guess persists
                assert a[k][3] == b[k][3], (name, t, k)
                continue
            assert np.array_equal(a[k], b[k], equal_nan=True), (name, t, k)


def test_system_assembly_and_first_iteration():
    from __graft_entry__ import load_package
    mg = load_package().meshgen
    x, t = mg.bar(3, 3, 8)
    m = mg.lumped_tet_mass(x, t, 1000.0)
    sysm = []
    for S in (Ref, lambda: Oracle(True)):
        s = S(); s.settings(0.04, 1)
        s.add_nodes(x.ravel(), np.repeat(m, 3))
        s.add_forces(KIND["TET_STVK"], t, [1e5, 1e5, 5])
        s.add_forces(KIND["ANCHOR"], mg.bar_anchor_nodes(3, 3), [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        assert s.initialize()
        sysm.append(s)
    r, o = sysm
    assert np.array_equal(r.global_idx(), o.global_idx()) and np.array_equal(r.wdiag, o.wdiag)
    r.step(); o.step()
    assert np.abs(r.x - o.x).max() < 1e-13


def test_collision_and_explicit_subset_vs_ref():
    from __graft_entry__ import load_package
    mg = load_package().meshgen
    x, t = mg.bar(2, 2, 4, h=0.1)
    x = x + np.array([0.0, 0.5, 0.0])
    m = mg.lumped_tet_mass(x, t, 100.0)
    n = x.shape[0]
    types = np.array([2, 1, 0], dtype=np.int32)
    params = np.array([[0.1, 0.35, 0, 0.12], [0.1, 0.1, 0.2, 0.15], [0, -0.1, 0, 0]], dtype=np.float64)
    sub = np.arange(0, n, 3, dtype=np.int32)
    r = Ref(); o = Oracle(True)
    for s in (r, o):
        s.settings(0.02, 10)
        s.add_nodes(x.ravel(), np.repeat(m, 3))
        s.add_forces(KIND["TET_LINEAR"], t, [800.0])
    r.add_collision(types, params, 32.0)
    o.add_forces(KIND["COLLISION"], np.arange(n), [32.0]); o.set_collision_shapes(types, params)
    r.add_gravity([0, -9.8, 0]); o.add_gravity([0, -9.8, 0])
    r.add_explicit_subset(sub, [0.3, 0, 0.1]); o.add_explicit(0, [0.3, 0, 0.1], sub)
    assert r.initialize() and o.initialize()
    assert r.rows == o.rows and np.array_equal(r.wdiag, o.wdiag)
    # the reference's single CollisionForce = our batch of one element per node
    assert r.global_idx()[-1] == o.global_idx()[t.shape[0]]
    for f in range(25):
        r.step(); o.step()
        assert np.abs(r.x - o.x).max() < 1e-11, f
    assert r.x.reshape(-1, 3)[:, 1].min() > -0.1 - 1e-3     # the floor holds


WIND_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from checkers import KIND, Oracle, Ref
from __graft_entry__ import load_package
mg = load_package().meshgen
x, tris = mg.sym_plane(6, 4, size=1.0)
hinges = mg.bend_hinges(tris)
n = x.shape[0]
r = Ref(); o = Oracle()
for s in (r, o):
    s.settings(0.04, 10)
    s.add_nodes(x.ravel(), np.full(3 * n, 0.5 / n))
    s.add_forces(KIND["TRI_STRAIN"], tris, [100.0, 0.95, 1.05, 1.0])
    s.add_forces(KIND["BEND"], hinges, [20.0])
    s.add_forces(KIND["ANCHOR"], [0, 6], [-1.0, 1.0])
    s.add_gravity([0, -9.8, 0])
r.add_wind(tris, [10, 0, 2]); o.add_explicit(1, [10, 0, 2], tris)
assert r.initialize() and o.initialize()
worst = 0.0
for f in range(5):
    r.step(); o.step()
    worst = max(worst, np.abs(r.x - o.x).max())
print("WORST", worst)
"""


def test_wind_vs_ref():
    """WindForce: the reference scatters under an omp critical and reads velocities other
    threads are updating (ExplicitForce.cpp:49-95), so with several threads it is not
    reproducible beyond ~1e-2.  Its serial execution (OMP_NUM_THREADS=1) is what the oracle
    restates: tight there, loose against the threaded run."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = WIND_CHILD % (os.path.join(root, "tests"), root)
    out1 = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
    assert out1.returncode == 0, out1.stderr[-1500:]
    assert float(out1.stdout.split("WORST")[1]) < 1e-10
    outn = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS="8"), capture_output=True, text=True)
    assert outn.returncode == 0, outn.stderr[-1500:]
    assert float(outn.stdout.split("WORST")[1]) < 0.1


def test_port_speed_within_20_percent_of_reference(pkg):
    """the C restatement may stand in for the reference as bench.py's cpu_baseline (kind "port")
    only if it is as fast as the reference, not faster by construction: same sample as bench.py's cpu_baseline (NH bar
    16x16x65 = 99,840 tets, 20 ADMM iterations per frame), both on this container's cores, ms per ADMM iteration within
    +-20 %.  (Measured here: port 127 ms, reference 138 ms; nnz(L) 17.3 M natural order vs 14.3 M AMD.)"""
    import os
    if (os.cpu_count() or 1) < 4:
        pytest.skip("needs a few cores to be meaningful")
    dims = (16, 16, 65)
    x, t = pkg.meshgen.bar(*dims)
    m = pkg.meshgen.lumped_tet_mass(x, t, 1000.0)
    # both systems alive, the same frames timed in alternation (whatever else loads this machine hits both), median of the per-frame ratios
    sims = {}
    for name, cls in (("port", Oracle), ("reference", Ref)):
        s = cls(); s.settings(0.04, 20)
        s.add_nodes(x.ravel(), np.repeat(m, 3))
        s.add_forces(KIND["TET_NH"], t, [1e5, 1e5, 5])
        s.add_forces(KIND["ANCHOR"], pkg.meshgen.bar_anchor_nodes(dims[0], dims[1]), [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        assert s.initialize()
        s.time_steps(1)
        sims[name] = s
    ms = {"port": [], "reference": []}
    for _ in range(4):
        for name in ("port", "reference"):
            ms[name].append(1e3 * sims[name].time_steps(1) / 20)
    ratio = float(np.median(np.array(ms["port"]) / np.array(ms["reference"])))
    assert 0.8 <= ratio <= 1.2, ms
