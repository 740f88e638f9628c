"""GPU parity tests (run with -m gpu on an MI355X): everything goes through the
C ABI (libadmm_hip.so) and is compared with the CPU oracle on the same inputs,
with the committed golden vectors from the compiled reference, and -- at the
benchmark's full size -- through size-independent properties.

Tolerances
  * local step of EVERY kind: BIT-EXACT vs the oracle and vs the
    reference's recorded project() tuples (Neo-Hookean -- the device evaluates
    glibc's log() algorithm, local_math.hpp admm_log --, StVK, corotational tet,
    tet volume, triangle strain / area, FungTriangle -- glibc's exp(), admm_exp --, bend,
    spring, anchors, collisions);
  * solve: residual <= 1e-11 relative against the library's own A, the device sweeps = their host restatement to 1e-10 (here);
    against the REFERENCE's solver.solve(b): tests/test_solve_parity.py (fixtures from the compiled reference, 1e-10);
  * one ADMM iteration (no chaos yet): 1e-11;  multi-frame trajectories:
    20 x the reference's own 1-ulp sensitivity (fixtures).
"""
import os

import numpy as np
import pytest

from checkers import KIND, KIND_NODES, KIND_ROWS, Oracle, extreme_matrices
from conftest import golden
from test_host_math import hm  # noqa: F401  (fixture: the host build of local_math.hpp + this host's libm)
from test_oracle_golden import tol

pytestmark = pytest.mark.gpu


def rand_tets(rng, n):
    xs = []
    while len(xs) < n:
        x = rng.normal(size=(4, 3)) * rng.uniform(0.05, 2)
        if abs(np.linalg.det(np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]]))) > 1e-3 * np.abs(x).max() ** 3:
            xs.append(x + 3 * rng.normal(size=3))
    return np.array(xs)


def build_disjoint(pkg, name, params, n, seed, shuffle=True):
    """n elements of one kind on their own nodes (4 nodes each); node ids of an
    element are shuffled so that the corner sorting of the device layout is exercised."""
    kind = KIND[name]
    rng = np.random.default_rng(seed)
    X = rand_tets(rng, n).reshape(-1, 3)
    nn = KIND_NODES[kind]
    perm = rng.permutation(4 * n) if shuffle else np.arange(4 * n)
    Xp = np.zeros_like(X); Xp[perm] = X            # node q of the element lives at id perm[q]
    idx = perm.reshape(n, 4)[:, :nn].astype(np.int32)
    s = pkg.System(device_id=0); s.set_timestep(0.04)
    o = Oracle(); o.settings(0.04, 1)
    m = np.ones(3 * 4 * n)
    s.add_nodes(Xp.ravel(), m); o.add_nodes(Xp.ravel(), m)
    s.add_forces(kind, idx, params); o.add_forces(kind, idx, params)
    s.initialize(); assert o.initialize()
    return s, o, Xp, idx, rng


def oracle_local_step(o, xcur, n, rows):
    """one local step of the oracle on x_cur (Dx = D x, then project each force)."""
    rr, cc, vv = o.D_triplets()
    Dx = np.zeros(o.rows)
    # column-ascending accumulation like Eigen's column-major product (System.cpp:54)
    k = np.lexsort((cc, rr))
    for r_, c_, v_ in zip(rr[k], cc[k], vv[k]):
        Dx[r_] += v_ * xcur[c_]
    u = o._view("u", o.rows); z = o._view("z", o.rows)
    import ctypes as C
    from checkers import _d
    for i in range(o.n_forces):
        f = o.lib.orc_get_force(o.h, i)
        g = f.contents.global_idx
        d = np.ascontiguousarray(Dx[g:g + rows]); uu = np.ascontiguousarray(u[g:g + rows]); zz = np.zeros(rows)
        o.lib.orc_force_project(f, 0.04, _d(d), _d(uu), _d(zz))
        u[g:g + rows] = uu; z[g:g + rows] = zz
    return u.copy().reshape(n, rows), z.copy().reshape(n, rows)


EXACT_CASES = [("TET_NH", [1e5, 1e5, 5]), ("TET_NH", [100.0, 150.0, 5]), ("TET_NH", [50.0, 80.0, 12]),
               ("TET_STVK", [100.0, 100.0, 5]), ("TET_STVK", [3e3, 1e3, 9]), ("TET_LINEAR", [10.0]), ("TET_VOLUME", [100.0, 0.9, 1.1]),
               ("BEND", [20.0]), ("SPRING", [50.0]), ("ANCHOR", [-1.0, 1.0]), ("TRI_AREA", [100.0, 4, 0.9, 1.1]), ("TRI_AREA", [30.0, 1, 1.0, 1.0]),
               ("TRI_FUNG", [50.0, 0.5, 2.0]), ("TRI_FUNG", [5.0, 0.5, 2.0]),
               ("TRI_STRAIN", [100.0, 0.95, 1.05, 1.0]), ("TRI_STRAIN", [40.0, 1.0, 1.0, 0.0])]


@pytest.mark.parametrize("name,params", EXACT_CASES)
def test_local_step_bit_exact(pkg, name, params):
    n = 777  # ragged: not a multiple of the 256-lane block
    s, o, X, idx, rng = build_disjoint(pkg, name, params, n, seed=KIND[name] + 10)
    rows = KIND_ROWS[KIND[name]]
    for it in range(4):
        amp = [0.0, 0.02, 0.3, 0.8][it]
        xcur = (X + amp * rng.normal(size=X.shape)).ravel()
        if it == 3:  # invert some elements (det F < 0 paths)
            xcur.reshape(-1, 3)[idx[::7, 0]] += 3.0
        s.local_step_only(xcur)
        g = s.read_local(0)
        u, z = oracle_local_step(o, xcur, n, rows)
        assert np.array_equal(g["z"], z, equal_nan=True), (name, it)
        assert np.array_equal(g["u"], u, equal_nan=True), (name, it)
        if name in ("TET_STVK", "TET_NH"):
            st = np.array([o.hyper_state(i)[0] for i in range(n)]); ni = np.array([o.hyper_state(i)[1] for i in range(n)])
            assert np.array_equal(g["state"], st, equal_nan=True) and np.array_equal(g["n_iters"], ni)


@pytest.mark.parametrize("tpb", [32, 16, 8])
@pytest.mark.parametrize("name,params", [("TET_NH", [100.0, 150.0, 5]), ("TET_STVK", [3e3, 1e3, 9])])
def test_fewer_tets_per_wave_is_bitwise_the_same(pkg, name, params, tpb):
    """ADMM_HIP_TPB (under-filled launches: `tpb` tets per one-wave block, the lanes beyond idle; block-level RHS pre-reduction,
    cost order and residual partials follow the block size): u, z, the warm start and the L-BFGS iteration counts of the local
    step are bitwise those of the 64-lane launch over several calls with carried state, and a full frame agrees to rounding
    (the block partial sums of the right-hand side meet in another order)."""
    n = 777
    def run(env):
        old = os.environ.get("ADMM_HIP_TPB")
        if env is None: os.environ.pop("ADMM_HIP_TPB", None)
        else: os.environ["ADMM_HIP_TPB"] = str(env)
        try:
            s, o, X, idx, rng = build_disjoint(pkg, name, params, n, seed=KIND[name] + 10)
        finally:
            if old is None: os.environ.pop("ADMM_HIP_TPB", None)
            else: os.environ["ADMM_HIP_TPB"] = old
        outs = []
        for it in range(4):
            amp = [0.0, 0.02, 0.3, 0.8][it]
            xcur = (X + amp * rng.normal(size=X.shape)).ravel()
            if it == 3:
                xcur.reshape(-1, 3)[idx[::7, 0]] += 3.0
            s.local_step_only(xcur)
            outs.append(s.read_local(0))
        s.step(3)
        return outs, s.m_x.copy()
    a, xa = run(None)
    b, xb = run(tpb)
    for ga, gb in zip(a, b):
        for k in ("u", "z", "state", "n_iters"):
            assert np.array_equal(ga[k], gb[k], equal_nan=True), (k, tpb)
    assert np.all(np.isfinite(xb)) and np.abs(xa - xb).max() <= 1e-9 * max(1.0, np.abs(xa).max())


@pytest.mark.parametrize("params", [[1e5, 1e5, 5], [100.0, 150.0, 5], [50.0, 80.0, 12]])
def test_local_step_neohookean(pkg, params):
    n = 1500
    s, o, X, idx, rng = build_disjoint(pkg, "TET_NH", params, n, seed=4)
    differ = 0; worst = 0.0; total = 0
    for it in range(4):
        amp = [0.0, 0.02, 0.3, 0.6][it]
        xcur = (X + amp * rng.normal(size=X.shape)).ravel()
        # keep both sides on identical inputs each iteration
        go = s.read_local(0)
        s.local_step_only(xcur)
        g = s.read_local(0)
        u, z = oracle_local_step(o, xcur, n, 9)
        ni = np.array([o.hyper_state(i)[1] for i in range(n)])
        same = ni == g["n_iters"]
        sc = np.maximum(1.0, np.abs(z).max(axis=1))
        err = np.abs(g["z"] - z).max(axis=1) / sc
        good = err == 0.0
        differ += int((~good).sum()); total += n
        assert np.isfinite(g["z"]).all() and same.all()
        # re-synchronise the device with the oracle so the next iteration starts from identical state
        st = np.array([o.hyper_state(i)[0] for i in range(n)])
        s.write_local(0, u=u, state=st)
    assert differ == 0, (differ, total)      # admm_log() is glibc's log: the whole prox is bit-identical


def test_local_step_triangle(pkg):
    n = 900
    s, o, X, idx, rng = build_disjoint(pkg, "TRI_STRAIN", [100.0, 0.95, 1.05, 1.0], n, seed=6)
    for it in range(3):
        xcur = (X + [0.0, 0.05, 0.3][it] * rng.normal(size=X.shape)).ravel()
        s.local_step_only(xcur)
        g = s.read_local(0)
        u, z = oracle_local_step(o, xcur, n, 6)
        assert np.array_equal(g["z"], z) and np.array_equal(g["u"], u)


def test_local_step_fung(pkg):
    """FungTriangle: bit-exact 3x2 Jacobi SVD, then an L-BFGS whose objective calls exp() -- glibc's algorithm on the
    device (admm_exp): every finite element bit-identical; the tuples the reference itself drives to NaN are skipped."""
    n = 1200
    s, o, X, idx, rng = build_disjoint(pkg, "TRI_FUNG", [50.0, 0.5, 2.0], n, seed=8)
    differ = 0; total = 0
    for it in range(4):
        xcur = (X + [0.0, 0.02, 0.1, 0.3][it] * rng.normal(size=X.shape)).ravel()
        s.local_step_only(xcur)
        g = s.read_local(0)
        u, z = oracle_local_step(o, xcur, n, 6)
        fin = np.isfinite(z).all(axis=1) & np.isfinite(u).all(axis=1)
        sc = np.maximum(1.0, np.abs(z[fin]).max(axis=1))
        err = np.abs(g["z"][fin] - z[fin]).max(axis=1) / sc
        differ += int((~(err == 0.0)).sum()); total += n
        assert np.isfinite(g["z"][fin]).all()
        st = np.array([o.hyper_state(i)[0] for i in range(n)])
        u[~fin] = 0.0; st[~np.isfinite(st)] = 1.0
        s.write_local(0, u=u, state=st)
        ou = o._view("u", o.rows); ou[:] = u.ravel()            # the oracle continues from the same (sanitised) u
    assert differ == 0, (differ, total)


@pytest.mark.parametrize("name", ["TET_STVK", "TET_LINEAR", "TET_VOLUME", "BEND", "SPRING", "ANCHOR", "TET_NH", "TRI_STRAIN", "TRI_AREA", "TRI_FUNG"])
def test_golden_project_tuples(pkg, name):
    """The committed per-project vectors captured from the COMPILED REFERENCE
    (tests/golden/project_*.npz), replayed through the GPU kernels: every fixture
    element is one element of a batch, its recorded D_i x rows are fed through
    admm_hip_local_step_dx, u and the warm-start state carry over on the device
    exactly like across ADMM iterations."""
    g = golden("project_%s.npz" % name)
    kind = int(g["kind"]); nn = KIND_NODES[kind]
    N = g["x_rest"].shape[0]
    X = g["x_rest"].reshape(-1, 3)
    idx = np.arange(4 * N, dtype=np.int32).reshape(N, 4)[:, :nn]
    s = pkg.System(device_id=0); s.set_timestep(0.04)
    s.add_nodes(X.ravel(), np.ones(3 * 4 * N))
    s.add_forces(kind, idx, g["params"])
    s.initialize()
    rest = s.read_rest(0)
    assert np.array_equal(rest["weight"], g["init"][:, 0])            # incl. the fp32 sqrtf path
    if name.startswith("TET"):
        assert np.array_equal(rest["rest"], g["init"][:, 1:13])       # B, bit-exact
    s.write_local(0, u=g["u0"])
    for c in range(g["Dx"].shape[1]):                     # every kind: bit for bit what the compiled reference produced
        s.local_step_dx(0, g["Dx"][:, c])
        out = s.read_local(0)
        assert np.array_equal(out["z"], g["z"][:, c], equal_nan=True), (name, c)
        assert np.array_equal(out["u"], g["u"][:, c], equal_nan=True), (name, c)
        if name in ("TET_STVK", "TET_NH"):
            assert np.array_equal(out["n_iters"], g["n_iters"][:, c])
    if name in ("TET_STVK", "TET_NH"):
        assert np.array_equal(s.read_local(0)["state"], g["state"], equal_nan=True)


@pytest.mark.parametrize("name,params", [("TET_LINEAR", [10.0]), ("TET_VOLUME", [100.0, 0.9, 1.1]), ("TET_NH", [100.0, 150.0, 5]), ("TET_STVK", [3e3, 1e3, 9]),
                                         ("TRI_STRAIN", [100.0, 0.95, 1.05, 1.0]), ("TRI_AREA", [100.0, 4, 0.9, 1.1]), ("TRI_FUNG", [50.0, 0.5, 2.0])])
def test_svd_and_prox_corner_cases_bit_exact(pkg, name, params):
    """The Jacobi SVD and the proxes on inputs at the edges of the format (checkers.extreme_matrices) fed straight into the kernels
    (admm_hip_local_step_dx): u, z, warm start and iteration counts bit for bit the oracle's -- the round-4 kernels use the
    compiler's sqrt / reciprocal expansions WITHOUT their rescaling and special-case wrappers where the argument's range is
    known; this is where a wrong range assumption would show."""
    import warnings
    kind = KIND[name]
    n = 256
    rng = np.random.default_rng(77 + kind)
    rows = KIND_ROWS[kind]; nn = KIND_NODES[kind]
    Dx = np.ascontiguousarray(extreme_matrices(rng, n)[:, :rows])      # (triangles: the first two columns, a 3x2 matrix)
    x_rest = np.array([0.0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1])
    X = np.tile(x_rest.reshape(4, 3), (n, 1)) + np.repeat(np.arange(n), 4)[:, None] * np.array([3.0, 0, 0])
    idx = np.arange(4 * n, dtype=np.int32).reshape(n, 4)[:, :nn]
    s = pkg.System(device_id=0); s.set_timestep(0.04)
    s.add_nodes(X.ravel(), np.ones(3 * 4 * n))
    s.add_forces(kind, idx, params)
    s.initialize()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s.local_step_dx(0, Dx)
        out = s.read_local(0)
        for e in range(n):
            o = Oracle.project_single(kind, X[4 * e:4 * e + nn].ravel(), params, Dx[e])
            assert np.array_equal(out["z"][e], o["z"][0], equal_nan=True), (name, e, Dx[e], out["z"][e], o["z"][0])
            assert np.array_equal(out["u"][e], o["u"][0], equal_nan=True), (name, e)
            if name in ("TET_NH", "TET_STVK"):
                assert np.array_equal(out["state"][e], o["state"], equal_nan=True), (name, e, Dx[e])
            if name in ("TET_NH", "TET_STVK", "TRI_FUNG"):
                assert out["n_iters"][e] == o["n_iters"][0], (name, e)


def _bar_pair(pkg, kind, dims, iters):
    mg = pkg.meshgen
    x, t = mg.bar(*dims)
    m = mg.lumped_tet_mass(x, t, 1000.0)
    s = pkg.make_bar_system(*dims, kind=kind)
    s.initialize()
    o = Oracle(); o.settings(0.04, iters)
    o.add_nodes(x.ravel(), np.repeat(m, 3))
    o.add_forces(kind, t, [1e5, 1e5, 5])
    o.add_forces(KIND["ANCHOR"], mg.bar_anchor_nodes(dims[0], dims[1]), [-1.0, 1.0])
    o.add_gravity([0, -9.8, 0])
    assert o.initialize()
    return s, o


@pytest.mark.parametrize("dense_max,leaf", [("0", "16"), ("0", "0"), ("2048", "0")])
def test_solve_residual_and_device_sweeps_vs_host_sweeps(pkg, monkeypatch, dense_max, leaf):
    """Self-consistency of the solve (the reference's solve is pinned in tests/test_solve_parity.py): ||A x - b|| with the library's own
    assembled A, the device sweeps against their host restatement, bitwise reproducibility.  Both solve paths: the supernodal panel sweeps (dense_max 0; leaf 16 = deep tree of narrow supernodes, wave items; leaf 0 =
    automatic, here 256: wide supernodes, block items) and, for small systems, x = A^-1 b with the explicit inverse"""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", dense_max)
    monkeypatch.setenv("ADMM_HIP_LEAF", leaf)
    s, o = _bar_pair(pkg, KIND["TET_NH"], (6, 5, 17), 1)
    n = s.n_nodes
    assert s.info()["dense_solve"] == (1 if dense_max != "0" else 0)
    rng = np.random.default_rng(0)
    for _ in range(3):
        b = rng.normal(size=3 * n)
        x = s.solve_only(b)
        assert np.abs(s.apply_A(x) - b).max() < 1e-11 * np.abs(b).max()
        assert np.abs(x - s.debug_panel_solve_host(b)).max() < 1e-10 * np.abs(x).max()
    # bitwise reproducible (fixed reduction orders, no atomics)
    b = rng.normal(size=3 * n)
    assert np.array_equal(s.solve_only(b), s.solve_only(b))


@pytest.mark.parametrize("nw,cw2,small_nw", [("16", True, "4"), ("16", False, "8"), ("8", True, "2"), ("4", True, "4"), ("4", False, "4")])
def test_backward_kernel_shapes(pkg, monkeypatch, nw, cw2, small_nw):
    """Every (columns per wave, waves per block) pair the documented knobs can produce has a kernel: the level's work items
    are cut for bwd_nw * bwd_cw columns per block and the dispatch must launch exactly that instantiation (a missing branch
    used to fall through to a narrower kernel and leave columns of x unwritten -- ADVICE r2).  A x = b with the knobs forced
    onto every level of a bar with wide and narrow supernodes, and the default shapes' solution to rounding."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "64")
    dims = (12, 12, 30)
    ref = pkg.make_bar_system(*dims, device_id=0); ref.initialize()
    monkeypatch.setenv("ADMM_HIP_BWD_NW", nw)
    monkeypatch.setenv("ADMM_HIP_BWD_NW_MIN_COLS", "0")
    monkeypatch.setenv("ADMM_HIP_BWD_SMALL_NW", small_nw)
    if cw2:
        monkeypatch.setenv("ADMM_HIP_BWD_CW2_MIN", "1"); monkeypatch.setenv("ADMM_HIP_BWD_CW2_MAX", "100000000")
    else:
        monkeypatch.setenv("ADMM_HIP_BWD_CW2_MIN", "0")
    s = pkg.make_bar_system(*dims, device_id=0); s.initialize()
    n = s.n_nodes
    rng = np.random.default_rng(11)
    for _ in range(2):
        b = rng.normal(size=3 * n)
        x = s.solve_only(b)
        assert np.isfinite(x).all()
        assert np.abs(s.apply_A(x) - b).max() < 1e-11 * np.abs(b).max()
        # (not bitwise the default shapes' result: a block's first column decides which lane sums which rows)
        assert np.abs(x - ref.solve_only(b)).max() < 1e-12 * np.abs(x).max()
        assert np.array_equal(x, s.solve_only(b))


def test_prereduced_rhs_and_unkept_z(pkg, monkeypatch):
    """Round-3 changes of the tet kernels' epilogue that must not change what is computed:
    * block-level pre-reduction of the RHS shares (one slot per (64-tet block, node), summed in LDS in a fixed order) against
      one slot per corner (ADMM_HIP_PRERED=0): the local step's own outputs u, z, state are bitwise the same, the assembled
      right-hand side differs only in the order of its per-node sums (one iteration: rounding level), runs are bitwise reproducible;
    * admm_hip_keep_z(0) -- production frames do not store the tets' z, nobody reads it back (reference: curr_z is overwritten by
      every project(), System.cpp:57-58) -- leaves x, v, u and the warm-start state bitwise untouched; the parity entry point
      (local_step_only) still delivers z."""
    dims = (5, 4, 11)            # 1320 tets: 20 full blocks + one of 40 tets
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")

    def make(kind="TET_NH"):
        s = pkg.make_bar_system(*dims, kind=KIND[kind], device_id=0); s.initialize(); return s
    a = make(); a2 = make()
    monkeypatch.setenv("ADMM_HIP_PRERED", "0")
    b = make()
    monkeypatch.delenv("ADMM_HIP_PRERED")
    x, _ = pkg.meshgen.bar(*dims)
    xs = x.ravel() * (1.0 + 0.01 * np.sin(np.arange(x.size)))
    a.local_step_only(xs); a2.local_step_only(xs); b.local_step_only(xs)
    la, lb = a.read_local(0), b.read_local(0)
    for k in ("u", "z", "state"):
        assert np.array_equal(la[k], lb[k]), k
    a.step(1); a2.step(1); b.step(1)
    assert np.array_equal(a.m_x, a2.m_x)
    assert np.abs(a.m_x - b.m_x).max() < 1e-12
    # z not kept: same trajectory, same u / state; z is whatever the last keeping call left
    c = make(); c.keep_z(False)
    d = make()
    for _ in range(2):
        c.step(5); d.step(5)
    assert np.array_equal(c.m_x, d.m_x) and np.array_equal(c.m_v, d.m_v)
    lc, ld = c.read_local(0), d.read_local(0)
    assert np.array_equal(lc["u"], ld["u"]) and np.array_equal(lc["state"], ld["state"])
    c.local_step_only(xs); d.local_step_only(xs)
    assert np.array_equal(c.read_local(0)["z"], d.read_local(0)["z"])


@pytest.mark.parametrize("dims,leaf", [((6, 5, 17), "16"), ((8, 8, 40), "0"), ((12, 12, 30), "64")])
def test_device_factorization_vs_host(pkg, monkeypatch, dims, leaf):
    """The numeric multifrontal factorization on the GPU (csrc/factor_dev.hpp: MFMA fp64 products per 64-column block, block
    inverses, doubling for L11^-1) against the host factorization of the same symbolic structure: the panels agree to rounding,
    the solves agree, and A x = b holds with the device's factor (System.cpp:138-140 is what both replace)."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", leaf)
    nx, ny, nz = dims
    monkeypatch.setenv("ADMM_HIP_FACTOR", "host")
    sh = pkg.make_bar_system(nx, ny, nz, device_id=0)
    sh.initialize()
    monkeypatch.delenv("ADMM_HIP_FACTOR")
    sd = pkg.make_bar_system(nx, ny, nz, device_id=0)
    sd.initialize()
    n = sd.n_nodes
    rng = np.random.default_rng(3)
    for _ in range(2):
        b = rng.normal(size=3 * n)
        xd, xh = sd.solve_only(b), sh.solve_only(b)
        assert np.abs(sd.apply_A(xd) - b).max() < 1e-11 * np.abs(b).max()
        assert np.abs(xd - xh).max() < 1e-11 * np.abs(xh).max()
        assert np.abs(xd - sd.debug_panel_solve_host(b)).max() < 1e-10 * np.abs(xd).max()       # host sweeps over the DEVICE's panels
    # re-factorization after a weight change goes through the same path
    for s in (sd, sh):
        s.set_weights(1, np.full(s.batch_sizes[1] if hasattr(s, "batch_sizes") else len(pkg.meshgen.bar_anchor_nodes(nx, ny)), 3.0))
        s.recompute_weights()
    b = rng.normal(size=3 * n)
    xd, xh = sd.solve_only(b), sh.solve_only(b)
    assert np.abs(sd.apply_A(xd) - b).max() < 1e-11 * np.abs(b).max()
    assert np.abs(xd - xh).max() < 1e-11 * np.abs(xh).max()
    # a frame of the simulation: the two factors differ in the last bits, and the Neo-Hookean line search amplifies last-bit
    # differences of its input to ~1e-6..1e-5 within a frame (the reference against itself with a 1-ulp perturbed input:
    # DESIGN.md section 3) -- same bound as the smoke test
    sd.step(5); sh.step(5)
    assert np.abs(sd.m_x - sh.m_x).max() < 5e-5


@pytest.mark.parametrize("m,n,k", [(64, 64, 64), (1, 1, 1), (130, 70, 37), (200, 200, 64), (65, 129, 200), (333, 5, 17)])
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_gemm_f64_kernel(pkg, m, n, k, flags):
    """The fp64 MFMA product kernel of the device factorization (csrc/factor_dev.hpp gemm_f64_kernel) against numpy: every
    operand orientation, ragged sizes, alpha / beta, leading dimensions larger than the extents."""
    s = pkg.make_bar_system(2, 2, 3); s.initialize()
    rng = np.random.default_rng(m * 1000 + n * 10 + k + flags)
    ta, tb = flags & 1, flags & 2
    A = np.asfortranarray(rng.normal(size=((k if ta else m) + 3, (m if ta else k))))      # leading dimension = extent + 3
    B = np.asfortranarray(rng.normal(size=((n if tb else k) + 2, (k if tb else n))))
    Cm = np.asfortranarray(rng.normal(size=(m + 5, n)))
    opA = A[:k, :m].T if ta else A[:m, :k]
    opB = B[:n, :k].T if tb else B[:k, :n]
    want = Cm.copy(order="F")
    want[:m, :n] = 0.75 * want[:m, :n] - 1.5 * (opA @ opB)
    got = s.debug_gemm(A, B, Cm, m, n, k, flags=flags, alpha=-1.5, beta=0.75)
    assert np.array_equal(got[m:], want[m:])                       # rows beyond m untouched
    assert np.abs(got[:m] - want[:m]).max() <= 1e-13 * k * max(1.0, np.abs(want).max())


def test_gemm_f64_kernel_triangular_flags(pkg):
    """The structure flags: lower tiles only (symmetric update), K from the tile's column (lower triangular B), K from
    max(row, column) tile (A^T A of a lower triangular A)."""
    s = pkg.make_bar_system(2, 2, 3); s.initialize()
    rng = np.random.default_rng(11)
    n = 150
    Lo = np.asfortranarray(np.tril(rng.normal(size=(n, n))))
    A = np.asfortranarray(rng.normal(size=(n, 64)))
    C0 = np.asfortranarray(rng.normal(size=(n, n)))
    got = s.debug_gemm(A, A, C0.copy(order="F"), n, n, 64, flags=2 | 4, alpha=-1.0, beta=1.0)        # C -= A A^T, tiles on / below the diagonal
    want = C0 - A @ A.T
    for ti in range(3):
        for tj in range(3):
            blk = (slice(64 * ti, min(n, 64 * ti + 64)), slice(64 * tj, min(n, 64 * tj + 64)))
            if tj <= ti: assert np.abs(got[blk] - want[blk]).max() < 1e-12
            else: assert np.array_equal(got[blk], C0[blk])
    R = np.asfortranarray(rng.normal(size=(90, n)))
    got = s.debug_gemm(R, Lo, np.zeros((90, n), order="F"), 90, n, n, flags=8)                      # R Lo, Lo lower triangular
    assert np.abs(got - R @ Lo).max() < 1e-12 * n
    got = s.debug_gemm(Lo, Lo, np.zeros((n, n), order="F"), n, n, n, flags=1 | 16)                     # Lo^T Lo
    assert np.abs(got - Lo.T @ Lo).max() < 1e-12 * n


@pytest.mark.parametrize("w", [1, 2, 15, 16, 17, 33, 63, 64])
def test_potrf_inv_kernel(pkg, w):
    """Cholesky of a diagonal block and the factor's inverse (potrf_inv_kernel) against numpy; a block that is not positive
    definite is reported."""
    s = pkg.make_bar_system(2, 2, 3); s.initialize()
    rng = np.random.default_rng(w)
    M = rng.normal(size=(w, w + 3))
    S = M @ M.T + 0.1 * np.eye(w)
    Lg, Xg = s.debug_potrf_inv(S)
    Lw = np.linalg.cholesky(S)
    assert np.abs(Lg - Lw).max() < 1e-12 * np.abs(Lw).max()
    assert np.abs(Xg @ Lw - np.eye(w)).max() < 1e-10
    assert np.array_equal(np.triu(Xg, 1), np.zeros((w, w)))
    if w > 1:
        S[w - 1, w - 1] = -1.0
        with pytest.raises(Exception):
            s.debug_potrf_inv(S)


def test_frames_bitwise_reproducible(pkg):
    """Two fresh systems of one scene give bitwise equal states after several frames: the device factorization has a fixed summation
    order, the sweeps have no atomics, and the cost-ordered launch of the tet blocks (active here: 3 264 blocks) only decides when a
    block runs, never what it computes."""
    states = []
    for _ in range(2):
        s = pkg.make_bar_system(32, 32, 34)
        s.initialize()
        assert s.info()["device_factor"] == 1
        for _ in range(3):
            s.step(10)
        states.append((s.m_x.copy(), s.m_v.copy()))
        del s
    assert np.array_equal(states[0][0], states[1][0]) and np.array_equal(states[0][1], states[1][1])


def test_device_factorization_disconnected_mixed_scene(pkg, monkeypatch):
    """Two components (bar + cloth: two roots of the elimination tree, fronts from 1 to a few hundred rows, triangle / hinge /
    anchor elements next to the tets) through the device factorization and through the host one."""
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    monkeypatch.setenv("ADMM_HIP_LEAF", "32")
    systems = []
    for where in ("host", None):
        if where: monkeypatch.setenv("ADMM_HIP_FACTOR", where)
        else: monkeypatch.delenv("ADMM_HIP_FACTOR")
        s, _ = pkg.make_mixed_system(6, 6, 14, 14, 14, device_id=0)
        s.initialize()
        systems.append(s)
    sh, sd = systems
    assert sh.info()["device_factor"] == 0 and sd.info()["device_factor"] == 1
    n = sd.n_nodes
    rng = np.random.default_rng(5)
    for _ in range(2):
        b = rng.normal(size=3 * n)
        xd, xh = sd.solve_only(b), sh.solve_only(b)
        assert np.abs(sd.apply_A(xd) - b).max() < 1e-11 * np.abs(b).max()
        assert np.abs(xd - xh).max() < 1e-11 * np.abs(xh).max()


@pytest.mark.parametrize("name,kind", [("nh", "TET_NH"), ("stvk", "TET_STVK")])
def test_bar_one_iteration_and_trajectory(pkg, name, kind):
    g = golden("traj_bar_%s.npz" % name)
    dims = tuple(g["dims"])
    s, o = _bar_pair(pkg, KIND[kind], dims, 1)
    s.step(1); o.step()
    xg = s.m_x
    assert np.abs(xg - o.x).max() < 1e-11
    assert np.abs(xg - g["x_one_iter"]).max() < 1e-11       # golden: the compiled reference itself
    assert np.abs(s.m_v - o.v).max() < 1e-9
    s, o = _bar_pair(pkg, KIND[kind], dims, 20)
    for f in range(3):
        s.step(20); o.step()
        assert np.abs(s.m_x - g["x_frames"][f]).max() < tol(g, f)
        assert np.abs(s.m_x - o.x).max() < tol(g, f)


@pytest.mark.parametrize("name", ["nh_5400", "stvk_50700"])
def test_baseline_throughput_sizes(pkg, name):
    """BASELINE.json configs[1] / configs[2] at the sizes bench.py times them (Neo-Hookean bar 10x10x9 = 5 400 tets --
    the explicit-inverse solve --, StVK bar 13x13x50 = 50 700 tets -- panel sweeps; 20 iterations per frame) against the oracle and the
    compiled reference's fixtures (tests/golden/make_golden.py baseline_bars):
      * ONE ADMM iteration from a strongly deformed start (checkers.deformed_start: + and * only): the local step sees bit-identical input,
        so u, z, the warm start and the L-BFGS iteration counts are BIT-EXACT for every tet and anchor (oracle: all of them; reference: the
        recorded ones); x after the solve <= 1e-11, v <= 1e-9;
      * three frames from rest inside 20 x the reference's own 1-ulp sensitivity (the envelope every trajectory fixture uses)."""
    from checkers import deformed_start
    g = golden("traj_bar_%s.npz" % name)
    dims = tuple(int(v) for v in g["dims"]); kind = int(g["kind"]); sc = float(g["start_scale"])
    s, o = _bar_pair(pkg, kind, dims, 1)
    x0 = deformed_start(s.m_x * sc) / sc
    s.m_x = x0; o.x = x0
    s.step(1); o.step()
    nt = 6 * dims[0] * dims[1] * dims[2]
    tets = np.arange(0, nt, int(g["tet_stride"]))
    lt, la = s.read_local(0), s.read_local(1)
    ou = o.u; oz = o.z
    assert np.array_equal(lt["u"], ou[:9 * nt].reshape(nt, 9)) and np.array_equal(lt["z"], oz[:9 * nt].reshape(nt, 9))          # every tet, vs the oracle
    assert np.array_equal(la["u"], ou[9 * nt:].reshape(-1, 3)) and np.array_equal(la["z"], oz[9 * nt:].reshape(-1, 3))
    assert np.array_equal(lt["u"][tets], g["u_tets"]) and np.array_equal(lt["z"][tets], g["z_tets"])                              # the recorded ones, vs the compiled reference
    assert np.array_equal(lt["state"][tets], g["state_tets"]) and np.array_equal(lt["n_iters"][tets], g["n_iters_tets"])
    assert np.array_equal(la["u"], g["u_anchors"]) and np.array_equal(la["z"], g["z_anchors"])
    assert np.abs(s.m_x - g["x_one_iter"]).max() < 1e-11 and np.abs(s.m_x - o.x).max() < 1e-11
    assert np.abs(s.m_v - g["v_one_iter"]).max() < 1e-9
    s, o = _bar_pair(pkg, kind, dims, int(g["iters"]))
    for f in range(3):
        s.step(int(g["iters"])); o.step()
        assert np.abs(s.m_x - g["x_frames"][f]).max() < tol(g, f), (f, np.abs(s.m_x - g["x_frames"][f]).max())
        assert np.abs(s.m_x - o.x).max() < tol(g, f), f


def test_known_answers(pkg):
    g = golden("known_answers.npz")
    s = pkg.System(device_id=0); s.set_timestep(1.0)
    x = np.zeros(12); x[1] = 1; x[8] = 1; x[9] = 1
    s.add_nodes(x, np.ones(12))
    s.add_forces(KIND["ANCHOR"], [0, 1, 2], [-1.0, 1.0])
    s.add_forces(KIND["TET_LINEAR"], [[0, 1, 2, 3]], [1.0])
    s.initialize()
    xx = s.m_x; xx[9] = 200.0; s.m_x = xx
    s.step(20)
    out = s.m_x
    assert "%.6g" % out[9] == "171.571"                      # what singletet.cpp prints
    assert abs(out[9] - 171.57142857142716) < 1e-9
    assert np.abs(out - g["singletet_x"]).max() < 1e-9
    s = pkg.System(device_id=0); s.set_timestep(1.0)
    s.add_nodes(np.zeros(3), np.ones(3))
    s.add_gravity([0.0, float(np.float32(-9.8)), 0.0])
    s.initialize()
    for f in range(4):
        s.step(20)
        assert np.abs(s.m_x - g["singlenode_x"][f]).max() < 1e-12


def test_mesh_fixtures(pkg):
    g = golden("traj_dillo_nh.npz")
    n = g["x"].shape[0]
    s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
    s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
    s.add_forces(KIND["TET_NH"], g["tets"], [float(g["mu"]), float(g["lam"]), int(g["max_iter"])])
    s.add_forces(KIND["ANCHOR"], g["anchors"], [-1.0, 1.0])
    s.add_gravity([0, -9.8, 0])
    s.initialize()
    for f in range(g["x_frames"].shape[0]):
        s.step(int(g["iters"]))
        assert np.abs(s.m_x - g["x_frames"][f]).max() < tol(g, f)
    g = golden("traj_bunny_stvk.npz")
    n = g["x"].shape[0]
    s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
    s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
    s.add_forces(KIND["TET_STVK"], g["tets"], [float(g["mu"]), float(g["lam"]), int(g["max_iter"])])
    s.initialize()
    s.m_x = s.m_x * float(g["scale"])
    for f in range(g["x_frames"].shape[0]):
        s.step(int(g["iters"]))
        assert np.abs(s.m_x - g["x_frames"][f]).max() < tol(g, f)


def test_cloth_fixture(pkg):
    g = golden("traj_cloth.npz")
    n = g["x"].shape[0]
    s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
    s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
    s.add_forces(KIND["TRI_STRAIN"], g["tris"], [float(g["k_tri"]), g["lim"][0], g["lim"][1], 1.0])
    s.add_forces(KIND["BEND"], g["hinges"], [float(g["k_bend"])])
    s.add_forces(KIND["ANCHOR"], g["anchors"], [-1.0, 1.0])
    s.add_gravity([0, -9.8, 0])
    s.initialize()
    for f in range(g["x_frames"].shape[0]):
        s.step(int(g["iters"]))
        assert np.abs(s.m_x - g["x_frames"][f]).max() < 1e-9   # continuous algorithm: tight


@pytest.mark.parametrize("name", ["triarea", "fung"])
def test_skin_fixtures(pkg, name):
    """TriArea (+ bend) and FungTriangle membranes against the compiled reference's trajectories.
    TriArea has no transcendental: tight.  Fung evaluates exp() inside a truncated L-BFGS, which the
    reference itself amplifies (fixture ulp_sensitivity): 1e-9 or 1e4 x that envelope."""
    g = golden("traj_skin_%s.npz" % name)
    n = g["x"].shape[0]
    s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
    s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
    s.add_forces(int(g["kind"]), g["tris"], g["params"])
    if bool(g["with_bend"]):
        s.add_forces(KIND["BEND"], g["hinges"], [float(g["k_bend"])])
    s.add_forces(KIND["ANCHOR"], g["anchors"], [-1.0, 1.0])
    s.add_gravity([0, -9.8, 0])
    s.initialize()
    for f in range(g["x_frames"].shape[0]):
        s.step(int(g["iters"]))
        bound = 1e-9 if name == "triarea" else max(1e-9, 1e4 * float(g["ulp_sensitivity"][f]))
        assert np.abs(s.m_x - g["x_frames"][f]).max() < bound, f


def test_residuals_and_early_exit(pkg):
    """Opt-in residual tracking (admm_hip_enable_residuals / set_tolerance) against the oracle's restatement of
    the comment at System.cpp:64-65: per-iteration |r|, |s| of a cloth frame and of a StVK bar frame, the
    untouched trajectory, and the iteration at which a tolerance ends the ADMM loop."""
    g = golden("traj_cloth.npz")
    n = g["x"].shape[0]

    def build(S):
        s = S
        s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
        s.add_forces(KIND["TRI_STRAIN"], g["tris"], [float(g["k_tri"]), g["lim"][0], g["lim"][1], 1.0])
        s.add_forces(KIND["BEND"], g["hinges"], [float(g["k_bend"])])
        s.add_forces(KIND["ANCHOR"], g["anchors"], [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        return s
    s = build(pkg.System(device_id=0)); s.set_timestep(float(g["dt"])); s.initialize(); s.enable_residuals(True)
    o = build(Oracle()); o.settings(float(g["dt"]), 30); assert o.initialize(); o.track_residuals(True)
    for f in range(2):
        s.step(30); o.step()
        r, sd, it = s.residuals(); ro, so, ito = o.residuals()
        assert it == ito == 30
        assert np.abs(r - ro).max() < 1e-7 * ro.max() and np.abs(sd - so).max() < 1e-7 * so.max()
        assert np.abs(s.m_x - g["x_frames"][f]).max() < 1e-9
    # early exit at the same iteration as the oracle
    s2 = build(pkg.System(device_id=0)); s2.set_timestep(float(g["dt"])); s2.initialize()
    o2 = build(Oracle()); o2.settings(float(g["dt"]), 30); assert o2.initialize()
    tr, ts = float(np.sqrt(ro[9] * ro[10])), float(so.max() * 2)     # between two consecutive oracle values
    s2.set_tolerance(tr, ts, 1); o2.track_residuals(True, tr, ts)
    s2.step(30); o2.step()
    assert s2.residuals()[2] == o2.residuals()[2] < 30
    assert np.abs(s2.m_x - o2.x).max() < 1e-9
    s2.set_tolerance(tr, ts, 4)                                       # tested every 4th iteration only
    s2.step(30)
    assert s2.residuals()[2] % 4 == 0
    # hyperelastic bar (u, z of 9 rows, anchors after tets)
    sb, ob = _bar_pair(pkg, KIND["TET_STVK"], (3, 3, 8), 12)
    sb.enable_residuals(True); ob.track_residuals(True)
    sb.step(12); ob.step()
    r, sd, it = sb.residuals(); ro, so, _ = ob.residuals()
    # the truncated L-BFGS amplifies the rounding differences of the two solvers within the frame
    assert it == 12 and np.abs(r - ro).max() < 1e-4 * ro.max() and np.abs(sd - so).max() < 1e-4 * so.max()


def test_sparse_and_dense_solve_paths_agree(pkg, monkeypatch):
    """The same scene through the panel sweeps and through the explicit inverse: the solves agree to rounding
    (different summation orders), cloth trajectories (no truncated minimiser) to 1e-9."""
    g = golden("traj_cloth.npz")
    n = g["x"].shape[0]
    xs, sol = [], []
    b = np.random.default_rng(3).normal(size=3 * n)
    for dm in ("0", "4096"):
        monkeypatch.setenv("ADMM_HIP_DENSE_MAX", dm)
        s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
        s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
        s.add_forces(KIND["TRI_STRAIN"], g["tris"], [float(g["k_tri"]), g["lim"][0], g["lim"][1], 1.0])
        s.add_forces(KIND["BEND"], g["hinges"], [float(g["k_bend"])])
        s.add_forces(KIND["ANCHOR"], g["anchors"], [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        s.initialize()
        assert s.info()["dense_solve"] == (dm != "0")
        sol.append(s.solve_only(b))
        for f in range(3):
            s.step(int(g["iters"]))
            assert np.abs(s.m_x - g["x_frames"][f]).max() < 1e-9
        xs.append(s.m_x.copy())
    assert np.abs(sol[0] - sol[1]).max() < 1e-11 * np.abs(sol[0]).max()
    assert np.abs(xs[0] - xs[1]).max() < 1e-10


def test_graph_replay_matches_eager_launches(pkg, monkeypatch):
    """admm_hip_step replays one captured ADMM iteration per iteration (HIP graph); same kernels, same arguments:
    bitwise equal to launching them one by one -- over several frames, with a moving anchor retargeted between frames
    (host-mutable parameters live in device buffers the captured kernels read) and after recompute_weights (graph rebuilt)."""
    out = []
    for g in ("0", "1"):
        monkeypatch.setenv("ADMM_HIP_GRAPH", g)
        mg = pkg.meshgen
        x, t = mg.bar(3, 3, 10)
        m = mg.lumped_tet_mass(x, t, 1000.0)
        s = pkg.System(device_id=0); s.set_timestep(0.04)
        s.add_nodes(x.ravel(), np.repeat(m, 3))
        s.add_forces(KIND["TET_STVK"], t, [1e5, 1e5, 5])
        s.add_forces(KIND["ANCHOR"], mg.bar_anchor_nodes(3, 3), [-1.0, 1.0])
        tip = x.shape[0] - 1
        b = s.add_forces(KIND["ANCHOR"], [tip], [-1.0, 1.0], targets=x[tip][None, :])   # targets given = MovingAnchor
        s.add_gravity([0, -9.8, 0])
        s.initialize()
        for f in range(4):
            s.update_anchors(b, targets=(x[tip] + [0.02 * f, 0.0, 0.0])[None, :], active=[1 if f < 3 else 0])
            s.step(10)
        w = s.read_rest(0)["weight"] * 1.5
        s.set_weights(0, w); s.recompute_weights()
        s.step(10)
        out.append(s.m_x.copy())
    assert np.array_equal(out[0], out[1])


def test_frame_graph_matches_per_iteration_graph(pkg, monkeypatch):
    """Below 100k nodes the whole ADMM loop of a frame is ONE captured graph (one launch per frame); same kernels in the same order as
    one graph launch per iteration and as eager launches: bitwise equal, also when the iteration count changes between calls (the
    frame graph is captured again) and on a scene above the dense-solve limit (panel sweeps inside the graph)."""
    for dims, dense_max in (((3, 3, 10), None), ((6, 6, 30), "0")):
        out = []
        for mode in ("frame", "iteration", "eager"):
            monkeypatch.setenv("ADMM_HIP_FRAME_GRAPH", "1" if mode == "frame" else "0")
            monkeypatch.setenv("ADMM_HIP_GRAPH", "0" if mode == "eager" else "1")
            if dense_max is not None: monkeypatch.setenv("ADMM_HIP_DENSE_MAX", dense_max)
            s = pkg.make_bar_system(*dims, kind=KIND["TET_NH"])
            s.initialize()
            for iters in (20, 20, 7, 1, 20):
                s.step(iters)
            out.append(s.m_x.copy())
            del s
        assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
        monkeypatch.delenv("ADMM_HIP_DENSE_MAX", raising=False)


def test_one_launch_local_step_matches_one_launch_per_batch(pkg, monkeypatch):
    """Scenes with several batches run their whole local step as ONE launch (project_multi_kernel: the batches' blocks back to back);
    same per-element arithmetic, own outputs per element: bitwise equal to one launch per batch --
    the small mixed scene (NH + StVK tets, cloth triangles, hinges, anchors: five kinds), eager and graph-replayed."""
    out = []
    for env in ({"ADMM_HIP_LOCAL_MULTI": "1"}, {"ADMM_HIP_LOCAL_MULTI": "0"}, {"ADMM_HIP_LOCAL_MULTI": "1", "ADMM_HIP_GRAPH": "0"}, {"ADMM_HIP_LOCAL_MULTI": "0", "ADMM_HIP_GRAPH": "0"}):
        for k in ("ADMM_HIP_LOCAL_MULTI", "ADMM_HIP_GRAPH"): monkeypatch.delenv(k, raising=False)
        for k, v in env.items(): monkeypatch.setenv(k, v)
        s, _ = pkg.make_mixed_system(4, 4, 12, 12, 12)
        s.initialize()
        for _ in range(3): s.step(10)
        out.append((s.m_x.copy(), [s.read_local(b)["u"].copy() for b in range(len(s.batches))]))
        del s
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0])
        for a, b in zip(out[0][1], o[1]): assert np.array_equal(a, b)


def test_edge_cases(pkg):
    # empty batches, a single element, moving anchors (active and released)
    s = pkg.System(device_id=0); s.set_timestep(0.04)
    x = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.], [2, 2, 2]])
    s.add_nodes(x.ravel(), np.ones(15))
    s.add_forces(KIND["TET_NH"], np.zeros((0, 4), np.int32), np.zeros((0, 3)))
    s.add_forces(KIND["TET_STVK"], [[0, 1, 2, 3]], [100.0, 100.0, 5])
    b = s.add_forces(KIND["ANCHOR"], [4, 0], [[-1.0, 1.0], [-1.0, 0.0]], targets=[[2, 2, 3.0], [9, 9, 9.0]])
    s.add_gravity([0, -9.8, 0])
    s.initialize()
    o = Oracle(); o.settings(0.04, 5)
    o.add_nodes(x.ravel(), np.ones(15))
    o.add_forces(KIND["TET_STVK"], [[0, 1, 2, 3]], [100.0, 100.0, 5])
    h0 = o.add_moving_anchor(4, [2, 2, 3.0], True, -1.0)
    o.add_moving_anchor(0, [9, 9, 9.0], False, -1.0)
    o.add_gravity([0, -9.8, 0])
    assert o.initialize()
    for f in range(3):
        s.step(5); o.step()
        assert np.abs(s.m_x - o.x).max() < 1e-9
    # released anchor follows the node (AnchorForce.cpp:80-83 writes point->pos)
    tg = s.read_local(b)["state"]
    assert np.abs(tg[1] - np.array(list(o.force(2).pos))).max() < 1e-9
    # collapsed element ("collapsed to a point", TetForce.cpp:344-347): finite output
    s.m_x = np.zeros(15)
    s.step(5)
    assert np.isfinite(s.m_x).all()


def test_recompute_weights(pkg):
    """poordillo's H/F keys: anchor weight -> 0, System::recompute_weights (System.cpp:159-179)."""
    s, o = _bar_pair(pkg, KIND["TET_STVK"], (3, 3, 6), 10)
    s.step(10)
    w = s.read_rest(1)["weight"]; w[:] = 0.0
    s.set_weights(1, w)
    s.recompute_weights()
    x0 = s.m_x.reshape(-1, 3)[:16].copy()
    for _ in range(3):
        s.step(10)
    x1 = s.m_x.reshape(-1, 3)[:16]
    assert (x1[:, 1] < x0[:, 1] - 1e-3).all()       # released face now falls
    b = np.random.default_rng(0).normal(size=3 * s.n_nodes)
    assert np.abs(s.apply_A(s.solve_only(b)) - b).max() < 1e-10 * np.abs(b).max()


def test_full_size_properties(pkg):
    """BASELINE.json's full size (1,001,472 tets): size-independent properties."""
    s = pkg.make_bar_system(32, 32, 163)
    s.initialize()
    n = s.n_nodes
    assert s.n_tets == 1001472 and n == 178596
    rng = np.random.default_rng(0)
    b = rng.normal(size=3 * n)
    x = s.solve_only(b)
    assert np.abs(s.apply_A(x) - b).max() < 1e-10 * np.abs(b).max()          # the factor solves the assembled system
    # linearity of the solve
    b2 = rng.normal(size=3 * n)
    assert np.abs(s.solve_only(b + 2 * b2) - (x + 2 * s.solve_only(b2))).max() < 1e-9 * np.abs(x).max()
    x0 = s.m_x.copy()
    # rest state without gravity is a fixed point of the ADMM frame
    s.set_gravity(0, [0, 0, 0])
    s.step(3)
    assert np.abs(s.m_x - x0).max() < 1e-12
    s.set_gravity(0, [0, -9.8, 0])
    s.step(20); s.step(20)
    x2 = s.m_x
    assert np.isfinite(x2).all()
    # anchored face stays put (weight 1000 penalty: to ~1e-6), free end sags
    face = np.arange(33 * 33)
    assert np.abs(x2.reshape(-1, 3)[face] - x0.reshape(-1, 3)[face]).max() < 1e-4
    assert x2.reshape(-1, 3)[-1, 1] < x0.reshape(-1, 3)[-1, 1] - 1e-3
    # run-to-run determinism at full size: rewind and replay bit for bit
    u0 = s.read_local(0)
    s.m_x = x0; s.m_v = np.zeros_like(x0)
    s.write_local(0, u=np.zeros_like(u0["u"]), state=np.ones_like(u0["state"]))
    s.write_local(1, u=np.zeros((33 * 33, 3)))
    s.step(20); xa = s.m_x
    s.m_x = x0; s.m_v = np.zeros_like(x0)
    s.write_local(0, u=np.zeros_like(u0["u"]), state=np.ones_like(u0["state"]))
    s.write_local(1, u=np.zeros((33 * 33, 3)))
    s.step(20); xb = s.m_x
    assert np.array_equal(xa, xb)


def test_full_size_bar_vs_compiled_reference(pkg):
    """The benchmark workload itself against the COMPILED REFERENCE: tests/golden/traj_bar_1M.npz holds the reference's
    positions (every 8th node + sums over all nodes) after one frame = 20 ADMM iterations of the 1,001,472-tet Neo-Hookean
    bar, and its own sensitivity to a 1-ulp perturbation of the start (make_golden_fullsize.py: about an hour of reference
    CPU time, of which 23 min are its initialize).  The local steps are bit-identical, the solves differ by rounding; the
    bound is 20 x the reference's sensitivity like for every other trajectory fixture."""
    import os
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj_bar_1M.npz")):
        pytest.skip("full-size fixture not generated")
    g = golden("traj_bar_1M.npz")
    s = pkg.make_bar_system(*[int(v) for v in g["dims"]])
    s.initialize()
    assert s.n_nodes == int(g["n_nodes"])
    s.step(int(g["iters"]))
    x = s.m_x.reshape(-1, 3)
    bound = max(1e-9, 20.0 * float(g["ulp_sensitivity"]))
    err = np.abs(x[::int(g["stride"])] - g["x_sample"]).max()
    assert err < bound, (err, bound)
    assert abs(np.abs(x).sum() - float(g["sum_abs"])) < bound * x.size
    assert abs((x * x).sum() - float(g["sum_sq"])) < bound * x.size


def test_full_size_bar_one_iteration_vs_compiled_reference(pkg):
    """The TIGHT check at the headline size (tests/golden/traj_bar_1M_one_iter.npz, make_golden_fullsize.py one_iter: the compiled reference
    on the 1,001,472-tet Neo-Hookean bar, 1 064 s of initialize).  (1) Its solver.solve(b) for three right-hand sides, at every 8th node:
    <= 1e-10 x max|x| -- the bound of the small solve fixtures, here on the benchmark's own factor.  (2) ONE ADMM iteration from a strongly
    deformed start built from + and * only (checkers.deformed_start: the same bits on every host): the local step of all 1 001 472 tets
    sees bit-identical input, so u, z, the warm start and the L-BFGS iteration counts of the recorded tets (every 64th) and of all 1 089
    anchors are BIT-EXACT; the sums of |u|, u^2, |z|, z^2 over ALL tets agree to summation-order rounding; x after the one solve <= 1e-10,
    v = (x - x_old) / dt <= 1e-8 -- no 20-iteration chaos in between (the 20-iteration frame: test_full_size_bar_vs_compiled_reference)."""
    import os
    from checkers import deformed_start, solve_rhs
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj_bar_1M_one_iter.npz")):
        pytest.skip("full-size fixture not generated")
    g = golden("traj_bar_1M_one_iter.npz")
    dims = [int(v) for v in g["dims"]]
    s = pkg.make_bar_system(*dims)
    s.initialize()
    assert s.n_nodes == int(g["n_nodes"])
    st = int(g["stride"])
    x0 = s.m_x.copy()
    mx, mt = pkg.meshgen.bar(*dims)
    B = solve_rhs(int(g["solve_seed"]), x0, np.repeat(pkg.meshgen.lumped_tet_mass(mx, mt, 1000.0), 3))      # (m_masses as make_golden.bar_system set them)
    for k in range(3):
        xs = s.solve_only(B[k]).reshape(-1, 3)[::st]
        assert np.abs(xs - g["solve_x_sample"][k]).max() < 1e-10 * float(g["solve_x_max"][k]), (k, np.abs(xs - g["solve_x_sample"][k]).max())
    s.m_x = deformed_start(x0)
    s.step(1)
    nt = 6 * dims[0] * dims[1] * dims[2]
    tets = np.arange(0, nt, int(g["tet_stride"]))
    lt, la = s.read_local(0), s.read_local(1)
    assert np.array_equal(lt["u"][tets], g["u_tets"]) and np.array_equal(lt["z"][tets], g["z_tets"])
    assert np.array_equal(lt["state"][tets], g["state_tets"]) and np.array_equal(lt["n_iters"][tets], g["n_iters_tets"])
    assert np.array_equal(la["u"], g["u_anchors"]) and np.array_equal(la["z"], g["z_anchors"])
    for arr, key in ((np.abs(lt["u"]).sum(), "u_sum_abs"), ((lt["u"] ** 2).sum(), "u_sum_sq"), (np.abs(lt["z"]).sum(), "z_sum_abs"), ((lt["z"] ** 2).sum(), "z_sum_sq")):
        assert abs(arr - float(g[key])) <= 1e-11 * abs(float(g[key])), key
    x = s.m_x.reshape(-1, 3)
    err = np.abs(x[::st] - g["x_sample"]).max()
    assert err < 1e-10, err
    assert np.abs(s.m_v.reshape(-1, 3)[::st] - g["v_sample"]).max() < 1e-8
    assert abs(np.abs(x).sum() - float(g["sum_abs"])) < 1e-10 * x.size


def test_full_size_mixed_vs_compiled_reference(pkg):
    """BASELINE.json configs[4] at full size (498,888 NH + StVK tets, 99,856 cloth triangles, 149k hinges, anchors) against the
    compiled reference after one frame of 20 iterations (tests/golden/traj_mixed_full.npz, make_golden_fullsize.py mixed):
    every force kernel of the scene at scale, bounded by 20 x the reference's own 1-ulp sensitivity."""
    import os
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj_mixed_full.npz")):
        pytest.skip("full-size fixture not generated")
    g = golden("traj_mixed_full.npz")
    s = pkg.make_mixed_system(*[int(v) for v in g["bar_dims"]], *[int(v) for v in g["cloth"]])[0]
    s.initialize()
    assert s.n_nodes == int(g["n_nodes"])
    s.step(int(g["iters"]))
    x = s.m_x.reshape(-1, 3)
    bound = max(1e-9, 20.0 * float(g["ulp_sensitivity"]))
    err = np.abs(x[::int(g["stride"])] - g["x_sample"]).max()
    assert err < bound, (err, bound)
    assert abs(np.abs(x).sum() - float(g["sum_abs"])) < bound * x.size


def test_full_size_mixed_scene_properties(pkg):
    """BASELINE.json configs[4] at full size (498,888 NH+StVK tets, 99,856 triangles, 149k hinges, anchors; two
    disconnected bodies in one factorization): every force kernel in one step -- size-independent properties."""
    s, d = pkg.make_mixed_system(26, 26, 123, 158, 158)
    s.initialize()
    info = s.info()
    assert info["n_elems_total"] == 498888 + 99856 + d["forces"][3][1].shape[0] + d["forces"][4][1].shape[0]
    n = s.n_nodes
    rng = np.random.default_rng(1)
    b = rng.normal(size=3 * n)
    x = s.solve_only(b)
    assert np.abs(s.apply_A(x) - b).max() < 1e-10 * np.abs(b).max()
    x0 = s.m_x.copy()
    s.set_gravity(0, [0, 0, 0])
    s.step(2)
    assert np.abs(s.m_x - x0).max() < 1e-8                     # rest state is a fixed point for all five kinds (cloth nodes weigh 1e-5 kg: rounding is amplified)
    s.set_gravity(0, [0, -9.8, 0])
    s.enable_residuals(True)
    s.step(20)
    r, sd, it = s.residuals()
    assert it == 20 and np.isfinite(r).all() and np.isfinite(sd).all() and r[-1] < r[1]
    s.enable_residuals(False)
    s.step(20)
    x2 = s.m_x.reshape(-1, 3)
    assert np.isfinite(x2).all()
    off = d["X"].shape[0] - (159 * 159 + 158 * 158)
    assert np.abs(x2[off] - d["X"][off]).max() < 1e-4          # cloth corner anchor holds
    assert x2[off + 159 * 80 + 80, 1] < d["X"][off + 159 * 80 + 80, 1] - 1e-3   # cloth interior falls


def test_collision_fixture(pkg):
    """plinkopony-like scene against the compiled reference's trajectory (continuous algorithm: tight)."""
    g = golden("traj_collision.npz")
    n = g["x"].shape[0]
    s = pkg.System(device_id=0); s.set_timestep(float(g["dt"]))
    s.add_nodes(g["x"].ravel(), np.repeat(g["mass"], 3))
    s.add_forces(KIND["TET_LINEAR"], g["tets"], [float(g["k"])])
    b = s.add_forces(KIND["COLLISION"], np.arange(n), [float(g["weight"])])
    s.set_collision_shapes(g["types"], g["params"])
    s.add_gravity([0, -9.8, 0])
    s.initialize()
    fi = 0
    for f in range(int(g["frames"][-1]) + 1):
        s.step(int(g["iters"]))
        if f == g["frames"][fi]:
            assert np.abs(s.m_x - g["x_frames"][fi]).max() < max(tol(g, fi), 1e-9), f
            fi += 1
    # the collision projection itself, bit-exact against the oracle on random points
    rng = np.random.default_rng(3)
    o = Oracle(); o.settings(0.02, 1)
    o.add_nodes(g["x"].ravel(), np.repeat(g["mass"], 3))
    o.add_forces(KIND["COLLISION"], np.arange(n), [32.0]); o.set_collision_shapes(g["types"], g["params"])
    assert o.initialize()
    s2 = pkg.System(device_id=0); s2.set_timestep(0.02)
    s2.add_nodes(g["x"].ravel(), np.repeat(g["mass"], 3))
    s2.add_forces(KIND["COLLISION"], np.arange(n), [32.0]); s2.set_collision_shapes(g["types"], g["params"])
    s2.initialize()
    for it in range(3):
        xc = rng.uniform(-0.4, 0.9, size=3 * n)
        s2.local_step_only(xc)
        u, z = oracle_local_step(o, xc, n, 3)
        out = s2.read_local(0)
        assert np.array_equal(out["z"], z) and np.array_equal(out["u"], u)


def test_explicit_forces(pkg):
    """ExplicitForce on an index subset (bit-exact frame) and WindForce (two-pass form) on a cloth."""
    mg = pkg.meshgen
    x, tris = mg.sym_plane(6, 4, size=1.0)
    hinges = mg.bend_hinges(tris)
    n = x.shape[0]
    sub = np.arange(1, n, 2, dtype=np.int32)

    def build(S):
        s = S
        s.add_nodes(x.ravel(), np.full(3 * n, 0.5 / n))
        s.add_forces(KIND["TRI_STRAIN"], tris, [100.0, 0.95, 1.05, 1.0])
        s.add_forces(KIND["BEND"], hinges, [20.0])
        s.add_forces(KIND["ANCHOR"], [0, 6], [-1.0, 1.0])
        s.add_gravity([0, -9.8, 0])
        return s
    s = build(pkg.System(device_id=0)); s.set_timestep(0.04)
    o = build(Oracle()); o.settings(0.04, 10)
    s.add_explicit(pkg.EXPLICIT["CONST"], [0.5, 0.0, 0.2], sub); o.add_explicit(0, [0.5, 0.0, 0.2], sub)
    s.initialize(); assert o.initialize()
    for f in range(4):
        s.step(10); o.step()
        assert np.abs(s.m_x - o.x).max() < 1e-9
    # wind: the reference's loop in serial order (each triangle sees the increments of the ones before it)
    s = build(pkg.System(device_id=0)); s.set_timestep(0.04)
    w = s.add_explicit(pkg.EXPLICIT["WIND"], [10.0, 0.0, 2.0], tris)
    s.initialize()
    s.step(0)                                    # explicit forces only (admm_iters = 0): x += dt v
    v = s.m_v.reshape(-1, 3)
    vexp = np.tile(0.04 * np.array([0, -9.8, 0]), (n, 1))
    wind = np.array([10.0, 0, 2.0])
    for t in tris:
        a = x[t[1]] - x[t[0]]; b = x[t[2]] - x[t[0]]
        nrm = np.cross(a, b); nn = np.sqrt(nrm[0] * nrm[0] + (nrm[1] * nrm[1] + nrm[2] * nrm[2]))
        nh = nrm / nn
        vr = (vexp[t[0]] + vexp[t[1]] + vexp[t[2]]) / 3.0 - wind
        vn = nh[0] * vr[0] + (nh[1] * vr[1] + nh[2] * vr[2])
        force = (-1000.0 * (0.5 * nn) * vn * abs(vn)) * nh * 0.33 * 0.04
        for c in range(3):
            vexp[t[c]] += force
    assert np.abs(v - vexp).max() < 1e-13 * max(1.0, np.abs(vexp).max())
    # ... and directly against the ORACLE's wind path (orc wind_project, the restatement of ExplicitForce.cpp:42-98 in serial
    # triangle order that tests/test_oracle_vs_ref.py pins to the compiled reference): velocities after the explicit forces of a
    # frame, then whole frames with the wind blowing while the cloth moves (the wind reads the moving x and v every frame)
    ow = build(Oracle()); ow.settings(0.04, 0)
    ow.add_explicit(1, [10.0, 0.0, 2.0], tris)
    assert ow.initialize()
    ow.step()
    assert np.abs(s.m_v - ow.v).max() < 1e-13 * max(1.0, np.abs(ow.v).max())
    assert np.abs(s.m_x - ow.x).max() < 1e-13
    s3 = build(pkg.System(device_id=0)); s3.set_timestep(0.04)
    s3.add_explicit(pkg.EXPLICIT["WIND"], [10.0, 0.0, 2.0], tris)
    s3.initialize()
    o3 = build(Oracle()); o3.settings(0.04, 10)
    o3.add_explicit(1, [10.0, 0.0, 2.0], tris)
    assert o3.initialize()
    for f in range(4):
        s3.step(10); o3.step()
        assert np.abs(s3.m_x - o3.x).max() < 1e-9 and np.abs(s3.m_v - o3.v).max() < 1e-8, f
    s.set_gravity(w, [0.0, 0.0, 0.0])            # direction is host-mutable (windyflag.cpp:141-152)
    s.step(5)
    assert np.isfinite(s.m_x).all()


def test_mixed_scene(pkg):
    """config 5 in miniature: NH + StVK tets, cloth (triangle strain + bend) and anchors in one step;
    two disconnected bodies in one factorization."""
    s, d = pkg.make_mixed_system(4, 3, 9, 8, 6)
    s.initialize()
    o = Oracle(); o.settings(0.04, 1)
    o.add_nodes(d["X"].ravel(), np.repeat(d["M"], 3))
    for name, idx, par in d["forces"]:
        o.add_forces(KIND[name], idx, par)
    o.add_gravity([0, -9.8, 0])
    assert o.initialize()
    assert s.info()["rows_compact"] == o.rows
    b = np.random.default_rng(0).normal(size=3 * s.n_nodes)
    assert np.abs(s.apply_A(s.solve_only(b)) - b).max() < 1e-10 * np.abs(b).max()
    s.step(1); o.step()
    assert np.abs(s.m_x - o.x).max() < 1e-10
    for _ in range(3):
        s.step(20)
    assert np.isfinite(s.m_x).all()


def test_device_log_exp_bitwise(pkg, hm):
    """admm_log / admm_exp as the GPU executes them (fused multiply-adds, table loads, subnormal handling on the device)
    against this host's libm, 2M arguments each: every bit.  (tests/test_host_math.py checks the same header on the CPU.)"""
    import ctypes as C
    s = pkg.make_bar_system(2, 2, 3)
    s.initialize()
    rng = np.random.default_rng(17)
    specials = np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, -1.0, 5e-324, 2.2250738585072014e-308, 1e-310, 1.7976931348623157e308,
                         1 - 2.0 ** -4, 1 + float.fromhex("0x1.09p-4"), 709.782712893384, -745.1332191019411, -708.3964185322641, 512.0, -512.0, 1024.0, -1024.0])
    xl = np.concatenate([rng.uniform(0.5, 2.0, 500_000), 1.0 + rng.normal(size=500_000) * 10.0 ** rng.uniform(-12, -0.5, 500_000),
                         np.exp(rng.uniform(-740, 709, 500_000)), rng.uniform(0.93, 1.07, 400_000),
                         np.frombuffer(rng.integers(0, 2 ** 63, 100_000, dtype=np.int64).tobytes(), np.float64), specials])
    xe = np.concatenate([rng.uniform(-5, 5, 500_000), rng.uniform(-746, 710, 500_000), rng.uniform(-1100, 1100, 300_000), rng.uniform(-745.2, -707.0, 300_000),
                         rng.uniform(700, 709.8, 200_000), rng.normal(size=100_000) * 10.0 ** rng.uniform(-20, 0, 100_000),
                         np.frombuffer(rng.integers(-2 ** 63, 2 ** 63, 100_000, dtype=np.int64).tobytes(), np.float64), specials])
    for op, x, ref_fn in ((0, xl, hm.hm_libm_log), (1, xe, hm.hm_libm_exp)):
        x = np.ascontiguousarray(x)
        ref = np.zeros_like(x)
        ref_fn(C.c_int(x.size), x.ctypes.data_as(C.POINTER(C.c_double)), ref.ctypes.data_as(C.POINTER(C.c_double)))
        got = s.debug_math(op, x)
        ok = ~np.isnan(ref)
        bad = np.nonzero(got.view(np.int64)[ok] != ref.view(np.int64)[ok])[0]
        assert bad.size == 0, (op, bad.size, x[ok][bad[:5]], got[ok][bad[:5]], ref[ok][bad[:5]])
        assert np.all(np.isnan(got[~ok]))


def test_rhs_slot_layouts_agree_bitwise(pkg, monkeypatch):
    """The RHS slots are laid out rank-major (default) or node-sorted (fallback for meshes with a few very high-valence nodes,
    ADMM_HIP_SLOTS_NODE_SORTED=1): same per-node summation order, so the trajectories must be bitwise identical -- with the
    panel sweeps and with the small-system inverse, residual tracking included (it shares the layout)."""
    out = []
    for env in (None, "1"):
        if env is None:
            monkeypatch.delenv("ADMM_HIP_SLOTS_NODE_SORTED", raising=False)
        else:
            monkeypatch.setenv("ADMM_HIP_SLOTS_NODE_SORTED", env)
        s = pkg.make_mixed_system(5, 4, 11, 9, 7)[0]
        s.initialize()
        s.enable_residuals(True)
        xs = []
        for _ in range(3):
            s.step(8); xs.append(s.m_x.copy())
        r, d, _n = s.residuals()
        out.append((xs, np.array(r), np.array(d)))
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_irregular_delaunay_mesh(pkg, monkeypatch):
    """An unstructured mesh: Delaunay tets of random points in a box (irregular valence: 5-60 tets per node, badly shaped
    elements filtered at 1e-3 of the mean volume).  Assembly, ordering, both slot layouts and the sweeps on a topology
    that is nothing like the structured bar: one ADMM iteration against the oracle, solve residual, a few frames stay finite."""
    from scipy.spatial import Delaunay
    monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    rng = np.random.default_rng(42)
    pts = rng.uniform(0, 1, size=(2500, 3)) * np.array([1.0, 1.0, 3.0])
    tets = Delaunay(pts).simplices.astype(np.int32)
    v = np.einsum("ij,ij->i", pts[tets[:, 1]] - pts[tets[:, 0]], np.cross(pts[tets[:, 2]] - pts[tets[:, 0]], pts[tets[:, 3]] - pts[tets[:, 0]])) / 6.0
    tets = tets[np.abs(v) > 1e-3 * np.abs(v).mean()]
    used = np.unique(tets)
    assert used.size == pts.shape[0]
    m = np.zeros(pts.shape[0])
    np.add.at(m, tets.ravel(), np.repeat(np.abs(v[np.abs(v) > 1e-3 * np.abs(v).mean()]) * 1000.0 / 4.0, 4))
    anchors = np.nonzero(pts[:, 2] < 0.15)[0].astype(np.int32)
    deg = np.bincount(tets.ravel(), minlength=pts.shape[0])
    assert deg.max() > 2.5 * deg.mean()                      # genuinely irregular
    xs = []
    for layout in (None, "1"):
        if layout is None:
            monkeypatch.delenv("ADMM_HIP_SLOTS_NODE_SORTED", raising=False)
        else:
            monkeypatch.setenv("ADMM_HIP_SLOTS_NODE_SORTED", layout)
        s = pkg.System(device_id=0); s.set_timestep(0.02)
        s.add_nodes(pts.ravel(), np.repeat(m, 3))
        s.add_forces(KIND["TET_STVK"], tets, [5e4, 5e4, 5])
        s.add_forces(KIND["ANCHOR"], anchors, [-1.0, 1.0])
        s.add_gravity([0.0, -9.8, 0.0])
        s.initialize()
        if layout is None:
            o = Oracle(); o.settings(0.02, 1)
            o.add_nodes(pts.ravel(), np.repeat(m, 3))
            o.add_forces(KIND["TET_STVK"], tets, [5e4, 5e4, 5])
            o.add_forces(KIND["ANCHOR"], anchors, [-1.0, 1.0])
            o.add_gravity([0, -9.8, 0])
            assert o.initialize()
            b = rng.normal(size=3 * s.n_nodes)
            x = s.solve_only(b)
            assert np.abs(s.apply_A(x) - b).max() < 1e-9 * np.abs(b).max()
            s.step(1); o.step()
            assert np.abs(s.m_x - o.x).max() < 1e-9
        else:
            s.step(1)
        for _ in range(3):
            s.step(10)
        assert np.isfinite(s.m_x).all()
        xs.append(s.m_x.copy())
    assert np.array_equal(xs[0], xs[1])                       # both slot layouts, bit for bit


@pytest.mark.parametrize("direct", ["0", None])
def test_state_boundary_and_sampled_timing(pkg, monkeypatch, direct):
    """The class API's frame boundary (admm_hip_upload_state / download_state: one DMA per vector out of / into page-locked
    caller memory, reordering to the factor's node order on the device; systems of up to 12 288 nodes: NO DMA, the kernels address a
    page-locked [x | v] buffer directly -- `direct`: both paths) against get/set; and timing events around every k-th
    iteration only: same trajectory bit for bit, phase sums scaled to the frame and consistent with the frame's real span."""
    if direct is not None: monkeypatch.setenv("ADMM_HIP_STATE_DIRECT", direct)
    s = pkg.make_bar_system(6, 5, 14, kind=KIND["TET_STVK"])
    s.initialize()
    ref = pkg.make_bar_system(6, 5, 14, kind=KIND["TET_STVK"])
    ref.initialize()
    n3 = 3 * s.n_nodes
    hx = np.empty(n3); hv = np.empty(n3)
    s.pin_host(hx); s.pin_host(hv)
    rng = np.random.default_rng(4)
    x0 = s.m_x * (1.0 + 1e-3 * rng.normal(size=n3)); v0 = 1e-2 * rng.normal(size=n3)
    hx[:] = x0; hv[:] = v0
    s.upload_state(hx, hv)
    assert np.array_equal(s.m_x, x0) and np.array_equal(s.m_v, v0)          # upload == set
    ref.m_x = x0; ref.m_v = v0
    s.enable_timing(3)                                                        # events around every 3rd iteration
    for f in range(3):
        s.upload_state(hx, hv); s.step(10); s.download_state(hx, hv)
        ref.step(10)
        assert np.array_equal(hx, ref.m_x) and np.array_equal(hv, ref.m_v)  # download == get; sampled timing changes nothing
        t = s.timing()
        assert t["iters"] == 10 and t["total_ms"] > 0
        phases = t["local_ms"] + t["rhs_ms"] + t["allreduce_ms"] + t["solve_fwd_ms"] + t["solve_bwd_ms"]
        assert 0.3 * t["total_ms"] < phases < 2.0 * t["total_ms"]
    s.upload_state(hx, None)                                                  # one vector alone
    assert np.array_equal(s.m_v, ref.m_v)
    s.pin_host(hx, False); s.pin_host(hv, False)


def test_state_boundary_with_a_partly_registered_vector(pkg, monkeypatch):
    """admm_hip_pin_host takes any byte count: a vector page-locked over FEWER bytes than it holds (or whose registered span ends
    inside it) must take the DMA path -- the zero-copy kernels would run off the end of the mapping -- and still arrive intact."""
    monkeypatch.setenv("ADMM_HIP_STATE_DIRECT", "0")      # the large-system path (zero copy / DMA), not the small systems' staging buffer
    s = pkg.make_bar_system(6, 5, 14, kind=KIND["TET_STVK"]); s.initialize()
    n3 = 3 * s.n_nodes
    page = 4096 // 8
    buf = np.zeros(2 * n3 + 4 * page)
    off = (-buf.ctypes.data // 8) % page                      # page-aligned start inside buf
    hx = buf[off:off + n3]; hv = buf[off + n3 + page:off + 2 * n3 + page]
    rng = np.random.default_rng(9)
    hx[:] = s.m_x * (1.0 + 1e-3 * rng.normal(size=n3)); hv[:] = 1e-2 * rng.normal(size=n3)
    x0, v0 = hx.copy(), hv.copy()
    half = (n3 // 2 // page) * page * 8                       # whole pages, about half of the vector
    pin = lambda a, nbytes, on: s._chk(s.L.admm_hip_pin_host(s.h, a.ctypes.data, nbytes, on))
    pin(hx, half, 1); pin(hv, half, 1)
    try:      # (a registration left behind would poison later host-to-device copies of this process: HIP refuses spans that straddle one)
        s.upload_state(hx, hv)
        assert np.array_equal(s.m_x, x0) and np.array_equal(s.m_v, v0)
        hx[:] = 0.0; hv[:] = 0.0
        s.download_state(hx, hv)
        assert np.array_equal(hx, x0) and np.array_equal(hv, v0)
    finally:
        pin(hx, half, 0); pin(hv, half, 0)
    s.pin_host(hx); s.pin_host(hv)                            # the whole vectors: zero copy, same values
    try:
        s.upload_state(hx, hv); s.sync()                      # (asynchronous: the vectors are the kernel's until the next synchronising call)
        hx[:] = 0.0; hv[:] = 0.0
        s.download_state(hx, hv)
        assert np.array_equal(hx, x0) and np.array_equal(hv, v0)
    finally:
        s.pin_host(hx, False); s.pin_host(hv, False)


def test_timing_is_reset_by_enable_and_by_untimed_steps(pkg):
    """enable, step, step, disable, steps, enable, step: the step before the last one was NOT timed -- admm_hip_get_timing_previous
    must refuse instead of handing out the old stretch's events; and enable_timing itself forgets whatever was pending."""
    s = pkg.make_bar_system(4, 4, 8, kind=KIND["TET_STVK"]); s.initialize()
    s.enable_timing(1); s.step(4); s.step(4)
    s.enable_timing(0)
    for _ in range(3):
        s.step(4)
    s.enable_timing(1); s.step(4)
    with pytest.raises(pkg.AdmmHipError):
        s.timing_previous()
    assert s.timing()["iters"] == 4
    s.step(4); s.step(4)
    s.enable_timing(2)                                        # a new setting: nothing is pending
    with pytest.raises(pkg.AdmmHipError):
        s.timing_previous()


def test_timing_of_the_previous_step(pkg):
    """admm_hip_get_timing_previous: a timed step keeps its events until the step after the next is recorded, so a loop can read frame
    f - 1 after queueing frame f (what bench.py does: no idle GPU between frames).  Each frame is read exactly once, in order, the
    numbers are a frame's (iters, positive span, phases inside it), a second read and an untimed predecessor are refused, and the
    trajectory is the untimed one bit for bit."""
    s = pkg.make_bar_system(6, 5, 14, kind=KIND["TET_STVK"]); s.initialize()
    ref = pkg.make_bar_system(6, 5, 14, kind=KIND["TET_STVK"]); ref.initialize()
    s.step(10); ref.step(10)                      # an untimed step first
    s.enable_timing(2)
    s.step(10); ref.step(10)
    with pytest.raises(pkg.AdmmHipError):         # the step before the last one was not timed
        s.timing_previous()
    seen = []
    for f in range(3):
        s.step(7 + f); ref.step(7 + f)
        seen.append(s.timing_previous())          # the step before: 10, 7, 8 iterations
    seen.append(s.timing())                       # the last one: 9 iterations
    assert [t["iters"] for t in seen] == [10, 7, 8, 9]
    for t in seen:
        phases = t["local_ms"] + t["rhs_ms"] + t["allreduce_ms"] + t["solve_fwd_ms"] + t["solve_bwd_ms"]
        assert t["total_ms"] > 0 and 0.3 * t["total_ms"] < phases < 2.0 * t["total_ms"]
    with pytest.raises(pkg.AdmmHipError):         # read already
        s.timing_previous()
    assert np.array_equal(s.m_x, ref.m_x) and np.array_equal(s.m_v, ref.m_v)


def test_checkpoint_resume_is_bitwise(pkg):
    """The solver's state between frames is (m_x, m_v, u of every force, the hyperelastic warm start last_prox_result + init_hess):
    the reference cannot serialise it (its save() writes geometry only); through the C ABI a fresh context that
    receives this state continues the trajectory bit for bit."""
    def make():
        s, _ = pkg.make_mixed_system(4, 3, 10, 8, 6)
        s.initialize()
        return s
    a = make()
    for _ in range(3):
        a.step(10)
    ck = dict(x=a.m_x.copy(), v=a.m_v.copy(), loc=[a.read_local(b) for b in range(len(a.batches))])
    for _ in range(2):
        a.step(10)
    b = make()
    b.m_x = ck["x"]; b.m_v = ck["v"]
    for bi, loc in enumerate(ck["loc"]):
        kind = b.batches[bi][0]
        b.write_local(bi, u=loc["u"], state=loc["state"] if pkg.KIND_STATE[kind] else None)
    for _ in range(2):
        b.step(10)
    assert np.array_equal(a.m_x, b.m_x) and np.array_equal(a.m_v, b.m_v)


def test_create_destroy_cycles_leave_no_device_memory_behind(pkg):
    """admm_hip_destroy owns everything a context made (factor panels, batches, streams, events, captured graphs, pinned staging):
    60 create -> initialize -> step -> destroy cycles over three kinds of scene -- sparse sweeps, the dense small-system path, a mixed
    scene with residual tracking and timing events -- must leave the device's free memory where it was (one context of the bar below
    holds ~40 MB, so a leaked context shows)."""
    import gc
    import torch

    def cycle(i):
        if i % 3 == 0:
            s = pkg.make_bar_system(8, 8, 40)
        elif i % 3 == 1:
            s = pkg.make_bar_system(3, 3, 5, kind=pkg.KIND["TET_STVK"])
        else:
            s, _ = pkg.make_mixed_system(4, 3, 10, 8, 6)
            s.enable_residuals(True)
        s.initialize()
        if i % 2:
            s.enable_timing(1)
        s.step(5); s.step(5)
        if i % 2:
            s.timing()
        x = s.m_x.copy()
        s.__del__()
        return x

    for i in range(6):      # warm-up: library-level caches (module load, the runtime's pools) settle
        cycle(i)
    gc.collect(); torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    first = [cycle(i) for i in range(3)]
    for i in range(3, 60):
        x = cycle(i)
        if i >= 57:
            assert np.array_equal(x, first[i % 3])      # and the 20th context of a scene computes what the first did
    gc.collect(); torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, "device memory lost over 57 cycles: %.1f MB" % ((free0 - free1) / 2 ** 20)
