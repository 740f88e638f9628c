"""CPU: host side of admm_hip_finalize -- Force::initialize restatement (bit-exact
rest data / weights / global_idx), the assembled scalar system, the nested
dissection + multifrontal factor (validated through the panel form's two
sweeps evaluated on the host by the debug hook)."""
import numpy as np
import pytest

from checkers import KIND, Oracle
from conftest import golden


def test_rest_data_matches_oracle_bit_exact(pkg):
    mg = pkg.meshgen
    x, t = mg.bar(3, 3, 7)
    rng = np.random.default_rng(0)
    x = x + 0.01 * rng.normal(size=x.shape)       # generic (non axis-aligned) tets
    m = mg.lumped_tet_mass(x, t, 1000.0)
    xs, tris = mg.sym_plane(4, 3)
    hinges = mg.bend_hinges(tris)
    off = x.shape[0]
    X = np.concatenate([x, xs + np.array([0, 2.0, 0]) + 0.01 * rng.normal(size=xs.shape)])
    M = np.concatenate([m, np.full(xs.shape[0], 0.01)])
    s = pkg.System(device_id=-1); s.set_timestep(0.04)
    o = Oracle(); o.settings(0.04, 1)
    s.add_nodes(X.ravel(), np.repeat(M, 3)); o.add_nodes(X.ravel(), np.repeat(M, 3))
    batches = [("TET_NH", t, [1e5, 2e5, 5]), ("TET_STVK", t[:50], [100., 80., 5]), ("TET_LINEAR", t[50:90], [10.]), ("TET_VOLUME", t[90:120], [5., .9, 1.1]),
               ("TRI_STRAIN", tris + off, [100., .95, 1.05, 1.]), ("BEND", hinges + off, [20.]), ("SPRING", np.array([[0, 5], [3, 9]]), [50.]),
               ("ANCHOR", np.arange(16), [-1., 1.])]
    for name, idx, p in batches:
        s.add_forces(KIND[name], idx, p); o.add_forces(KIND[name], idx, p)
    s.initialize(); assert o.initialize()
    fi = 0
    for b, (name, idx, p) in enumerate(batches):
        r = s.read_rest(b)
        n = np.asarray(idx).reshape(-1, pkg.KIND_NODES[KIND[name]]).shape[0]
        for e in range(n):
            f = o.force(fi); fi += 1
            assert r["weight"][e] == f.weight, (name, e)
            assert r["global_idx"][e] == f.global_idx, (name, e)
            if name.startswith("TET"):
                assert np.array_equal(r["rest"][e], np.array(list(f.B)))
            elif name == "TRI_STRAIN":
                assert np.array_equal(r["rest"][e][:6], np.array(list(f.B))[:6])
            elif name == "BEND":
                assert np.array_equal(r["rest"][e][:4], np.array(list(f.alpha)))
            elif name == "SPRING":
                assert r["rest"][e][0] == f.measure
    assert s.info()["rows_compact"] == o.rows


def test_golden_weights(pkg):
    """weights against the compiled reference's (fixture), incl. the fp32 sqrtf path (TetForce.cpp:307)."""
    g = golden("assembly_bar.npz")
    s = pkg.make_bar_system(*tuple(g["dims"]), device_id=-1)
    s.initialize()
    nt = s.n_tets
    assert np.array_equal(s.read_rest(0)["weight"], g["weights"][:nt])
    assert np.array_equal(s.read_rest(1)["weight"], g["weights"][nt:])
    # compact global_idx = reference global_idx with 36 -> 9 rows per tet
    gi = s.read_rest(0)["global_idx"]
    assert np.array_equal(gi * 4, g["global_idx"][:nt])


@pytest.mark.parametrize("dims,leaf", [((3, 3, 5), 4), ((4, 4, 12), 16), ((6, 5, 17), 8), ((8, 8, 20), 32)])
def test_factor_solves_assembled_system(pkg, dims, leaf, monkeypatch):
    monkeypatch.setenv("ADMM_HIP_LEAF", str(leaf))
    s = pkg.make_bar_system(*dims, device_id=-1)
    s.initialize()
    n = s.n_nodes
    rng = np.random.default_rng(1)
    b = rng.normal(size=3 * n)
    x = s.debug_panel_solve_host(b)
    r = s.apply_A(x) - b
    assert np.abs(r).max() < 1e-11 * np.abs(b).max()
    # A is M + dt^2 D^T W^2 D: compare its action with the oracle's D, W
    mg = pkg.meshgen
    xx, t = mg.bar(*dims)
    m = mg.lumped_tet_mass(xx, t, 1000.0)
    o = Oracle(); o.settings(0.04, 1)
    o.add_nodes(xx.ravel(), np.repeat(m, 3))
    o.add_forces(KIND["TET_NH"], t, [1e5, 1e5, 5])
    o.add_forces(KIND["ANCHOR"], mg.bar_anchor_nodes(dims[0], dims[1]), [-1., 1.])
    assert o.initialize()
    rr, cc, vv = o.D_triplets()
    W = o.wdiag
    v = rng.normal(size=3 * n)
    Dv = np.zeros(o.rows); np.add.at(Dv, rr, vv * v[cc])
    y = np.zeros(3 * n); np.add.at(y, cc, vv * (0.04 ** 2 * W[rr] ** 2 * Dv[rr]))
    y += np.repeat(m, 3) * v
    assert np.abs(s.apply_A(v) - y).max() < 1e-10 * np.abs(y).max()


@pytest.mark.parametrize("env", [{"ADMM_HIP_MERGE": "0", "ADMM_HIP_MERGE_ROOT": "0"}, {"ADMM_HIP_MERGE": "0"}, {"ADMM_HIP_MERGE": "50"}, {"ADMM_HIP_MERGE": "50", "ADMM_HIP_MERGE_DEPTH": "3"},
                                 {"ADMM_HIP_MERGE": "0", "ADMM_HIP_MERGE_SMALL": "200"}])
def test_tree_shapes_factor_and_solve(pkg, env, monkeypatch):
    """Binary dissection tree, merged root, four-way nodes everywhere (the default below 160k nodes), eight-way nodes, four-way nodes only near the
    leaves: every shape of the elimination tree factors and solves the assembled system; fewer levels with the merged shapes."""
    monkeypatch.setenv("ADMM_HIP_LEAF", "8"); monkeypatch.setenv("ADMM_HIP_DENSE_MAX", "0")
    for k, v in env.items(): monkeypatch.setenv(k, v)
    s = pkg.make_bar_system(5, 5, 24, device_id=-1)
    s.initialize()
    n = s.n_nodes
    b = np.random.default_rng(3).normal(size=3 * n)
    x = s.debug_panel_solve_host(b)
    assert np.abs(s.apply_A(x) - b).max() < 1e-11 * np.abs(b).max()
    levels = s.info()["n_levels"]
    if env.get("ADMM_HIP_MERGE") == "50":
        monkeypatch.setenv("ADMM_HIP_MERGE", "0"); monkeypatch.setenv("ADMM_HIP_MERGE_ROOT", "0"); monkeypatch.delenv("ADMM_HIP_MERGE_DEPTH", raising=False)
        s2 = pkg.make_bar_system(5, 5, 24, device_id=-1); s2.initialize()
        assert levels < s2.info()["n_levels"]


@pytest.mark.parametrize("search", ["1", "0"])
def test_tree_search_factor_solves(pkg, search, monkeypatch, capfd):
    """Above the dense limit and with no ordering knob set, host_factor compares candidate elimination trees by the sweeps' cost model
    (ADMM_HIP_TREE_SEARCH=0: the rule-based tree only); whatever tree it ends up with factors and solves the system -- a bar and a
    two-body scene (two elimination-tree roots)."""
    monkeypatch.setenv("ADMM_HIP_TREE_SEARCH", search); monkeypatch.setenv("ADMM_HIP_VERBOSE", "1")
    for make in (lambda: pkg.make_bar_system(8, 8, 40, device_id=-1), lambda: pkg.make_mixed_system(6, 6, 24, 30, 30, device_id=-1)[0]):
        s = make(); s.initialize()
        assert s.info()["dense_solve"] == 0
        n = s.n_nodes
        b = np.random.default_rng(11).normal(size=3 * n)
        x = s.debug_panel_solve_host(b)
        assert np.abs(s.apply_A(x) - b).max() < 1e-10 * np.abs(b).max()
    err = capfd.readouterr().err
    assert ("tree search:" in err) == (search == "1")


def test_mixed_scene_factor(pkg):
    g = golden("traj_cloth.npz")
    n = g["x"].shape[0]
    s = pkg.System(device_id=-1); s.set_timestep(0.04)
    s.add_nodes(g["x"].ravel(), np.full(3 * n, float(g["mass"])))
    s.add_forces(KIND["TRI_STRAIN"], g["tris"], [100., .95, 1.05, 1.])
    s.add_forces(KIND["BEND"], g["hinges"], [20.])
    s.add_forces(KIND["ANCHOR"], g["anchors"], [-1., 1.])
    s.initialize()
    b = np.random.default_rng(2).normal(size=3 * n)
    x = s.debug_panel_solve_host(b)
    assert np.abs(s.apply_A(x) - b).max() < 1e-10 * np.abs(b).max()
