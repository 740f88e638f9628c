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


CASES = [("TET_NH", [1e5, 1e5, 5]), ("TET_NH", [50, 80, 20]), ("TET_STVK", [100, 100, 5]), ("TET_STVK", [3e3, 1e3, 12]), ("TET_LINEAR", [1.0]),
         ("TET_VOLUME", [100, 0.9, 1.1]), ("TRI_STRAIN", [100, .95, 1.05, 1]), ("TRI_STRAIN", [10, .5, 2, 0]), ("BEND", [20.]), ("SPRING", [50.]),
         ("ANCHOR", [55., 1])]


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
        keys = ("z", "u", "init") + (("state", "n_iters") if name in ("TET_NH", "TET_STVK") else ())
        for k in keys:
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
