"""CPU (host-only contexts): the generic batch of user-defined forces -- assembly of its share of A_s, the K (x) I3 check,
argument errors, and that an element spanning all nodes (a CollisionForce with user-written shapes) assembles in linear time."""
import time

import numpy as np
import pytest


def _sys(pkg, n, x=None):
    s = pkg.System(device_id=-1)
    s.set_timestep(0.04)
    x = np.random.default_rng(0).normal(size=(n, 3)) if x is None else x
    s.add_nodes(x.ravel(), np.ones(3 * n))
    return s, x


def _spring_triplets(pairs):
    ne = pairs.shape[0]
    rows = np.repeat(np.arange(ne) * 3, 3) + np.tile(np.arange(3), ne)
    tr = np.concatenate([rows, rows]).astype(np.int32)
    tc = np.concatenate([3 * np.repeat(pairs[:, 0], 3) + np.tile(np.arange(3), ne), 3 * np.repeat(pairs[:, 1], 3) + np.tile(np.arange(3), ne)]).astype(np.int32)
    tv = np.concatenate([np.ones(3 * ne), -np.ones(3 * ne)])
    return np.arange(ne + 1) * 3, tr, tc, tv


def test_generic_spring_assembles_like_builtin_spring(pkg):
    n = 40
    rng = np.random.default_rng(1)
    pairs = np.array([(i, j) for i in range(n) for j in rng.choice(n, 3, replace=False) if i != j], np.int32)
    k = 250.0
    a, x = _sys(pkg, n)
    a.add_forces(pkg.KIND["SPRING"], pairs, [k]); a.initialize()
    b, _ = _sys(pkg, n, x)
    erp, tr, tc, tv = _spring_triplets(pairs)
    b.add_generic(erp, tr, tc, tv, np.full(3 * pairs.shape[0], np.sqrt(k))); b.initialize()
    v = rng.normal(size=3 * n)
    assert np.array_equal(a.apply_A(v), b.apply_A(v))             # the same scalar system, entry for entry
    assert a.info()["nnz_A"] == b.info()["nnz_A"] and b.info()["rows_compact"] == 3 * pairs.shape[0]


def test_selectors_outside_the_scalar_system_are_refused(pkg):
    s, _ = _sys(pkg, 4)
    # one row that couples the x of node 0 with the y of node 1
    s.add_generic([0, 1], [0, 0], [0, 4], [1.0, -1.0], [1.0])
    with pytest.raises(pkg.AdmmHipError, match="not of the form K"):
        s.initialize()
    s, _ = _sys(pkg, 4)
    # x / y / z rows of one force with different weights
    s.add_generic([0, 3], [0, 1, 2], [0, 1, 2], [1.0, 1.0, 1.0], [1.0, 2.0, 1.0])
    with pytest.raises(pkg.AdmmHipError, match="not of the form K"):
        s.initialize()
    s, _ = _sys(pkg, 4)
    with pytest.raises(pkg.AdmmHipError):
        s.add_generic([0, 3], [0, 1, 5], [0, 1, 2], [1.0, 1.0, 1.0], [1.0, 1.0, 1.0])      # triplet row 5 of 3 rows
    s, _ = _sys(pkg, 4)
    s.add_generic([0, 3], [0, 1, 2], [30, 31, 32], [1.0, 1.0, 1.0], [1.0, 1.0, 1.0])       # node 10 of 4
    with pytest.raises(pkg.AdmmHipError, match="references node"):
        s.initialize()


def test_element_over_all_nodes_assembles_in_linear_time(pkg):
    """A CollisionForce with a user-written shape is ONE user force whose identity rows span every node (CollisionForce.cpp:29-36)."""
    n = 60000
    s, _ = _sys(pkg, n, np.random.default_rng(2).uniform(size=(n, 3)))
    rows = np.arange(3 * n, dtype=np.int32)
    s.add_generic([0, 3 * n], rows, rows, np.ones(3 * n), np.full(3 * n, 32.0))
    t0 = time.time()
    s.initialize()
    assert time.time() - t0 < 20.0
    inf = s.info()
    assert inf["nnz_A"] == n and inf["rows_compact"] == 3 * n      # diagonal: mass + dt^2 w^2
    v = np.ones(3 * n)
    assert np.allclose(s.apply_A(v), 1.0 + 0.04 * 0.04 * 32.0 * 32.0)


def test_duplicate_triplets_are_summed(pkg):
    """Eigen's setFromTriplets sums duplicates (System.cpp:125); so does the generic batch."""
    n = 6
    pairs = np.array([(0, 1), (2, 5), (3, 4)], np.int32)
    erp, tr, tc, tv = _spring_triplets(pairs)
    a, x = _sys(pkg, n)
    a.add_generic(erp, tr, tc, tv, np.full(9, 2.0)); a.initialize()
    b, _ = _sys(pkg, n, x)
    b.add_generic(erp, np.concatenate([tr, tr]), np.concatenate([tc, tc]), np.concatenate([0.25 * tv, 0.75 * tv]), np.full(9, 2.0)); b.initialize()
    v = np.random.default_rng(3).normal(size=3 * n)
    assert np.allclose(a.apply_A(v), b.apply_A(v), rtol=0, atol=1e-15) and a.info()["nnz_A"] == b.info()["nnz_A"]
