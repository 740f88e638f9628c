"""CPU: the C-ABI library loads, exports every symbol include/admm_hip.h
declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "admm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(admm_hip_[a-z_A-Z0-9]+)\s*\(", txt)) - {"admm_hip_allreduce_fn"})


def test_exports(pkg):
    lib = pkg.lib()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libadmm_hip.so does not export %s" % s


def test_kind_tables_match_header(pkg):
    txt = open(os.path.join(ROOT, "include", "admm_kinds.h")).read()
    for name, table in (("ADMM_KIND_NODES", pkg.KIND_NODES), ("ADMM_KIND_ROWS", pkg.KIND_ROWS), ("ADMM_KIND_PARAMS", pkg.KIND_PARAMS),
                        ("ADMM_KIND_STATE", pkg.KIND_STATE)):
        m = re.search(name + r"\[ADMM_KIND_COUNT\]\s*=\s*\{([^}]*)\}", txt)
        assert [int(v) for v in m.group(1).split(",")] == table


def test_no_gpu_means_no_compute(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.AdmmHipError):
        pkg.System(device_id=0)          # must fail loudly: no device
    s = pkg.make_bar_system(2, 2, 3, device_id=-1)   # host-only context: assembly + factor only
    s.initialize()
    with pytest.raises(pkg.AdmmHipError):
        s.step(1)
    with pytest.raises(pkg.AdmmHipError):
        s.solve_only(np.zeros(3 * s.n_nodes))
    with pytest.raises(pkg.AdmmHipError):
        s.local_step_only(np.zeros(3 * s.n_nodes))


def test_argument_errors(pkg):
    s = pkg.System(device_id=-1)
    s.add_nodes(np.zeros(9), np.ones(9))
    with pytest.raises(pkg.AdmmHipError):
        s.add_forces(pkg.KIND["TET_NH"], [[0, 1, 2, 7]], [1e5, 1e5, 5]) or s.initialize()   # node 7 does not exist
    s = pkg.System(device_id=-1)
    s.add_nodes(np.zeros(6), np.array([1, 1, 2, 1, 1, 1.]))   # anisotropic mass: outside the scalar-system path
    with pytest.raises(pkg.AdmmHipError):
        s.initialize()
    s = pkg.System(device_id=-1)
    with pytest.raises(pkg.AdmmHipError):
        s.initialize()                                          # no nodes (System.cpp:108-111)
