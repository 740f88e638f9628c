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


def test_every_environment_knob_is_documented():
    """Every ADMM_* / BENCH_* environment variable the library, the host classes, the Python plumbing or bench.py read is named in
    README.md's knob table -- and the table names none the sources no longer read (round 4 had 11 undocumented knobs)."""
    import glob
    srcs = glob.glob(os.path.join(ROOT, "admm-elastic-sca_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "admm-elastic-sca_amd", "host", "**", "*.hpp"), recursive=True) + \
        glob.glob(os.path.join(ROOT, "admm-elastic-sca_amd", "*.py")) + [os.path.join(ROOT, "bench.py")]
    read = set()
    for f in srcs:
        if os.path.isdir(f):
            continue
        txt = open(f, errors="replace").read()
        read |= set(re.findall(r'getenv\(\s*"((?:ADMM|BENCH)_[A-Z0-9_]+)"', txt))
        read |= set(re.findall(r'environ(?:\.get|\.setdefault|\.pop)?[\(\[]\s*"((?:ADMM|BENCH)_[A-Z0-9_]+)"', txt))
        read |= set(re.findall(r'"(ADMM_HIP_JOB_TAG)"', txt))
    assert len(read) >= 40, sorted(read)
    readme = open(os.path.join(ROOT, "README.md")).read()
    table = readme[readme.index("## Environment knobs"):]
    named = set(re.findall(r"`((?:ADMM|BENCH)_[A-Z0-9_]+)", table))
    missing = sorted(read - named)
    assert not missing, "environment knobs read by the sources but missing from README.md's table: %s" % missing
    removed = {"ADMM_HIP_PIPE", "ADMM_HIP_GROUPS", "ADMM_HIP_LOCAL_STREAMS", "ADMM_HIP_TET_LDS_PAD", "ADMM_HIP_STREAM_CUMASK", "ADMM_MULTI_EPL"}      # named as removed, on purpose
    stale = sorted(k for k in named - read - removed if not k.endswith("_"))
    assert not stale, "README.md's knob table names variables nothing reads any more: %s" % stale
