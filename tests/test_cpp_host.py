"""The C++ host-side mirror of the reference's class API
(admm-elastic-sca_amd/host/admm/*.hpp over the C ABI): the reference's two
sample programs and a ForceBuilder-style scene, compiled with g++ against the
mirror and libadmm_hip.so.  CPU: they compile, link and fail loudly without a
GPU.  GPU: known answers and parity with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from checkers import KIND, Oracle
from conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "admm-elastic-sca_amd")
BUILD = os.path.join(ROOT, "tests", "_build")


def compile_cpp(name, pkg):
    pkg.lib()
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, name)
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    cmd = ["g++", "-std=c++11", "-O2", "-DADMM_HOST_NO_EIGEN", "-I" + os.path.join(PKG, "host"), "-I" + os.path.join(ROOT, "include"), src, "-o", out,
           "-L" + PKG, "-ladmm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return out


def have_gpu():
    import torch
    return torch.cuda.is_available()


def test_compiles_and_fails_loudly_without_gpu(pkg):
    for name in ("singletet", "singlenode", "scene_bar", "scene_plinko"):
        exe = compile_cpp(name, pkg)
        assert os.path.exists(exe)
    if have_gpu():
        pytest.skip("GPU present: behaviour covered by the gpu tests")
    r = subprocess.run([os.path.join(BUILD, "singletet")], capture_output=True, text=True)
    assert r.returncode == 2                       # initialize() returned false
    assert "no usable HIP device" in r.stderr or "no HIP device" in r.stderr


@pytest.mark.gpu
def test_singletet_singlenode(pkg):
    g = golden("known_answers.npz")
    r = subprocess.run([compile_cpp("singletet", pkg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Node 4 x: 171.571" in r.stdout          # the reference's printed answer
    full = [float(v) for v in r.stdout.split("full:")[1].split()]
    assert abs(full[0] - 171.57142857142716) < 1e-9 and np.abs(np.array(full) - g["singletet_x"][9:12]).max() < 1e-9
    r = subprocess.run([compile_cpp("singlenode", pkg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("step:")]
    assert [l.split("pos: ")[1] for l in lines] == ["(0, -9.8, 0)", "(0, -29.4, 0)", "(0, -58.8, 0)", "(0, -98, 0)"]


@pytest.mark.gpu
@pytest.mark.parametrize("typ", [0, 1])
def test_scene_through_class_api(pkg, tmp_path, typ):
    mg = pkg.meshgen
    dims = (3, 3, 8)
    x, t = mg.bar(*dims)
    m = mg.lumped_tet_mass(x, t, 1000.0)
    anchors = mg.bar_anchor_nodes(dims[0], dims[1])
    moving = x.shape[0] - 1
    inp = tmp_path / "in.bin"; outp = tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("4i", x.shape[0], t.shape[0], anchors.size, typ))
        f.write(x.astype(np.float64).tobytes()); f.write(np.repeat(m, 3).tobytes()); f.write(t.astype(np.int32).tobytes())
        f.write(anchors.astype(np.int32).tobytes()); f.write(struct.pack("i", moving))
    frames, iters = 6, 10
    r = subprocess.run([compile_cpp("scene_bar", pkg), str(inp), str(outp), str(frames), str(iters)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = np.fromfile(outp, dtype=np.float64)
    n3 = 3 * x.shape[0]
    X = raw[:frames * n3].reshape(frames, n3)
    nf = t.shape[0] + anchors.size + 1
    meta = raw[frames * n3:frames * n3 + 2 * nf].reshape(nf, 2)
    cp_final = raw[-3:]
    # the same scene on the oracle, control point scripted identically
    o = Oracle(); o.settings(0.04, iters)
    o.add_nodes(x.ravel(), np.repeat(m, 3))
    o.add_forces(KIND["TET_STVK" if typ else "TET_NH"], t, [1e5, 1e5, 5])
    o.add_forces(KIND["ANCHOR"], anchors, [-1.0, 1.0])
    start = x[moving].copy(); end = start + np.array([0, 0.05, 0])
    h = o.add_moving_anchor(moving, start, True, -1.0)
    o.add_gravity([0, -9.8, 0])
    assert o.initialize()
    assert np.array_equal(meta[:, 0].astype(np.int64), o.global_idx())       # element indexing
    assert np.array_equal(meta[:, 1], o.weights())                           # weights bit-exact
    elapsed = 0.0
    for fr in range(frames):
        tr = elapsed / 0.2
        pos = start if elapsed < 0 else (end if tr > 1 else start + (3 * tr * tr - 2 * tr ** 3) * (end - start))
        active = fr < 3
        if active:
            o.set_control_point(h, pos, True)
        else:
            o.set_control_point(h, np.array(list(o.force(h).pos)), False)
        o.step(); elapsed += 0.04
        assert np.abs(X[fr] - o.x).max() < 2e-4, fr       # within the truncated-prox sensitivity
    assert np.abs(cp_final - np.array(list(o.force(h).pos))).max() < 2e-4


@pytest.mark.gpu
def test_plinko_scene_through_class_api(pkg, tmp_path):
    g = golden("traj_collision.npz")
    n = g["x"].shape[0]
    inp = tmp_path / "in.bin"; outp = tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("3i", n, g["tets"].shape[0], g["types"].size))
        f.write(g["x"].astype(np.float64).tobytes()); f.write(np.repeat(g["mass"], 3).tobytes()); f.write(g["tets"].astype(np.int32).tobytes())
        f.write(g["types"].astype(np.int32).tobytes()); f.write(g["params"].astype(np.float64).tobytes())
    frames = int(g["frames"][-1]) + 1
    r = subprocess.run([compile_cpp("scene_plinko", pkg), str(inp), str(outp), str(frames), str(int(g["iters"]))], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "weight 32" in r.stdout and "global_idx %d" % (9 * g["tets"].shape[0]) in r.stdout
    X = np.fromfile(outp, dtype=np.float64).reshape(frames, 3 * n)
    for fi, f in enumerate(g["frames"]):
        assert np.abs(X[f] - g["x_frames"][fi]).max() < 1e-9      # the compiled reference's trajectory
