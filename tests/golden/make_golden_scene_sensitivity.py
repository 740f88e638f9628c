#!/usr/bin/env python3
"""tests/golden/scene_sensitivity.npz: how far the REFERENCE's own trajectory of the two shipped hyperelastic scenes moves when its
start is perturbed by 1, 2 and 3 ulps (its truncated L-BFGS / More'-Thuente prox is discontinuous in its input) --
the per-frame envelope that test_shipped_scene_trajectories asserts against, instead of one flat tolerance per scene.

Runs in the BUILD container only (oracle/_ref/libscene_ref.so = the reference's SimContext + ForceBuilder + loader, compiled from
/root/reference by `make -C oracle scene_ref`); one child process per (scene, perturbation): the reference keeps static state.
Output: per scene `<name>_env` [frames] = max over the perturbed starts of max |x' - x| per frame, `<name>_frames`.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "..", "oracle", "_ref", "libscene_ref.so")
SCENES = {"poordillo": ("scenes/poordillo/poordillo.xml", "none", 2), "bunnyexpand": ("scenes/bunnyexpand/bunnyexpand.xml", "scale1.3", 2)}


def child(name, ulps, out_path):
    xml, setup, frames = SCENES[name]
    L = C.CDLL(LIB)
    L.refscene_load.restype = C.c_void_p
    L.refscene_load.argtypes = [C.c_char_p]
    for f in ("refscene_initialize", "refscene_step", "refscene_dof"):
        getattr(L, f).argtypes = [C.c_void_p]
    dp = np.ctypeslib.ndpointer(np.float64, flags="C")
    L.refscene_get_x.argtypes = [C.c_void_p, dp]
    L.refscene_set_x.argtypes = [C.c_void_p, dp]
    h = L.refscene_load(os.path.join(HERE, xml).encode())
    assert h and L.refscene_initialize(h)
    dof = L.refscene_dof(h)
    x = np.zeros(dof)
    L.refscene_get_x(h, x)
    if setup == "scale1.3":
        x = np.ascontiguousarray(x * 1.3)
    for _ in range(ulps):
        x = np.nextafter(x, np.inf)
    L.refscene_set_x(h, np.ascontiguousarray(x))
    traj = []
    for _ in range(frames):
        L.refscene_step(h)
        L.refscene_get_x(h, x)
        traj.append(x.copy())
    np.save(out_path, np.array(traj))
    os._exit(0)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], int(sys.argv[2]), sys.argv[3])
    else:
        if not os.path.exists(LIB):
            sys.exit("build oracle/_ref/libscene_ref.so first: make -C oracle scene_ref")
        out = {}
        for nm, (_, _, frames) in SCENES.items():
            runs = []
            for ulps in (0, 1, 2, 3):
                tmp = "/tmp/scene_sens_%s_%d.npy" % (nm, ulps)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), nm, str(ulps), tmp], env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
                assert r.returncode == 0, r.stdout + r.stderr
                runs.append(np.load(tmp))
            base = np.load(os.path.join(HERE, "scene_%s.npz" % nm))["traj"]
            assert np.array_equal(base, runs[0]), "the unperturbed run does not reproduce the committed fixture"
            env = np.max([np.abs(r - runs[0]).max(axis=1) for r in runs[1:]], axis=0)
            out[nm + "_env"] = env; out[nm + "_frames"] = frames
            print(nm, "per-frame envelope", env, "max |x|", np.abs(base).max())
        np.savez_compressed(os.path.join(HERE, "scene_sensitivity.npz"), **out)
