#!/usr/bin/env python3
"""Generate tests/golden/traj_dillo_grab.npz: BASELINE.json configs[2] as SURVEY 8(d) config 3 specifies it --
the shipped poordillo scene (2761 NH tets, dt 0.06, 10 iterations) with the sample's MovingAnchors on hand and
foot, dragged by helper::smooth_move to +2 x / -2 x over t in [1, 3] s (samples/poordillo/poordillo.cpp:133-166).

Runs in the BUILD container only: oracle/_ref/dillo_ref is tests/cpp/dillo_main.cpp compiled with the REAL
reference (its own SimContext, ForceBuilder, mclscene loader and admm::System; `make -C oracle dillo_ref`).

Outputs: x_frames [55][dof] of the scripted run; ulp_sensitivity [55] = max over starts perturbed by 1, 2, 3 ulps of |x' - x| per
frame (the reference against itself: the resolution any comparison with it can have); the same for a second run
in which the hand is released at frame 30 (the sample's H key: active = false, weight = 0, recompute_weights),
frames 28..44; the hand's first control point after that run (a released anchor follows its node).
"""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
EXE = os.path.join(ROOT, "oracle", "_ref", "dillo_ref")
XML = os.path.join(HERE, "scenes", "poordillo", "poordillo.xml")
FRAMES, RELEASE = 55, 30


def run(release=-1, ulps=0):
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "o.bin")
        subprocess.check_call([EXE, XML, f, str(FRAMES), str(release), str(ulps)], env=dict(os.environ, OMP_NUM_THREADS="1"), stdout=subprocess.DEVNULL)
        hdr = np.fromfile(f, dtype=np.int32, count=3)
        raw = np.fromfile(f, dtype=np.float64, offset=12)
    dof = int(hdr[0])
    return hdr, raw[:FRAMES * dof].reshape(FRAMES, dof), raw[FRAMES * dof:]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dillo_ref"])
    hdr, X, _ = run()
    env = np.max([np.abs(run(ulps=k)[1] - X).max(axis=1) for k in (1, 2, 3)], axis=0)
    _, R, cp = run(release=RELEASE)
    env_r = np.max([np.abs(run(release=RELEASE, ulps=k)[1] - R).max(axis=1) for k in (1, 2, 3)], axis=0)
    keep = np.arange(RELEASE - 2, RELEASE + 15)
    np.savez_compressed(os.path.join(HERE, "traj_dillo_grab.npz"), dof=hdr[0], n_hand=hdr[1], n_foot=hdr[2], frames=FRAMES, release_frame=RELEASE,
                        x_frames=X, ulp_sensitivity=env,
                        release_keep=keep, x_release=R[keep], ulp_sensitivity_release=env_r[keep], hand_cp_after_release=cp)
    print("scripted run: sensitivity to 1-3 ulp perturbations per frame", env)
    print("release run :", env_r[keep])
    print("largest displacement", np.abs(X[-1] - X[0]).max())


if __name__ == "__main__":
    main()
