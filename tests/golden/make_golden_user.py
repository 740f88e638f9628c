#!/usr/bin/env python3
"""Generate tests/golden/user_force.npz: what the REAL reference computes for user-written plug-ins.

Runs in the BUILD container only.  tests/cpp/user_force.cpp (user subclasses of admm::Force,
admm::ExplicitForce and admm::CollisionShape) is compiled, unchanged, with the reference's own headers
and sources where they lie under /root/reference (g++ directly, the flags oracle/Makefile uses for
libadmm_ref.so) into oracle/_ref/user_force_ref, and run for every mode.  The GPU tests build the same
file against admm-elastic-sca_amd/host/admm + libadmm_hip.so and compare.

Outputs: x_mode<k> [frames][3 n] (m_x after every frame), gw_mode<k> [user forces][2] (global_idx, weight).
"""
import glob
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
REF = "/root/reference/deps/admm-elastic-sca"
FRAMES, ITERS, N = 6, 15, 9


def main():
    out = os.path.join(ROOT, "oracle", "_ref", "user_force_ref")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-std=c++11", "-O2", "-fopenmp", "-w", "-isystem", REF + "/deps/Eigen3", "-I" + REF + "/deps/cppoptlib/include",
           "-I" + REF + "/src/system", "-I" + REF + "/src/collision", os.path.join(ROOT, "tests", "cpp", "user_force.cpp")] + \
          sorted(glob.glob(REF + "/src/system/*.cpp")) + ["-o", out]
    subprocess.check_call(cmd)
    res = {"frames": FRAMES, "iters": ITERS, "n": N}
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for mode in (0, 1, 2, 3):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "o.bin")
            subprocess.check_call([out, str(mode), f, str(FRAMES), str(ITERS), str(N)], env=env)
            raw = np.fromfile(f, dtype=np.float64)
        n3 = 3 * N * N
        res["x_mode%d" % mode] = raw[:FRAMES * n3].reshape(FRAMES, n3)
        res["gw_mode%d" % mode] = raw[FRAMES * n3:].reshape(-1, 2)
    assert np.array_equal(res["x_mode0"], res["x_mode1"]), "the reference itself: MySpring != Spring"
    np.savez_compressed(os.path.join(HERE, "user_force.npz"), **res)
    print("wrote user_force.npz:", {k: getattr(v, "shape", v) for k, v in res.items()})


if __name__ == "__main__":
    main()
