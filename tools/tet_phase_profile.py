#!/usr/bin/env python3
"""Phase attribution of project_tet_kernel (GPU box): builds a variant of
libadmm_hip.so with -DADMM_TET_PROFILE (s_memtime stamps at the phase boundaries,
per-lane loop counts reduced per wave), runs the bar, prints where a wave's
lifetime goes and how much of each loop is lost to lanes waiting for the slowest
lane of their wave.

  python tools/tet_phase_profile.py [dims=32x32x163] [frames=3]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(*%r, device_id=0)
s.initialize()
lib = pkg.lib()
buf = (ctypes.c_ulonglong * 128)()
frames = %d
for f in range(frames):
    s.step(20)
    s.m_x
    assert lib.admm_hip_debug_tet_profile(buf) == 0
    v = np.array(buf[:], dtype=np.float64)
    waves = v[16]
    names = ["load (idx, x gather, rest, u)", "oriented_svd", "L-BFGS prox", "recompose", "projection total (1+2+3+par/state)", "epilogue (u, z, RHS slots)"]
    tot = v[0] + v[4] + v[5]
    print("frame %%d: %%d wave-launches, mean wave lifetime %%.0f ticks" %% (f, waves, tot / waves))
    for i, nm in enumerate(names):
        print("   %%-40s %%6.1f %%%%" %% (nm, 100 * v[i] / tot))
    for nm, i in (("Jacobi sweeps", 8), ("Jacobi rotations", 10), ("L-BFGS outer iterations", 12), ("line-search evaluations", 14)):
        print("   %%-28s mean per tet %%6.2f   mean of wave maxima %%6.2f   -> lane efficiency %%.2f" %% (nm, v[i] / (64 * waves), v[i + 1] / (64 * waves), v[i] / max(v[i + 1], 1)))
    print("   line-search evaluations at a point evaluated just before (previous trial point or the base point): %%.1f %%%% of the lane evaluations; "
          "%%.1f %%%% of the wave-level evaluation steps have ONLY such lanes active" %% (100 * v[20] / max(v[18], 1), 100 * v[24] / max(v[22], 1)))
    h = v[32:64]; hw = v[64:96]
    print("   line-search evaluations per tet (%%):   " + " ".join("%%d:%%.1f" %% (k, 100 * h[k] / max(h.sum(), 1)) for k in range(32) if h[k]))
    print("   wave maximum of the same (%%):          " + " ".join("%%d:%%.1f" %% (k, 100 * hw[k] / max(hw.sum(), 1)) for k in range(32) if hw[k]))
    rg = v[96:128]; w = max(rg[0], 1)
    print("   region executions per WAVE (a wave issues a region once whatever its lane count): Jacobi sweeps %%.2f, rotations %%.2f, L-BFGS outer iterations %%.2f, history pairs %%.2f, "
          "line-search evaluations %%.2f (of which %%.2f go on to a step selection), log table branch %%.2f, log near-1 branch %%.2f, L-BFGS restarts %%.3f" %% (
              rg[1] / w, rg[2] / w, rg[6] / w, rg[7] / w, rg[8] / w, rg[9] / w, rg[10] / w, rg[11] / w, rg[14] / w))
    if rg[16]:
        print("   step selections (wave level): %%.2f per wave, %%.1f %%%% with every active lane in ONE case; a lane in case 1 / 2 / 3 / 4 present in %%.0f / %%.0f / %%.0f / %%.0f %%%% of them; %%.1f lanes active on average" %% (
            rg[16] / w, 100 * rg[17] / rg[16], 100 * rg[18] / rg[16], 100 * rg[19] / rg[16], 100 * rg[20] / rg[16], 100 * rg[21] / rg[16], rg[22] / rg[16]))
    print("   REGIONS " + " ".join("%%d:%%.4f" %% (k, rg[k] / w) for k in range(32) if rg[k]))
'''


def main():
    dims, frames = (32, 32, 163), 3
    for a in sys.argv[1:]:
        if a.startswith("dims="):
            dims = tuple(int(v) for v in a[5:].split("x"))
        if a.startswith("frames="):
            frames = int(a[7:])
    from __graft_entry__ import load_package
    pkg = load_package()
    out = os.path.join(ROOT, "admm-elastic-sca_amd", "_build", "libadmm_hip_prof.so")
    pkg._build.build(force=False, extra_hip_flags=["-DADMM_TET_PROFILE"], out=out, tag="_prof")
    env = dict(os.environ, ADMM_HIP_LIB=out)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, dims, frames)], env=env, capture_output=True, text=True)
    print(r.stdout)
    if r.returncode:
        print(r.stderr[-2000:])


if __name__ == "__main__":
    main()
