#!/usr/bin/env python3
"""Probe (GPU box): how does the tet kernel's time scale with the waves per SIMD?  Unused dynamic LDS (ADMM_HIP_TET_LDS_PAD) caps
the resident one-wave blocks per CU: 160 KB / (6 KB staging + pad).  Neo-Hookean (251 VGPRs: 2 waves per SIMD by registers) at
2 and 1; LinearTetStrain (164 VGPRs: 3 by registers; SVD + recompose, no line search) at 3, 2 and 1.
  python tools/probe/occupancy_scaling.py [dims=32x32x163]
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(*%(dims)r, kind=pkg.KIND[%(kind)r], device_id=0)
s.initialize()
for _ in range(3): s.step(20)
s.enable_timing(1)
loc = 0.0
for _ in range(2):
    s.step(20); loc += s.timing()["local_ms"] / 40.0
print("   %%-12s pad %%6d B (<= %%d waves per SIMD by LDS): local step %%.1f us" %% (%(kind)r, %(pad)d, (160 * 1024 // (6144 + %(pad)d)) // 4, 1e3 * loc), flush=True)
'''
dims = (32, 32, 163)
for a in sys.argv[1:]:
    if a.startswith("dims="): dims = tuple(int(v) for v in a[5:].split("x"))
print("%s bar, tet kernel time by waves per SIMD" % (dims,))
for kind, pads in (("TET_NH", (0, 14336, 34816)), ("TET_STVK", (0, 14336, 34816)), ("TET_LINEAR", (0, 7168, 14336, 34816))):
    for pad in pads:
        env = dict(os.environ, ADMM_HIP_TET_LDS_PAD=str(pad))
        r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, dims=dims, kind=kind, pad=pad)], env=env, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
