"""GPU box: two fresh systems, same scene, 4 frames each -> bitwise equal states?  (cost-ordered launch, two-column backward levels,
device factorization: all deterministic in their results.)  python tools/probe/repro.py [nx ny nz]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package()
dims = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 32, 60)
out = []
for run in range(2):
    s = pkg.make_bar_system(*dims)
    s.initialize()
    for f in range(4):
        s.step(20)
    out.append((s.m_x.copy(), s.m_v.copy()))
    del s
print("tets", dims[0] * dims[1] * dims[2] * 6, "x equal:", np.array_equal(out[0][0], out[1][0]), "v equal:", np.array_equal(out[0][1], out[1][1]),
      "max |dx| %.3e" % np.abs(out[0][0] - out[1][0]).max())
