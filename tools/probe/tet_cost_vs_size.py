#!/usr/bin/env python3
"""GPU box: is the tet kernel's cost per tet at 4M / 16M tets a matter of the deformation (more line-search work) or of memory (x no longer cache-resident)?
Per size: frames 1..4 of the bar, local-step ms per iteration from the context's events, ns per tet, and the mean / max L-BFGS iteration count of the last local step."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from __graft_entry__ import load_package
pkg = load_package()
for dims in ((32, 32, 163), (64, 64, 163), (80, 80, 417)):
    s = pkg.make_bar_system(*dims)
    s.initialize()
    s.enable_timing(1)
    out = []
    for f in range(4):
        s.step(20)
        t = s.timing()
        it = s.read_local(0)["n_iters"]
        out.append("frame %d: local %.4f ms = %.3f ns/tet, L-BFGS iterations mean %.2f max %d" % (f + 1, t["local_ms"] / 20, 1e6 * t["local_ms"] / 20 / s.n_tets, it.mean(), it.max()))
    print("bar %dx%dx%d, %d tets:" % (dims + (s.n_tets,)))
    for o in out:
        print("   " + o)
    del s
