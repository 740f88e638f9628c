#!/usr/bin/env python3
"""Probe (GPU box, run under rocprofv3 --kernel-trace): repeated solves WITHOUT a local step in between, to compare the sweeps'
per-level durations with those inside a full ADMM iteration (what the tet kernel's traffic costs the forward sweep's first levels)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(32, 32, 163); s.keep_z(False); s.initialize()
b = np.random.default_rng(0).normal(size=3 * s.n_nodes)
for _ in range(6):
    x = s.solve_only(b)
print("done", float(np.abs(x).sum()))
