#!/usr/bin/env python3
"""Probe (GPU box, run under rocprofv3 --kernel-trace): what ONE rank's local step costs with REAL physics.  The fake-world runs
(tools/fake_world.sh) time a rank's launch sequence with a no-op all-reduce, i.e. with a wrong right-hand side: fine for the sweeps
(data-independent) but the tet kernel's time depends on its data.  Here a single-rank run with ADMM_HIP_PIPE=G in its serial mode
(timing events on) launches the tet kernel once per subtree group -- the same partition the rank sharding uses -- on correct data."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(32, 32, 163); s.keep_z(False); s.initialize()
s.enable_timing(1)
for _ in range(3): s.step(20)
s.sync()
print("done")
