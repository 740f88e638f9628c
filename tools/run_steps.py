#!/usr/bin/env python3
"""Runs a few frames of the bar (for rocprofv3 --pmc passes): python tools/run_steps.py nx ny nz frames"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
nx, ny, nz, frames = (int(v) for v in sys.argv[1:5])
s = pkg.make_bar_system(nx, ny, nz, device_id=0)
s.keep_z(False)      # production frames, like bench.py
s.initialize()
for _ in range(frames):
    s.step(20)
s.sync()
print("done", float(abs(s.m_x).sum()))
