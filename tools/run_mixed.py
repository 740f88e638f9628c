#!/usr/bin/env python3
"""Runs a few frames of the mixed scene of BASELINE configs[4] (for rocprofv3 traces): python tools/run_mixed.py frames"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
s, _ = pkg.make_mixed_system(26, 26, 123, 158, 158)
s.keep_z(False)
s.initialize()
for _ in range(int(sys.argv[1])):
    s.step(20)
s.sync()
print("done", float(abs(s.m_x).sum()))
