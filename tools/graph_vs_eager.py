#!/usr/bin/env python3
"""GPU box: ms per ADMM iteration of the 1M-tet bar, event-free: one HIP graph replay per iteration vs eager launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(32, 32, 163)
s.initialize()
for _ in range(2):
    s.step(20)
s.sync()
t = time.perf_counter()
for _ in range(5):
    s.step(20)
s.sync()
print("ADMM_HIP_GRAPH=%s: %.4f ms per ADMM iteration" % (os.environ.get("ADMM_HIP_GRAPH", "1"), (time.perf_counter() - t) / 100 * 1e3))
