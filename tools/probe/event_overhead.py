#!/usr/bin/env python3
"""Probe (GPU box): what do bench.py's HIP events cost the timed region?  One process, the 1M-tet bar, the SAME five frames every time (state rewound
to one checkpoint): event-free, events around every 20th / 10th / 4th / every ADMM iteration; two rounds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(32, 32, 163); s.keep_z(False); s.initialize()
for _ in range(2): s.step(20)
s.sync()
ck = dict(x=s.m_x.copy(), v=s.m_v.copy(), loc=[s.read_local(b) for b in range(len(s.batches))])
def rewind():
    s.m_x = ck["x"]; s.m_v = ck["v"]
    for bi, loc in enumerate(ck["loc"]):
        s.write_local(bi, u=loc["u"], state=loc["state"] if pkg.KIND_STATE[s.batches[bi][0]] else None)
    s.sync()
for rnd in range(2):
    for stride in (0, 20, 10, 4, 1):
        rewind()
        s.enable_timing(stride)
        t = time.perf_counter()
        for f in range(5):
            s.step(20)
            if stride and f > 0: s.timing_previous()
        if stride: s.timing()
        s.sync()
        dt = (time.perf_counter() - t) / 100
        print("round %d  events around every %2s ADMM iteration: %.4f ms per iteration (%.4g iters/s x tets)" % (rnd, stride if stride else "no", 1e3 * dt, s.n_tets / dt), flush=True)
