#!/usr/bin/env python3
"""GPU box: a long run of the headline workload -- frames x 20 ADMM iterations of the 1M-tet bar -- for stability: finite state, no device-memory growth,
frame time per block of 100 frames.   python tools/probe/soak.py [frames]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
s = pkg.make_bar_system(32, 32, 163); s.keep_z(False); s.initialize()
for _ in range(3):
    s.step(20)
s.sync()
free0 = torch.cuda.mem_get_info()[0]
print("soak: 1,001,472-tet NH bar, %d frames x 20 ADMM iterations" % frames)
t_all = time.perf_counter()
for blk in range(0, frames, 100):
    n = min(100, frames - blk)
    t = time.perf_counter()
    for _ in range(n):
        s.step(20)
    s.sync()
    dt = (time.perf_counter() - t) / n
    x = s.m_x
    print("frames %4d-%4d: %.3f ms per frame (%.4f ms per iteration), max |x| %.4f, finite %s, device memory delta %+d bytes" % (
        blk + 4, blk + 3 + n, 1e3 * dt, 1e3 * dt / 20, np.abs(x).max(), bool(np.isfinite(x).all()), torch.cuda.mem_get_info()[0] - free0))
    assert np.isfinite(x).all()
print("soak: %.1f s wall for %d frames; ok" % (time.perf_counter() - t_all, frames))
