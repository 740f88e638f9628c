#!/usr/bin/env python3
"""Probe (GPU box): V INDEPENDENT 1M-tet bars stepping concurrently on one GPU (one stream + host thread each).
Aggregate throughput vs V = 1 says how much latency-bound sweeps and VALU-bound tet kernels of different jobs overlap."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
DIMS = (32, 32, 163)


def run(V, frames=4):
    streams = [torch.cuda.Stream() for r in range(V)]
    sims = [pkg.make_bar_system(*DIMS, stream=streams[r].cuda_stream) for r in range(V)]
    for s in sims: s.initialize()

    def work(r, n):
        for _ in range(n): sims[r].step(20)

    def go(n):
        th = [threading.Thread(target=work, args=(r, n)) for r in range(V)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
    go(2)
    t0 = time.perf_counter(); go(frames); dt = time.perf_counter() - t0
    print("V=%d independent bars: %.3f ms per ADMM iteration per bar, aggregate %.3e iters/s x tets" % (V, 1e3 * dt / (frames * 20), V * frames * 20 / dt * sims[0].n_tets), flush=True)


for V in (1, 2, 3):
    run(V)
