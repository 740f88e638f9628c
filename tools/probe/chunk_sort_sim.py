#!/usr/bin/env python3
"""Offline: would re-grouping the tets into waves by cost lower the waves' maximum line-search evaluation counts?  Reads the per-tet trace
gpurun_out/ls_trace.npz (tools/probe/ls_predict_gpu.py: 20 consecutive ADMM iterations of the 1M-tet bar) and deals chunks of C consecutive
tets to waves by (a) this iteration's cost (an oracle) and (b) the previous iteration's cost.  Result: profiles/r04/chunk_sort_simulation.txt."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nf = np.load(os.path.join(ROOT, "gpurun_out", "ls_trace.npz"))["nfev"].astype(np.int32)
it, n = nf.shape
nw = n // 64
base = nf[:, :nw * 64].reshape(it, nw, 64).max(2)
print("mean evaluations per tet %.2f, mean wave maximum %.2f; tets with >= 15 evaluations %.1f %%, waves holding one %.1f %%" % (nf.mean(), base.mean(), 100 * (nf >= 15).mean(), 100 * (base >= 15).mean()))
for C in (64, 32, 16, 8, 4, 1):
    nc = n // C; ch = nf[:, :nc * C].reshape(it, nc, C).max(2); g = 64 // C
    def dealt(key_of):
        tot = 0.0; cnt = 0
        for i in range(1, it):
            o = np.argsort(-key_of(i), kind="stable")
            tot += ch[i][o][:(nc // g) * g].reshape(-1, g).max(1).mean(); cnt += 1
        return tot / cnt
    print("chunk %2d: chunks with a slow tet %.1f %%; wave maximum with key = this iteration %.2f, key = previous iteration %.2f" % (C, 100 * (ch >= 15).mean(), dealt(lambda i: ch[i]), dealt(lambda i: ch[i - 1])))
