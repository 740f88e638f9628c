#!/usr/bin/env python3
"""Offline: what would re-packing the still-searching tets of a workgroup into fewer waves save?  Per-tet evaluation counts of 20 consecutive
ADMM iterations of the 1M-tet bar (gpurun_out/ls_trace.npz, tools/probe/ls_predict_gpu.py; counts are per tet and iteration, all line
searches of the tet together).  A workgroup of W waves = 64 W consecutive tets.  At a compaction point (before evaluation step p) the tets
that still need step p are packed, in order, into ceil(n / 64) waves; between points a wave runs until the last of ITS lanes is done.
Output: wave-steps per 64 tets (today: the mean wave maximum; floor: the mean per tet)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nf = np.load(os.path.join(ROOT, "gpurun_out", "ls_trace.npz"))["nfev"].astype(np.int32)[::4]      # every fourth iteration: enough
it, n = nf.shape
print("mean evaluations per tet %.2f (the floor: every lane busy), mean wave maximum today %.2f" % (nf.mean(), nf[:, :n // 64 * 64].reshape(it, -1, 64).max(2).mean()))


def cost(v, points):
    """wave-steps of one workgroup whose tets need v[i] evaluation steps"""
    total = 0
    cur = v
    t0 = 0                                      # steps done so far
    for p in list(points) + [10 ** 9]:
        # waves of the current packing run from step t0 + 1 up to step p - 1 (or their own end)
        m = cur.reshape(-1, 64).max(1) if cur.size % 64 == 0 else np.array([cur[i:i + 64].max() for i in range(0, cur.size, 64)])
        total += np.clip(np.minimum(m, p - 1) - t0, 0, None).sum()
        if p >= 10 ** 9: break
        cur = cur[cur >= p]                     # survivors, in order
        t0 = p - 1
        if cur.size == 0: break
    return total


for W in (2, 4, 8):
    g = n // (64 * W)
    a = nf[:, :g * 64 * W].reshape(it * g, W * 64)
    for points in ((), (4,), (3, 5), (3, 6), (4, 8), (3, 5, 8), (2, 3, 4, 5, 6, 8, 12)):
        tot = sum(cost(a[i], points) for i in range(0, a.shape[0], 3))
        cnt = len(range(0, a.shape[0], 3)) * W
        print("W = %d waves, compaction before steps %-24s: %.2f wave-steps per 64 tets, %d compactions" % (W, points, tot / cnt, len(points)))
