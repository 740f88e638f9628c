#!/usr/bin/env python3
"""Feasibility probe (GPU box): V subtree shards of the 1M bar as V contexts on ONE GPU, one stream + one host thread each,
all-reduce done on the device (events between the streams, no host synchronisation).  Question: does the GPU overlap one
shard's latency-bound sweeps with another shard's VALU-bound tet kernel?  Compare wall time per ADMM iteration with V = 1.

  python tools/probe/shards_one_gpu.py [V ...]      (default 1 2 4)
"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
DIMS = (32, 32, 163)


class _Ptr:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def run(V, frames=4, prio=False):
    streams = [torch.cuda.Stream(priority=(-1 if (prio and r % 2 == 0) else 0)) for r in range(V)]
    shards = [pkg.make_bar_system(*DIMS, rank=r, world=V, stream=streams[r].cuda_stream, shard_mode=pkg.SHARD["subtree"]) for r in range(V)]
    bar = threading.Barrier(V)
    bufs = [None] * V; ready = [None] * V; done = [None] * V; cache = [dict() for _ in range(V)]

    def make_hook(r):
        def hook(ptr, count, strm):
            t = cache[r].get(ptr)
            if t is None:
                t = torch.as_tensor(_Ptr(ptr, count), device="cuda:0"); cache[r][ptr] = t
            S = streams[r]
            bufs[r] = t
            e = torch.cuda.Event(); e.record(S); ready[r] = e
            bar.wait()
            with torch.cuda.stream(S):
                for q in range(V):
                    if q != r: S.wait_event(ready[q])
                tot = bufs[0].clone()
                for q in range(1, V): tot += bufs[q]
                e2 = torch.cuda.Event(); e2.record(S); done[r] = e2
            bar.wait()
            with torch.cuda.stream(S):
                for q in range(V):
                    if q != r: S.wait_event(done[q])
                t.copy_(tot)
            return 0
        return hook
    if V > 1:
        for r, s in enumerate(shards): s.set_allreduce(make_hook(r))
    for s in shards: s.initialize()
    errs = []

    def work(r, n):
        try:
            for _ in range(n): shards[r].step(20)
        except Exception as e:  # noqa
            errs.append(e); raise

    def go(n):
        th = [threading.Thread(target=work, args=(r, n)) for r in range(V)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
    go(2)
    t0 = time.perf_counter(); go(frames); dt = time.perf_counter() - t0
    assert not errs
    x = shards[0].m_x
    print("V=%d%s  %.3f ms per ADMM iteration   (%.3e iters/s x tets)   checksum %.9g  finite %s" % (V, " prio" if prio else "", 1e3 * dt / (frames * 20), frames * 20 / dt * shards[0].n_tets, np.abs(x).sum(), np.isfinite(x).all()), flush=True)
    for s in shards: s.close() if hasattr(s, "close") else None


if __name__ == "__main__":
    Vs = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    for V in Vs:
        run(V)
    run(2, prio=True)
