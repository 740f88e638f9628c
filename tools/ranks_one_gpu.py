#!/usr/bin/env python3
"""GPU box: what every RANK of an N-rank run computes per ADMM iteration, measured on ONE GPU with REAL physics.

N contexts (one per rank, the real sharded code path: own elements, own subtrees, replicated top, one exchange per
iteration) live in one process, each driven by its own host thread.  The all-reduce hook sums the ranks' buffers for real,
so the simulation is the N-rank simulation -- and it passes a baton: rank r may leave exchange k only after rank r - 1 has
arrived at exchange k + 1.  At any moment ONE rank's kernels are on the GPU, so each context's own HIP events
(admm_hip_enable_timing(1)) time that rank's local step, right-hand side and sweeps as if it had the GPU to itself.  What
is NOT measured: the collective itself (xGMI) -- the hook's wait shows up in allreduce_ms and is ignored.

Replaces the round-2..4 methodology (a fake world with a no-op all-reduce for the sweeps + ADMM_HIP_PIPE groups for the
local step): one run, right physics, no extra code path in the library.

  python tools/ranks_one_gpu.py --world 4 [--mode subtree|contiguous] [--dims 32 32 163 | --config mixed] [--frames 3] [--warm 1]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


class Baton:
    """all-reduce between `world` contexts of one process + the serialising baton (see the module docstring)"""

    def __init__(self, world, torch, serialise=True):
        self.world, self.torch, self.serialise = world, torch, serialise
        self.cv = threading.Condition()
        self.arrived = [0] * world          # exchanges rank r has entered
        self.summed = 0                     # exchanges whose sum is complete
        self.finished = [False] * world
        self.bufs = [None] * world
        self.failed = False

    class _Ptr:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def hook(self, r):
        torch = self.torch

        def fn(ptr, count, stream):
            torch.cuda.synchronize()        # this rank's kernels up to the exchange are done (nobody else is running)
            with self.cv:
                c = self.arrived[r]
                self.bufs[r] = torch.as_tensor(Baton._Ptr(ptr, count), device="cuda:0")
                self.arrived[r] = c + 1
                self.cv.notify_all()
                if not self.cv.wait_for(lambda: self.failed or all(a >= c + 1 for a in self.arrived), timeout=600):
                    self.failed = True
                if self.failed:
                    self.cv.notify_all(); return 1
                if r == 0:
                    tot = self.bufs[0].clone()
                    for q in range(1, self.world):
                        tot += self.bufs[q]
                    for q in range(self.world):
                        self.bufs[q].copy_(tot)
                    torch.cuda.synchronize()
                    self.summed = c + 1
                    self.cv.notify_all()
                else:
                    ok = self.cv.wait_for(lambda: self.failed or self.summed >= c + 1, timeout=600)
                    # the baton: the rank before me has done its post-exchange work and reached the NEXT exchange (or is through)
                    ok = ok and (not self.serialise or self.cv.wait_for(lambda: self.failed or self.finished[r - 1] or self.arrived[r - 1] >= c + 2, timeout=600))
                    if not ok or self.failed:
                        self.failed = True; self.cv.notify_all(); return 1
            return 0
        return fn

    def done(self, r):
        torch = self.torch
        torch.cuda.synchronize()
        with self.cv:
            self.finished[r] = True
            self.cv.notify_all()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--world", type=int, default=4)
    p.add_argument("--mode", default="subtree")
    p.add_argument("--dims", type=int, nargs=3, default=[32, 32, 163])
    p.add_argument("--frames", type=int, default=3)
    p.add_argument("--warm", type=int, default=1)
    p.add_argument("--iters", type=int, default=20)
    p.add_argument("--config", choices=["bar", "mixed"], default="bar", help="mixed: BASELINE configs[4] (26x26x123 NH + StVK bar, 158x158 cloth)")
    p.add_argument("--json", action="store_true")
    a = p.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    W = a.world
    if a.config == "mixed":
        shards = [pkg.make_mixed_system(26, 26, 123, 158, 158, rank=r, world=W, shard_mode=a.mode)[0] for r in range(W)]
    else:
        shards = [pkg.make_bar_system(*a.dims, rank=r, world=W, shard_mode=a.mode) for r in range(W)]
    bat0, bat = Baton(W, torch, serialise=False), Baton(W, torch)
    for r, s in enumerate(shards):
        s.set_allreduce(bat0.hook(r)); s.keep_z(False)      # (initialize: plain sums -- a rank that is through must not wait for a next exchange)
    # initialize() is a collective call under rank-local factorization (the default for subtree shards): one thread per rank
    t_init0 = time.time()
    pkg.initialize_together(shards)
    t_init = time.time() - t_init0
    for r, s in enumerate(shards):
        s.set_allreduce(bat.hook(r))
    keys = ("local_ms", "rhs_ms", "solve_fwd_ms", "solve_bwd_ms", "allreduce_ms", "total_ms")
    acc = [[dict.fromkeys(keys, 0.0) for _ in range(a.frames)] for _ in range(W)]
    errs = []

    def run(r):
        try:
            s = shards[r]
            for f in range(a.warm):
                s.step(a.iters)
            s.enable_timing(1)
            for f in range(a.frames):
                s.step(a.iters)
                t = s.timing()
                for k in keys:
                    acc[r][f][k] = t[k] / a.iters
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            with bat.cv:
                bat.failed = True; bat.cv.notify_all()
        finally:
            bat.done(r)
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise SystemExit("ranks failed: %r" % errs)
    xs = [s.m_x for s in shards]
    same = all(np.array_equal(xs[0], x) for x in xs[1:])
    infos = [s.info() for s in shards]
    out = {"world": W, "mode": a.mode, "dims": a.dims, "frames": a.frames, "warm": a.warm, "iters": a.iters, 
           "all_ranks_bitwise_equal": bool(same), "finite": bool(np.isfinite(xs[0]).all()), "x_checksum": float(np.abs(xs[0]).sum()),
           "per_frame": [], "elements": [int(i["n_elems_local"]) for i in infos], "nodes_own": [int(i["nodes_own"]) for i in infos], "nodes_top": int(infos[0]["nodes_top"]),
           "comm_bytes_per_iter": 8 * int(infos[0]["comm_doubles_iter"]), "dist_top": int(infos[0]["dist_top"]),
           "top_gb_streamed_per_iter": [round(8e-9 * (i["sweep_entries_top"] + i["sweep_entries_top_bwd"]), 4) for i in infos],
           "own_gb_streamed_per_iter": [round(8e-9 * 2 * i["sweep_entries_own"], 4) for i in infos],
           # rank-local factorization: what every rank factors and keeps (one GPU hosts all ranks here, so the wall time is the SUM of the ranks' work)
           "factor_local": [int(i["factor_local"]) for i in infos], "factor_gb_resident": [round(8e-9 * i["factor_doubles_resident"], 4) for i in infos],
           "factor_gb_whole": round(1e-9 * infos[0]["panel_bytes"], 4), "front_gb": [round(8e-9 * i["front_doubles"], 4) for i in infos],
           "factor_exchange_mb": round(8e-6 * infos[0]["factor_exchange_doubles"], 3), "t_numeric_s": [round(i["t_numeric_s"], 4) for i in infos],
           "initialize_all_ranks_on_one_gpu_s": round(t_init, 3)}
    for f in range(a.frames):
        row = {k: [round(acc[r][f][k], 4) for r in range(W)] for k in keys if k not in ("allreduce_ms", "total_ms")}
        busy = [acc[r][f]["local_ms"] + acc[r][f]["rhs_ms"] + acc[r][f]["solve_fwd_ms"] + acc[r][f]["solve_bwd_ms"] for r in range(W)]
        row["busy_ms"] = [round(b, 4) for b in busy]
        row["critical_path_ms"] = round(max(busy), 4)
        out["per_frame"].append(row)
    if a.json:
        print(json.dumps(out))
        return
    print("ranks on one GPU: world %d, %s shards, bar %s, %d warm-up + %d timed frames of %d iterations%s" % (
        W, a.mode, "x".join(map(str, a.dims)) if a.config == "bar" else "(mixed scene of configs[4])", a.warm, a.frames, a.iters, ""))
    print("elements per rank %s; nodes own %s + top %d; exchange %d bytes per iteration; all ranks bitwise equal: %s" % (
        out["elements"], out["nodes_own"], out["nodes_top"], out["comm_bytes_per_iter"], same))
    print("top of the tree: %s; a rank streams per iteration: own subtrees (both sweeps) %s GB, top %s GB" % (
        "DISTRIBUTED (one root supernode, its product split by rows; two collectives per iteration)" if out["dist_top"] else "replicated (every rank sweeps it; one collective per iteration)",
        out["own_gb_streamed_per_iter"], out["top_gb_streamed_per_iter"]))
    print("factor: %s; resident per rank %s GB of %.3f GB (sum %.3f GB); fronts per rank %s GB; exchanged once per factorization %.1f MB; numeric phase per rank %s s" % (
        "rank-local" if all(out["factor_local"]) else "whole on every rank", out["factor_gb_resident"], out["factor_gb_whole"], sum(out["factor_gb_resident"]), out["front_gb"],
        out["factor_exchange_mb"], out["t_numeric_s"]))
    for f, row in enumerate(out["per_frame"]):
        print("frame %d  ms per ADMM iteration and rank (communication excluded)" % f)
        for k in ("local_ms", "rhs_ms", "solve_fwd_ms", "solve_bwd_ms", "busy_ms"):
            print("   %-13s %s   slowest %.4f fastest %.4f" % (k, " ".join("%.4f" % v for v in row[k]), max(row[k]), min(row[k])))
        print("   critical path (slowest rank's kernels) %.4f ms" % row["critical_path_ms"])


if __name__ == "__main__":
    main()
