"""Elimination-tree policy A/B on mid-size bars: solve residual, bitwise run-to-run, forward / backward / local time per ADMM iteration for
the library's default and for every extra variant given as VAR=value[,VAR=value...] (ADMM_HIP_LEAF, ADMM_HIP_MERGE; with
profiles/r03/experiments/subtree_walkers.patch applied also ADMM_HIP_WALK...).  WALK_AB_SCENES=i,j selects scenes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
scenes = (((13, 13, 50), "TET_STVK"), ((10, 10, 30), "TET_NH"), ((20, 20, 60), "TET_NH"), ((24, 24, 100), "TET_NH"),
          ((22, 22, 70), "TET_NH"), ((24, 24, 75), "TET_NH"), ((28, 28, 120), "TET_NH"), ((32, 32, 163), "TET_NH"), ((26, 26, 123, 158, 158), "MIXED"))
if os.environ.get("WALK_AB_SCENES"): scenes = tuple(scenes[int(i)] for i in os.environ["WALK_AB_SCENES"].split(","))
variants = [("off", {"ADMM_HIP_WALK": "0"})]
for extra in sys.argv[1:]:
    kv = dict(p.split("=") for p in extra.split(","))
    variants.append((extra, dict({"ADMM_HIP_WALK": "-1"}, **kv)))
keys = ("ADMM_HIP_WALK", "ADMM_HIP_LEAF", "ADMM_HIP_MERGE", "ADMM_HIP_MERGE_DEPTH", "ADMM_HIP_MERGE_SMALL", "ADMM_HIP_ROOT_DEPTH", "ADMM_HIP_TREE_SEARCH", "ADMM_HIP_WALK_MAX_ENTRIES", "ADMM_HIP_WALK_MAX_NODES")
for dims, kind in scenes:
    for name, env in variants:
        for k in keys: os.environ.pop(k, None)
        os.environ.update(env)
        if kind == "MIXED": s, _ = pkg.make_mixed_system(*dims)      # configs[4]: NH + StVK tets, cloth triangles, hinges, anchors
        else: s = pkg.make_bar_system(*dims, kind=pkg.KIND[kind])
        s.keep_z(False); s.initialize()
        inf = s.info()
        n = inf["n_nodes"]
        rng = np.random.default_rng(1)
        b = rng.standard_normal((n, 3))
        x = s.solve_only(b); x2 = s.solve_only(b)
        res = np.abs(np.asarray(s.apply_A(x)).reshape(-1) - b.reshape(-1)).max() / np.abs(b).max()
        for _ in range(3): s.step(20)
        s.sync()
        t = time.perf_counter()
        for _ in range(5): s.step(20)
        s.sync(); t = (time.perf_counter() - t) / 100
        s.enable_timing(1)
        ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
        for _ in range(2):
            s.step(20); tm = s.timing()
            for k in ph: ph[k] += tm[k] / 40.0
        print("%-14s %-9s %-28s nodes %6d levels %2d walkers %4d x %d levels  resid %.1e  repeat %s  wall %.4f  fwd %.4f bwd %.4f local %.4f" % (
            "x".join(map(str, dims)), kind, name, n, inf["n_levels"], inf.get("walk_subtrees", 0), inf.get("walk_levels", 0), res, bool((np.asarray(x) == np.asarray(x2)).all()), 1e3 * t,
            ph["solve_fwd_ms"], ph["solve_bwd_ms"], ph["local_ms"]), flush=True)
        del s
