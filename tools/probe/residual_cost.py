#!/usr/bin/env python3
"""Probe (GPU box): what residual tracking costs per ADMM iteration at the 1M-tet bar -- off / on (fused into the tet kernel) /
on with the unfused passes (ADMM_HIP_RES_UNFUSED=1: snapshot copies + primal / dual passes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
dims = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [32, 32, 163]
res = {}
sims = {}
for name, env, on in (("off", None, False), ("on, fused", None, True), ("on, unfused passes", "1", True)):
    if env: os.environ["ADMM_HIP_RES_UNFUSED"] = env
    else: os.environ.pop("ADMM_HIP_RES_UNFUSED", None)
    s = pkg.make_bar_system(*dims); s.initialize()
    if on: s.enable_residuals(True)
    for _ in range(2): s.step(20)
    s.sync(); sims[name] = s
for r in range(2):
    for name, s in sims.items():
        t = time.perf_counter()
        for _ in range(4): s.step(20)
        s.sync(); t = (time.perf_counter() - t) / 80
        extra = ""
        if name != "off":
            rr, ss, _n = s.residuals()
            extra = "  |r| %.6e |s| %.6e (last iteration)" % (rr[-1], ss[-1])
        print("round %d  %-22s %.4f ms/iter%s" % (r, name, 1e3 * t, extra), flush=True)
for name, s in sims.items():      # where the extra time sits: phase events around every iteration (the residual launches fall into rhs_ms)
    s.enable_timing(1)
    ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
    for _ in range(2):
        s.step(20); tm = s.timing()
        for k in ph: ph[k] += tm[k] / 40.0
    print("%-22s %s" % (name, {k: round(v, 4) for k, v in ph.items()}), flush=True)
