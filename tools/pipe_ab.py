#!/usr/bin/env python3
"""GPU box: A/B of the pipelined-groups mode (ADMM_HIP_PIPE) against the plain launch on the 1M-tet bar (or --dims), wall clock
per ADMM iteration without timing events, alternated `--rounds` times on one box.
  python tools/pipe_ab.py [--dims 32 32 163] [--frames 5] [--rounds 2] [--configs "PIPE=2" "PIPE=2,PIPE_CHAIN=0" ...]
A config is a comma-separated list of ADMM_HIP_<KEY>=<value> settings; "base" = none."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
p = argparse.ArgumentParser()
p.add_argument("--dims", type=int, nargs=3, default=[32, 32, 163])
p.add_argument("--frames", type=int, default=5)
p.add_argument("--rounds", type=int, default=2)
p.add_argument("--kind", default="TET_NH")
p.add_argument("--configs", nargs="*", default=["base", "PIPE=2", "PIPE=2,PIPE_CHAIN=0", "PIPE=3", "PIPE=4"])
a = p.parse_args()
KEYS = set()
for c in a.configs:
    if c != "base":
        for kv in c.split(","):
            KEYS.add("ADMM_HIP_" + kv.split("=")[0])
sims = {}
for c in a.configs:
    for k in KEYS: os.environ.pop(k, None)
    if c != "base":
        for kv in c.split(","):
            k, v = kv.split("="); os.environ["ADMM_HIP_" + k] = v
    s = pkg.make_bar_system(*a.dims, kind=pkg.KIND[a.kind]); s.initialize()
    for _ in range(2): s.step(20)
    s.sync(); sims[c] = s
for r in range(a.rounds):
    for c in a.configs:
        s = sims[c]
        t = time.perf_counter()
        for _ in range(a.frames): s.step(20)
        s.sync()
        t = (time.perf_counter() - t) / (a.frames * 20)
        print("round %d  %-40s %.4f ms/iter   checksum %.12e" % (r, c, 1e3 * t, float(np.abs(s.m_x).sum())), flush=True)
