#!/usr/bin/env python3
"""Probe (GPU box): the class API's frame boundary (upload_state + step + download_state on page-locked vectors) against the resident
loop at the 1M-tet bar, with everything on the solver's stream.  (Round 3 tried v on a second transfer stream -- a second DMA engine beside x: the four
transfers alone 0.385 instead of 0.403 ms per frame, i.e. one engine already moves ~43 GB/s of the link's 64, but around the step the
frame was SLOWER (+3.5-4 % against +1.3-1.6 % over the resident loop): the extra stream's events sit in the prologue's way.  Not kept;
that variant is not in the library any more.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
sims = {}
for name in ("class API",):
    dims = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 32, 163)
    trace = len(sys.argv) > 4 and sys.argv[4] == "trace"      # under rocprofv3 (tools/class_api_timeline.py): three class-API frames, nothing else
    s = pkg.make_bar_system(*dims); s.keep_z(False); s.initialize()
    s.step(20); s.sync()
    hx = s.m_x.copy(); hv = s.m_v.copy(); s.pin_host(hx); s.pin_host(hv)
    sims[name] = (s, hx, hv)
N = 6 if len(sys.argv) < 4 or (len(sys.argv) > 4 and sys.argv[4] == "trace") else 40
if len(sys.argv) > 4 and sys.argv[4] == "trace":
    s, hx, hv = sims["class API"]
    for _ in range(3): s.upload_state(hx, hv); s.step(20); s.download_state(hx, hv)
    sys.exit(0)
for r in range(3):
    for name, (s, hx, hv) in sims.items():
        t = time.perf_counter()
        for _ in range(N): s.upload_state(hx, hv); s.step(20); s.download_state(hx, hv)
        tc = (time.perf_counter() - t) / N
        t = time.perf_counter()
        for _ in range(N): s.step(20)
        s.sync(); tr = (time.perf_counter() - t) / N
        t = time.perf_counter()
        for _ in range(N): s.upload_state(hx, hv); s.download_state(hx, hv)
        tx = (time.perf_counter() - t) / N
        print("round %d  %-12s class API %.3f ms/frame, resident %.3f: +%.2f %%; the four transfers alone %.3f ms" % (r, name, 1e3 * tc, 1e3 * tr, 100 * (tc / tr - 1), 1e3 * tx), flush=True)
