#!/usr/bin/env python3
"""Probe (GPU box): what bounds an UNDER-FILLED tet launch (fewer waves than the chip has wave slots)?

  python tools/probe/underfilled.py timeline  [dims=13x13x50] [kind=TET_STVK] [frames=14] [tpb=64]
      -DADMM_TET_TIMELINE build: every wave of the last iteration's launch with its duration, its largest per-lane count of
      line-search evaluations and L-BFGS iterations -> span, slowest waves, least-squares fit duration = a + b * evaluations
  python tools/probe/underfilled.py tpb [dims=...] [kind=...] [frames=14] [lib=path]
      shipped library: local step (HIP events) and wall per ADMM iteration for ADMM_HIP_TPB = 64 / 32 / 16 / 8 tets per one-wave block,
      each from the same fixed state (frames - 4 warm frames, then 4 timed ones)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

TIMELINE = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(*%(dims)r, kind=pkg.KIND[%(kind)r], device_id=0)
s.initialize()
lib = pkg.lib()
tpb = %(tpb)d
nw = (s.n_tets + tpb - 1) // tpb
lib.admm_hip_debug_tet_wave_times.argtypes = [ctypes.c_long, ctypes.c_void_p]
for _ in range(%(frames)d - 1): s.step(20)
s.m_x
assert lib.admm_hip_debug_tet_wave_times(nw, None) == 0
s.step(20); s.m_x                       # the buffer holds the LAST iteration's launch
buf = np.zeros((nw, 4), dtype=np.uint64)
assert lib.admm_hip_debug_tet_wave_times(nw, buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
dur, ev, it = en - st, t[:, 2], t[:, 3]
print("%%s %%s, frame %%d, last iteration, %%d tets per wave: %%d waves; launch span %%.1f us; wave duration mean %%.1f median %%.1f p90 %%.1f max %%.1f us; last wave starts at %%.1f us" %% (
    %(dims)r, %(kind)r, %(frames)d, tpb, len(t), en.max(), dur.mean(), np.median(dur), np.quantile(dur, 0.9), dur.max(), st.max()))
A = np.stack([np.ones(len(t)), ev.astype(float)], axis=1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print("   fit: wave duration = %%.1f us + %%.2f us x (largest per-lane evaluation count of the wave); residual rms %%.1f us" %% (coef[0], coef[1], np.sqrt(np.mean((A @ coef - dur) ** 2))))
print("   evaluations (wave maximum): mean %%.1f max %%d; L-BFGS iterations (wave maximum): mean %%.2f max %%d" %% (ev.mean(), ev.max(), it.mean(), it.max()))
o = np.argsort(-dur)[:8]
print("   slowest waves: " + "; ".join("%%.1f us (%%d evaluations, %%d iterations)" %% (dur[i], ev[i], it[i]) for i in o))
for lo, hi in ((0, 4), (5, 10), (11, 20), (21, 40), (41, 60), (61, 100), (101, 1000)):
    m = (ev >= lo) & (ev <= hi)
    if m.any(): print("   waves with %%3d..%%3d evaluations: %%5d, mean duration %%.1f us" %% (lo, hi, int(m.sum()), dur[m].mean()))
'''

TPB = r'''
import sys, time, numpy as np
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(*%(dims)r, kind=pkg.KIND[%(kind)r], device_id=0)
s.initialize()
for _ in range(%(frames)d - 4): s.step(20)
s.sync()
t = time.perf_counter()
for _ in range(2): s.step(20)
s.sync(); wall = (time.perf_counter() - t) / 40
s.enable_timing(1)
ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0)
for _ in range(2):
    s.step(20); tm = s.timing()
    for k in ph: ph[k] += tm[k] / 40.0
x = s.m_x
print("   TPB %%s: wall %%.1f us per iteration (frames %%d-%%d); events (frames %%d-%%d): local %%.1f rhs %%.1f fwd %%.1f bwd %%.1f us; checksum %%.17g" %% (
    %(tpbs)r, 1e3 * wall * 1e3 / 1e3, %(frames)d - 3, %(frames)d - 2, %(frames)d - 1, %(frames)d, 1e3 * ph["local_ms"], 1e3 * ph["rhs_ms"], 1e3 * ph["solve_fwd_ms"], 1e3 * ph["solve_bwd_ms"], float(np.abs(x).sum())), flush=True)
'''


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "tpb"
    dims, kind, frames, tpb, libpath = (13, 13, 50), "TET_STVK", 14, 64, None
    for a in sys.argv[2:]:
        if a.startswith("dims="): dims = tuple(int(v) for v in a[5:].split("x"))
        if a.startswith("kind="): kind = a[5:]
        if a.startswith("frames="): frames = int(a[7:])
        if a.startswith("tpb="): tpb = int(a[4:])
        if a.startswith("lib="): libpath = a[4:]
    from __graft_entry__ import load_package
    pkg = load_package()
    if mode == "timeline":
        out = os.path.join(ROOT, "admm-elastic-sca_amd", "_build", "libadmm_hip_tl.so")
        pkg._build.build(force=False, extra_hip_flags=["-DADMM_TET_TIMELINE"], out=out, tag="_tl")
        env = dict(os.environ, ADMM_HIP_LIB=out, ADMM_HIP_TPB=str(tpb), ADMM_HIP_GRAPH="0")
        r = subprocess.run([sys.executable, "-c", TIMELINE % dict(root=ROOT, dims=dims, kind=kind, frames=frames, tpb=tpb)], env=env, capture_output=True, text=True)
        print(r.stdout)
        if r.returncode: print(r.stderr[-3000:]); sys.exit(1)
        return
    print("%s %s: local step by tets per one-wave block" % (dims, kind))
    for t in ("64", "32", "16", "8"):
        env = dict(os.environ, ADMM_HIP_TPB=t)
        if libpath: env["ADMM_HIP_LIB"] = libpath
        r = subprocess.run([sys.executable, "-c", TPB % dict(root=ROOT, dims=dims, kind=kind, frames=frames, tpbs=t)], env=env, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode: print(r.stderr[-3000:]); sys.exit(1)


if __name__ == "__main__":
    main()
