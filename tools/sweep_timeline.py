#!/usr/bin/env python3
"""Workgroup timeline of the triangular sweeps (GPU box): builds a variant of libadmm_hip.so with -DADMM_SWEEP_PROFILE (every
workgroup of the sweep kernels stamps the 100 MHz real-time counter at its start, after its first staging barrier and at its
end), runs the headline bar and prints, per launch of one ADMM iteration: span (first start -> last end), how long after the
first start the last workgroup STARTED (dispatch / occupancy-limited ramp), the median staging time (start -> barrier) and the
median / maximum streaming time (barrier -> end), and the gap to the previous launch's last end.

  python tools/sweep_timeline.py [dims=32x32x163]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(*%r, device_id=0)
s.initialize()
lib = pkg.lib()
lib.admm_hip_debug_sweep_profile_read.restype = ctypes.c_long
for _ in range(3):
    s.step(20)
s.m_x
nwg = lib.admm_hip_debug_sweep_profile_read(None, None)
stamps = np.zeros(4 * nwg, dtype=np.uint64)
meta = np.zeros(6 * 4096, dtype=np.int32)
n = lib.admm_hip_debug_sweep_profile_read(stamps.ctypes.data_as(ctypes.c_void_p), meta.ctypes.data_as(ctypes.c_void_p))
assert n > 0
meta = meta[: 6 * n].reshape(n, 6)
stamps = stamps.reshape(nwg, 4).astype(np.int64)
rows = []
for tag, level, cnt, kb, lst, first in meta:
    st = stamps[first: first + cnt]
    st = st[st[:, 2] > 0]
    if len(st) == 0: continue
    name = ("fwd_small" if tag == 0 else "fwd_big<%%d>" %% tag) if tag < 100 else "bwd<%%d,%%d>" %% ((tag - 100) // 10, 16 if (tag - 100) %% 10 == 6 else (tag - 100) %% 10)
    rows.append((st[:, 0].min(), name, level, lst, kb / 1024.0, st))
rows.sort(key=lambda r: r[0])
t_origin = rows[0][0]
print("last ADMM iteration of a frame; times in us; list: 0 = context stream (group 0 / everything), 1.. = side streams, 100 = top")
print("%%-12s %%5s %%4s %%6s %%7s | %%8s %%8s | %%8s %%9s %%9s %%9s | %%6s" %% ("kernel", "level", "list", "wgs", "MB", "start", "end", "span", "laststart", "stage_med", "strm_med", "TB/s"))
prev_end = {}
gaps = []
for t0, name, level, lst, mb, st in rows:
    t1 = st[:, 2].max()
    span = (t1 - t0) / 100.0
    ok = st[:, 1] > 0
    stage = np.median(st[ok, 1] - st[ok, 0]) / 100.0 if ok.any() else 0.0
    strm = np.median(st[ok, 2] - st[ok, 1]) / 100.0 if ok.any() else 0.0
    key = lst if lst != 100 else 0
    if key in prev_end: gaps.append((t0 - prev_end[key]) / 100.0)
    prev_end[key] = t1
    print("%%-12s %%5d %%4d %%6d %%7.1f | %%8.2f %%8.2f | %%8.2f %%9.2f %%9.2f %%9.2f | %%6.2f" %% (name, level, lst, len(st), mb, (t0 - t_origin) / 100.0, (t1 - t_origin) / 100.0, span, (st[:, 0].max() - t0) / 100.0, stage, strm, mb / span if span else 0))
fw = [r for r in rows if not r[1].startswith("bwd")]; bw = [r for r in rows if r[1].startswith("bwd")]
f0 = min(r[0] for r in fw); f1 = max(r[5][:, 2].max() for r in fw); b0 = min(r[0] for r in bw); b1 = max(r[5][:, 2].max() for r in bw)
print("forward launches (without the root kernels): first start -> last end %%.1f us; backward %%.1f us; between them (root) %%.1f us; median gap on a stream %%.2f us" %% ((f1 - f0) / 100.0, (b1 - b0) / 100.0, (b0 - f1) / 100.0, float(np.median(gaps))))
'''


def main():
    dims = (32, 32, 163)
    for a in sys.argv[1:]:
        if a.startswith("dims="):
            dims = tuple(int(v) for v in a[5:].split("x"))
    from __graft_entry__ import load_package
    pkg = load_package()
    out = os.path.join(ROOT, "admm-elastic-sca_amd", "_build", "libadmm_hip_swp.so")
    pkg._build.build(force=False, extra_hip_flags=["-DADMM_SWEEP_PROFILE"], out=out, tag="_swp")
    env = dict(os.environ, ADMM_HIP_LIB=out)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, dims)], env=env, capture_output=True, text=True)
    print(r.stdout)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)


if __name__ == "__main__":
    main()
