#!/usr/bin/env python3
"""Probe (GPU box): when do the waves of ONE tet-kernel launch start and end?  (-DADMM_TET_TIMELINE variant build: the first lane
of every wave stamps the 100 MHz real-time counter at its start and its end.)  Prints the launch's span, how many waves are
alive over time, and what the span would be if the waves that are still running at the end had started first.

  python tools/probe/tet_timeline.py [dims=32x32x163]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(*%r, device_id=0)
s.initialize()
lib = pkg.lib()
nw = (s.n_tets + 63) // 64
lib.admm_hip_debug_tet_wave_times.argtypes = [ctypes.c_long, ctypes.c_void_p]
for _ in range(2): s.step(20)
s.m_x
assert lib.admm_hip_debug_tet_wave_times(nw, None) == 0
s.step(20); s.m_x                       # the buffer holds the LAST iteration's launch
buf = np.zeros((nw, 2), dtype=np.uint64)
assert lib.admm_hip_debug_tet_wave_times(nw, buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
ok = t[:, 1] > 0
t = t[ok]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
dur = en - st
span = en.max()
print("%%d waves; launch span %%.1f us; wave duration mean %%.1f, median %%.1f, p90 %%.1f, p99 %%.1f, max %%.1f us; sum of durations / 2048 slots = %%.1f us" %% (
    len(t), span, dur.mean(), np.median(dur), np.quantile(dur, 0.9), np.quantile(dur, 0.99), dur.max(), dur.sum() / 2048))
grid = np.linspace(0, span, 21)
alive = [(int(((st <= g) & (en > g)).sum())) for g in grid]
print("waves alive at 0, 5, ... 100 %%%% of the span: " + " ".join(str(a) for a in alive))
print("last wave start at %%.1f us; waves that start in the last 25 %%%% of the span: %%d, their mean duration %%.1f us" %% (st.max(), int((st > 0.75 * span).sum()), dur[st > 0.75 * span].mean() if (st > 0.75 * span).any() else 0.0))
# greedy list schedule on 2048 slots, longest first (what a launch order by last iteration's cost would approach)
import heapq
for name, order in (("as launched", np.arange(len(dur))), ("longest first", np.argsort(-dur))):
    slots = [0.0] * 2048
    heapq.heapify(slots)
    end = 0.0
    for i in order:
        s0 = heapq.heappop(slots); e0 = s0 + dur[i]; end = max(end, e0); heapq.heappush(slots, e0)
    print("list schedule of the measured durations on 2048 slots, %%s: %%.1f us" %% (name, end))
'''


def main():
    dims = (32, 32, 163)
    for a in sys.argv[1:]:
        if a.startswith("dims="):
            dims = tuple(int(v) for v in a[5:].split("x"))
    from __graft_entry__ import load_package
    pkg = load_package()
    out = os.path.join(ROOT, "admm-elastic-sca_amd", "_build", "libadmm_hip_tl.so")
    pkg._build.build(force=False, extra_hip_flags=["-DADMM_TET_TIMELINE"], out=out, tag="_tl")
    env = dict(os.environ, ADMM_HIP_LIB=out)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, dims)], env=env, capture_output=True, text=True)
    print(r.stdout)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)


if __name__ == "__main__":
    main()
