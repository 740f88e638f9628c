"""Probe (build container, CPU only): can the 20-evaluation line searches of the Neo-Hookean prox be predicted from what is
known BEFORE the search (initial gradient, objective value, warm start)?  Runs a bar through a traced build of the oracle
(-DORC_LS_TRACE) with one ADMM iteration per orc_step call replaced by reading the trace after every frame -- the trace holds
the LAST iteration of a frame, which is the regime where the slow searches live.

usage: python tools/probe/ls_predict.py [nx ny nz frames]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checkers
from __graft_entry__ import load_package

pkg = load_package()
nx, ny, nz, frames = (int(a) for a in (sys.argv[1:5] + ["8", "8", "40", "6"][len(sys.argv) - 1:]))
so = os.path.join(ROOT, "tools", "probe", "_build", "liboracle_trace.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
os.system("gcc -std=c99 -O2 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math -DORC_LS_TRACE -w -o %s %s -lm"
          % (so, os.path.join(ROOT, "oracle", "admm_oracle.c")))
checkers.ORACLE_DIR_SAVED = checkers.ORACLE_DIR
_real = os.path.join
# load the traced build instead of liboracle.so
checkers.Oracle.lib = None
orig = C.CDLL
C.CDLL = lambda p, *a, **k: orig(so if p.endswith("liboracle.so") else p, *a, **k)
checkers.Oracle.load()
C.CDLL = orig
lib = checkers.Oracle.lib

x, tets = pkg.meshgen.bar(nx, ny, nz)
m = pkg.meshgen.lumped_tet_mass(x, tets, 1000.0)
iters = int(os.environ.get("ITERS", "20"))
o = checkers.Oracle()
o.settings(0.04, iters)
o.add_nodes(x.ravel(), np.repeat(m, 3))
o.add_forces(pkg.KIND["TET_NH"], tets, [1e5, 1e5, 5])
o.add_forces(pkg.KIND["ANCHOR"], pkg.meshgen.bar_anchor_nodes(nx, ny), [-1.0, 1.0])
o.add_gravity([0, -9.8, 0])
assert o.initialize()
nt = len(tets)
buf = np.zeros((iters, nt, 6))
lib.orc_set_ls_trace.argtypes = [C.c_void_p, C.c_size_t]
lib.orc_set_ls_trace(buf.ctypes.data, nt * 6)
thr = float(os.environ.get("THR", "5e-3"))
for fr in range(frames):
    o.step()
    g0, f0, nfev = buf[:, :, 0], buf[:, :, 1], buf[:, :, 2] - 1          # the trace's own value() call not counted
    slow = nfev >= 15
    pred = g0 < thr
    print("frame %d: slow %.1f %% of (tet, iteration) pairs; evaluations mean %.2f; predicted slow %.1f %%, missed %.2f %%, false alarms %.2f %%" % (
        fr, 100 * slow.mean(), nfev.mean(), 100 * pred.mean(), 100 * (slow & ~pred).mean(), 100 * (~slow & pred).mean()))
    print("   slow %% per iteration:", " ".join("%.0f" % (100 * slow[i].mean()) for i in range(iters)))
    print("   missed %% per iteration:", " ".join("%.1f" % (100 * (slow[i] & ~pred[i]).mean()) for i in range(iters)))
    # waves: 64 consecutive tets
    nw = nt // 64
    def wave_cost(mask_sel):          # mean over waves of the per-wave maximum among selected lanes (0 if none)
        tot = 0.0
        for i in range(iters):
            ev = np.where(mask_sel[i], nfev[i], 0)[: nw * 64].reshape(nw, 64)
            tot += ev.max(axis=1).sum()
        return tot / (iters * nw)
    all_cost = wave_cost(np.ones_like(slow))
    fast_cost = wave_cost(~pred)                                  # first kernel: predicted-fast lanes only
    # second kernel: predicted-slow tets compacted in order
    sec = 0.0
    for i in range(iters):
        ev = nfev[i][pred[i]]
        k = (len(ev) + 63) // 64
        ev = np.concatenate([ev, np.zeros(k * 64 - len(ev))]).reshape(k, 64)
        sec += ev.max(axis=1).sum()
    sec /= iters * nw
    print("   wave-evaluations per wave: one kernel %.2f; split: fast kernel %.2f + compacted kernel %.2f (per original wave) = %.2f; ideal (mean) %.2f" % (
        all_cost, fast_cost, sec, fast_cost + sec, nfev.mean()))
