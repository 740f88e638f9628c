"""The mixed scene of BASELINE configs[4] on one GPU: A/B of one library knob, phase times per ADMM iteration and bitwise comparison of the frames.
usage: python tools/probe/mixed_ab.py VAR value_a value_b"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
var, va, vb = sys.argv[1:4]
sims = {}
for v in (va, vb):
    os.environ[var] = v
    s, _ = pkg.make_mixed_system(26, 26, 123, 158, 158); s.keep_z(False); s.initialize()
    for _ in range(2): s.step(20)
    s.sync(); sims[v] = s
res = {va: [], vb: []}
for rep in range(3):
    for v in (va, vb):
        s = sims[v]; t = time.perf_counter()
        for _ in range(3): s.step(20)
        s.sync(); res[v].append(1e6 * (time.perf_counter() - t) / 60)
print("bitwise equal:", bool((sims[va].m_x == sims[vb].m_x).all()))
for v in (va, vb):
    s = sims[v]; s.enable_timing(1)
    ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
    for _ in range(2):
        s.step(20); tm = s.timing()
        for k in ph: ph[k] += tm[k] / 40.0
    print("%s=%s" % (var, v), "wall us/iter", ["%.1f" % q for q in res[v]], {k: round(q, 4) for k, q in ph.items()}, flush=True)
