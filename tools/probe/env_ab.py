"""A/B of one library knob on small and mid-size bars: wall us per ADMM iteration, two systems alive, frames alternated.
usage: python tools/probe/env_ab.py VAR value_a value_b      (e.g. ADMM_HIP_FRAME_GRAPH 0 1; the knob is read when a system is created)"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
var, va, vb = sys.argv[1:4]
for dims, kind in (((10, 10, 9), "TET_NH"), ((13, 13, 50), "TET_STVK"), ((20, 20, 60), "TET_NH"), ((24, 24, 75), "TET_NH")):
    sims = {}
    for v in (va, vb):
        os.environ[var] = v
        s = pkg.make_bar_system(*dims, kind=pkg.KIND[kind]); s.keep_z(False); s.initialize()
        for _ in range(3): s.step(20)
        s.sync(); sims[v] = s
    res = {va: [], vb: []}
    for rep in range(4):
        for v in (va, vb):
            s = sims[v]; t = time.perf_counter()
            for _ in range(3): s.step(20)
            s.sync(); res[v].append(1e6 * (time.perf_counter() - t) / 60)
    same = bool((sims[va].m_x == sims[vb].m_x).all())
    print("x".join(map(str, dims)), kind, "%s=%s" % (var, va), ["%.1f" % v for v in res[va]], "%s=%s" % (var, vb), ["%.1f" % v for v in res[vb]], "bitwise equal:", same, flush=True)
