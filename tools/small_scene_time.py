"""GPU box: microseconds per ADMM iteration of the small / mid-size throughput configurations:
synthetic NH bar 10x10x9 = 5,400 tets and StVK bar 13x13x50 = 50,700 tets, 20 iterations per frame; plus a 2,592-tet bar."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
for dims, kind in (((6, 6, 12), "TET_NH"), ((10, 10, 9), "TET_NH"), ((13, 13, 50), "TET_STVK"), ((13, 13, 50), "TET_NH")):
    s = pkg.make_bar_system(*dims, kind=pkg.KIND[kind], device_id=0)
    s.initialize()
    for _ in range(3):
        s.step(20)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        s.step(20)
    s.sync()
    dt = time.perf_counter() - t0
    inf = s.info()
    print(kind, dims, "tets", s.n_tets, "nodes", inf["n_nodes"], "dense" if inf["dense_solve"] else "levels %d" % inf["n_levels"],
          "us/iter %.1f" % (dt / 400 * 1e6), "iters/s x tets %.3g" % (400 / dt * s.n_tets))
