import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
for dims, kind in (((13, 13, 50), "TET_STVK"), ((10, 10, 30), "TET_NH"), ((20, 20, 60), "TET_NH")):
    s = pkg.make_bar_system(*dims, kind=pkg.KIND[kind]); s.initialize()
    for _ in range(3): s.step(20)
    s.sync()
    t = time.perf_counter()
    for _ in range(5): s.step(20)
    s.sync(); t = (time.perf_counter() - t) / 100
    s.enable_timing(1)
    ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
    for _ in range(2):
        s.step(20); tm = s.timing()
        for k in ph: ph[k] += tm[k] / 40.0
    inf = s.info()
    print(dims, kind, "nodes", inf["n_nodes"], "levels", inf["n_levels"], "nnzL", inf["nnz_L"], "supernodes", inf["n_supernodes"], "wall %.4f ms/iter" % (1e3 * t), {k: round(v, 4) for k, v in ph.items()}, flush=True)
