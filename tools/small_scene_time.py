import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
for dims in ((6, 6, 12), (13, 13, 50)):
    s = pkg.make_bar_system(*dims, device_id=0)
    s.initialize()
    for _ in range(3): s.step(20)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(20): s.step(20)
    s.sync()
    dt = time.perf_counter() - t0
    print("graph", os.environ.get("ADMM_HIP_GRAPH", "1"), dims, "tets", s.n_tets, "us/iter %.1f" % (dt / 400 * 1e6))
