"""One graph launch per frame vs one per ADMM iteration (ADMM_HIP_FRAME_GRAPH=1 / 0): wall us per ADMM iteration, alternated."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
for dims, kind in (((10, 10, 9), "TET_NH"), ((13, 13, 50), "TET_STVK"), ((20, 20, 60), "TET_NH"), ((24, 24, 75), "TET_NH")):
    sims = {}
    for fg in ("0", "1"):
        os.environ["ADMM_HIP_FRAME_GRAPH"] = fg
        s = pkg.make_bar_system(*dims, kind=pkg.KIND[kind]); s.keep_z(False); s.initialize()
        for _ in range(3): s.step(20)
        s.sync(); sims[fg] = s
    res = {"0": [], "1": []}
    for rep in range(4):
        for fg in ("0", "1"):
            s = sims[fg]; t = time.perf_counter()
            for _ in range(3): s.step(20)
            s.sync(); res[fg].append(1e6 * (time.perf_counter() - t) / 60)
    print("x".join(map(str, dims)), kind, "per-iteration graph", ["%.1f" % v for v in res["0"]], "frame graph", ["%.1f" % v for v in res["1"]], flush=True)
