"""What a per-frame admm_hip_update_anchors costs on a small scene (5 400 tets): frames with and without the update call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
mg = pkg.meshgen
x, t = mg.bar(10, 10, 9)
m = mg.lumped_tet_mass(x, t, 1000.0)
s = pkg.System(device_id=0); s.set_timestep(0.04)
s.add_nodes(x.ravel(), np.repeat(m, 3))
s.add_forces(pkg.KIND["TET_NH"], t, [1e5, 1e5, 5])
s.add_forces(pkg.KIND["ANCHOR"], mg.bar_anchor_nodes(10, 10), [-1.0, 1.0])
tips = np.arange(x.shape[0] - 20, x.shape[0]).astype(np.int32)
b = s.add_forces(pkg.KIND["ANCHOR"], tips, [-1.0, 1.0], targets=x[tips])
s.add_gravity([0, -9.8, 0]); s.keep_z(False); s.initialize()
for _ in range(5): s.step(20)
s.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for f in range(40): s.step(20)
    s.sync(); ta = (time.perf_counter() - t0) / 40
    t0 = time.perf_counter()
    for f in range(40):
        s.update_anchors(b, targets=x[tips] + [0.001 * f, 0, 0], active=np.ones(tips.size, np.int32)); s.step(20)
    s.sync(); tb = (time.perf_counter() - t0) / 40
    print("frame %.3f ms, with a moving-anchor update per frame %.3f ms: +%.1f us" % (1e3 * ta, 1e3 * tb, 1e6 * (tb - ta)), flush=True)
