import os, sys, time, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import checkers
from __graft_entry__ import load_package
pkg = load_package()
dims = (16, 16, 65)
x, t = pkg.meshgen.bar(*dims); m = pkg.meshgen.lumped_tet_mass(x, t, 1000.0)
s = checkers.Ref(); s.settings(0.04, 20); s.add_nodes(x.ravel(), np.repeat(m, 3)); s.add_forces(pkg.KIND["TET_NH"], t, [1e5, 1e5, 5])
s.add_forces(pkg.KIND["ANCHOR"], pkg.meshgen.bar_anchor_nodes(16, 16), [-1.0, 1.0]); s.add_gravity([0, -9.8, 0])
assert s.initialize()
s.time_steps(1)
sec = s.time_steps(2)
print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "threads", checkers.Ref.load().ref_omp_threads(), "ms/iter %.1f" % (1e3 * sec / 40), flush=True)
