import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
for name, mk in (("bar1M", lambda: (pkg.make_bar_system(32, 32, 163), None)), ("mixed", lambda: pkg.make_mixed_system(26, 26, 123, 158, 158))):
    s = mk()[0]; s.initialize()
    x0 = s.m_x.copy()
    t = time.time()
    for f in range(150):
        s.step(20)
        if f % 50 == 49:
            x = s.m_x
            assert np.isfinite(x).all(), (name, f)
            print(name, "frame", f + 1, "max |x - x0| %.3f" % np.abs(x - x0).max(), "max |v| %.3f" % np.abs(s.m_v).max(), flush=True)
    print(name, "150 frames in %.1f s" % (time.time() - t), flush=True)
    del s
