import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
for world in (2, 4, 8):
    for rank in range(min(world, 2)):
        s = pkg.make_bar_system(32, 32, 163, rank=rank, world=world, shard_mode="subtree")
        s.set_allreduce(lambda p, c, st: 0)
        s.keep_z(False)
        s.initialize()
        inf = s.info()
        for _ in range(2): s.step(20)
        s.enable_timing(1)
        ph = dict(local_ms=0.0, total_ms=0.0)
        for _ in range(2):
            s.step(20); tm = s.timing()
            for k in ph: ph[k] += tm[k] / 40
        print("world %d rank %d: local elements %d of %d (%.1f %%), local %.3f ms, iteration %.3f ms" % (world, rank, inf["n_elems_local"], inf["n_elems_total"], 100.0 * inf["n_elems_local"] / inf["n_elems_total"], ph["local_ms"], ph["total_ms"]), flush=True)
        del s
