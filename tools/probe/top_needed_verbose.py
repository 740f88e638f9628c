import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ADMM_HIP_DENSE_MAX"]="0"; os.environ.setdefault("ADMM_HIP_LEAF", "16"); os.environ["ADMM_HIP_VERBOSE"]="1"
from __graft_entry__ import load_package
pkg = load_package()
import json
for world, dims in json.loads(sys.argv[1]) if len(sys.argv) > 1 else ((8, (5,4,30)), (8, (32,32,163)), (4, (32,32,163))):
    for r in (0, world//2, world-1):
        s = pkg.make_bar_system(*dims, kind=pkg.KIND["TET_STVK"], rank=r, world=world)
        s.set_shard_mode("subtree"); s.set_allreduce(lambda *a: None); s.initialize()
