"""GPU box: device memory left behind per context, with and without an in-library RCCL communicator (1 rank)."""
import os, sys, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
buf = torch.ones(4096, dtype=torch.float64, device="cuda")

def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]

def cyc(steps):
    s = pkg.make_bar_system(4, 4, 8)
    for st in steps:
        if st == "A": s.rccl_init(s.rccl_unique_id(), 0, 1); s.debug_allreduce(buf.data_ptr(), buf.numel())
        if st == "N": s.set_rccl_comm(None)
        if st == "F":
            try: s.debug_allreduce(buf.data_ptr(), buf.numel())
            except pkg.AdmmHipError: pass
        if st == "E": s.initialize(); s.step(3); s.sync()
        if st == "P": s.rccl_async_error()
    del s; gc.collect()

for steps in ("A", "AA", "AN", "ANF", "ANFA", "AE", "AAE", "AANFAE", "APAPNFAEP"):
    cyc(steps)
    f0 = free(); d = []
    for _ in range(3):
        cyc(steps); d.append(f0 - free())
    print("%-12s cumulative bytes held after 1..3 more cycles: %s" % (steps, d))
