
import os, sys, json
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.path.join('/root/repo', "tests"))
import torch
from __graft_entry__ import load_package
pkg = load_package()
assert torch.cuda.is_available()
torch.cuda.set_device(0)

def cycle():
    s = pkg.make_bar_system(4, 4, 8)
    assert s.rccl_async_error() == 0                      # no communicator: healthy by definition
    uid = s.rccl_unique_id()
    assert uid.any()
    s.rccl_init(uid, 0, 1)
    t = torch.arange(4096, dtype=torch.float64, device="cuda") * 0.37 - 5.0
    want = t.clone()
    torch.cuda.synchronize()
    for _ in range(3):
        s.debug_allreduce(t.data_ptr(), t.numel())         # one rank: the sum is the value itself, bit for bit
    assert torch.equal(t, want)
    assert s.rccl_async_error() == 0
    s.rccl_init(s.rccl_unique_id(), 0, 1)                  # a second communicator replaces (and destroys) the first
    s.debug_allreduce(t.data_ptr(), t.numel())
    assert torch.equal(t, want)
    s.set_rccl_comm(None)                                  # back to "no transport": the all-reduce must now refuse
    try:
        s.debug_allreduce(t.data_ptr(), t.numel())
        raise SystemExit("all-reduce without a transport did not fail")
    except pkg.AdmmHipError as e:
        assert "neither an RCCL communicator nor an all-reduce hook" in str(e), e
    s.rccl_init(s.rccl_unique_id(), 0, 1)
    s.initialize(); s.step(3); s.sync()                    # world 1: no collective in the loop, but the per-frame poll runs
    assert np.isfinite(s.m_x).all() and s.rccl_async_error() == 0
    del s                                                  # admm_hip_destroy -> comm_release -> ncclCommDestroy

cycle()
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
import gc
for _ in range(4):
    cycle(); torch.cuda.synchronize(); a=torch.cuda.mem_get_info()[0]; gc.collect(); torch.cuda.synchronize(); print('after cycle: held', free0-a, 'after gc', free0-torch.cuda.mem_get_info()[0], 'torch reserved', torch.cuda.memory_reserved())
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
leaked = free0 - free1
print("LIFECYCLE ok; device memory after 4 more cycles: %+d bytes" % (-leaked))
assert leaked < (64 << 20), leaked                         # RCCL keeps some process-wide state; a communicator's buffers must not pile up
