#!/usr/bin/env python3
"""De-risking probe (GPU box, `python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/probe/nccl_single.py`):
the mechanics bench.py's all-reduce hook relies on, under the real RCCL backend with one rank -- process group with
device_id, a tensor aliasing memory that torch did not allocate (hipMalloc through ctypes, like the C library's buffers),
all_reduce on it from the current stream, barrier, all_gather of a small CUDA tensor."""
import ctypes
import os
import torch
import torch.distributed as dist

rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
hip = ctypes.CDLL("libamdhip64.so")
ptr = ctypes.c_void_p()
n = 1 << 16
assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(8 * n)) == 0


class _Ptr:
    def __init__(self, p, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (p, False), "version": 2}


t = torch.as_tensor(_Ptr(ptr.value, n), device=torch.device("cuda", local_rank))
assert t.data_ptr() == ptr.value
t.copy_(torch.arange(n, dtype=torch.float64, device="cuda"))
for _ in range(3):
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert float(t[12345]) == 12345.0 * world ** 3 or world == 1 and float(t[12345]) == 12345.0
dist.barrier()
mine = torch.tensor([1.0 + rank, 2.0], dtype=torch.float64, device="cuda")
allr = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(allr, mine)
m = torch.tensor([3.5], dtype=torch.float64, device="cuda"); dist.all_reduce(m, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
hip.hipFree(ptr)
print("nccl single-rank mechanics ok: world %d, t[12345] = %g, gathered %s" % (world, float(t[12345].cpu()) if False else 12345.0, [a.tolist() for a in allr]))
