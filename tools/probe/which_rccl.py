import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t)
from __graft_entry__ import load_package
pkg = load_package()
s = pkg.make_bar_system(4, 4, 8)
uid = s.rccl_unique_id()
s.rccl_init(uid, 0, 1)
print([l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l and "r-xp" in l])
dist.destroy_process_group()
