#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / LDS of every kernel in libadmm_hip.so's device code (hipcc -S of admm_hip.hip with the build's flags).
usage: kernel_resources.py [substring] [extra hipcc flags ...]   -> also leaves the ISA in /tmp/isa/admm_hip.s"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "admm-elastic-sca_amd", "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
sub = sys.argv[1] if len(sys.argv) > 1 else ""
extra = sys.argv[2:]
os.makedirs("/tmp/isa", exist_ok=True)
out = "/tmp/isa/admm_hip.s"
src = os.path.join(ROOT, "admm-elastic-sca_amd", "csrc", "admm_hip.hip")
if len(sys.argv) > 2 and sys.argv[2].endswith(".hip"):
    src, extra = sys.argv[2], sys.argv[3:]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.HIP_FLAGS + extra + ["-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
txt = open(out).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", txt, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if sub not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s lds %5s spill_v %s" % (dem[:90], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count")))
