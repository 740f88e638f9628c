#!/usr/bin/env python3
"""A/B harness (GPU box): builds variants of libadmm_hip.so with extra hipcc
flags and times the phases of a frame on a mid-size bar, one process per variant.

  python tools/ab_local.py "name=-DADMM_TET_WAVES=4" "name2=-DADMM_TET_WAVES=5 -mllvm -disable-promote-alloca-to-vector"
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
dims = %r
s = pkg.make_bar_system(*dims, device_id=0)
s.initialize()
for _ in range(2): s.step(20)
s.enable_timing(True)
acc = {}
for _ in range(5):
    s.step(20)
    t = s.timing()
    for k, v in t.items(): acc[k] = acc.get(k, 0.0) + v
n = 5 * 20
print(json.dumps({k: (v / n if k.endswith("_ms") else v) for k, v in acc.items()}), float(np.abs(s.m_x).sum()))
'''


def main():
    from __graft_entry__ import load_package
    pkg = load_package()
    dims = (32, 32, 40)
    variants = [("default", None)]
    for a in sys.argv[1:]:
        if a.startswith("dims="):
            dims = tuple(int(v) for v in a[5:].split("x"))
            continue
        name, flags = a.split("=", 1)
        variants.append((name, flags.split()))
    for name, flags in variants:
        env = dict(os.environ)
        if flags is not None:
            out = os.path.join(ROOT, "admm-elastic-sca_amd", "_build", "libadmm_hip_%s.so" % name)
            pkg._build.build(force=False, extra_hip_flags=flags, out=out, tag="_" + name)
            env["ADMM_HIP_LIB"] = out
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, dims)], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
        print("%-28s %s" % (name, line))


if __name__ == "__main__":
    main()
