#!/usr/bin/env python3
"""A/B of two BUILDS of the library on one box (GPU box): every variant is compiled with its extra hipcc flags, then the variants' child
processes alternate `reps` times; phase times per ADMM iteration from HIP events, wall time from an event-free run, checksum of x.

  python tools/probe/lib_ab.py scene=mixed|bar:32x32x163[:KIND] reps=3 "a=" "b=-DADMM_MULTI_EPL=1" ["c=ADMM_HIP_SOME_KNOB=0"]
(a variant = hipcc flags, starting with '-', and / or VAR=value settings of run-time knobs for its child process)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
pkg = load_package()
scene = %(scene)r
if scene == "mixed":
    s, _ = pkg.make_mixed_system(26, 26, 123, 158, 158)
else:
    parts = scene.split(":"); dims = tuple(int(v) for v in parts[1].split("x")); kind = parts[2] if len(parts) > 2 else "TET_NH"
    s = pkg.make_bar_system(*dims, kind=pkg.KIND[kind])
s.keep_z(False); s.initialize()
for _ in range(3): s.step(20)
s.sync(); t = time.perf_counter()
for _ in range(4): s.step(20)
s.sync(); wall = (time.perf_counter() - t) / 80
s.enable_timing(1)
ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
for _ in range(3):
    s.step(20); tm = s.timing()
    for k in ph: ph[k] += tm[k] / 60.0
print("wall %%.1f us/iter | local %%.1f rhs %%.1f fwd %%.1f bwd %%.1f total %%.1f us | checksum %%.17g" %% (1e6 * wall, 1e3 * ph["local_ms"], 1e3 * ph["rhs_ms"], 1e3 * ph["solve_fwd_ms"], 1e3 * ph["solve_bwd_ms"], 1e3 * ph["total_ms"], float(np.abs(s.m_x).sum())))
'''
def main():
    from __graft_entry__ import load_package
    pkg = load_package()
    scene, reps, variants, envs = "mixed", 3, [], {}
    for a in sys.argv[1:]:
        if a.startswith("scene="): scene = a[6:]
        elif a.startswith("reps="): reps = int(a[5:])
        else:
            name, flags = a.split("=", 1)
            toks = flags.split()
            envs[name] = dict(t.split("=", 1) for t in toks if not t.startswith("-"))      # VAR=value tokens: the variant's environment (run-time knobs)
            variants.append((name, [t for t in toks if t.startswith("-")]))
    libs = {}
    for name, flags in variants:
        if not flags:
            libs[name] = None
            continue
        out = os.path.join(ROOT, "admm-elastic-sca_amd", "_build", "libadmm_hip_%s.so" % name)
        pkg._build.build(force=False, extra_hip_flags=flags, out=out, tag="_" + name)
        libs[name] = out
    for rep in range(reps):
        for name, _ in variants:
            env = dict(os.environ, **envs[name])
            if libs[name]: env["ADMM_HIP_LIB"] = libs[name]
            r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, scene=scene)], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
            print("rep %d  %-14s %s" % (rep, name, line), flush=True)
if __name__ == "__main__":
    main()
