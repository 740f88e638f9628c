#!/usr/bin/env python3
"""Probe (GPU box): per-tet trace of the Neo-Hookean prox on the headline bar -- line-search evaluations and the gradient at
the warm start -- from the -DADMM_TET_PROFILE variant build.  Question: can the 20-evaluation searches be predicted BEFORE the
search (so that those tets could be handed to a second, compacted launch), and what would the wave-level cost be?

  python tools/probe/ls_predict_gpu.py [dims=32x32x163] [frames=3]     -> stdout + gpurun_out/ls_trace.npz (last frame, quantised)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, os, ctypes, numpy as np
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
dims, frames, iters = %r, %d, 20
s = pkg.make_bar_system(*dims, device_id=0)
s.initialize()
lib = pkg.lib()
n = s.n_tets
npad = (n + 255) // 256 * 256
assert lib.admm_hip_debug_tet_trace(iters, npad) == 0
buf = np.zeros((iters, npad, 2), dtype=np.float32)
for f in range(frames):
    s.step(iters)
    assert lib.admm_hip_debug_tet_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    g0, nfev = buf[:, :n, 0].astype(np.float64), buf[:, :n, 1]
    slow = nfev >= 15
    nw = n // 64
    def wave_max(ev):                  # ev: (iters, n) -> mean over waves and iterations of the wave maximum
        return ev[:, : nw * 64].reshape(iters, nw, 64).max(axis=2).mean()
    one = wave_max(nfev)
    print("frame %%d: evaluations mean %%.2f, wave maximum %%.2f; slow (>= 15) %%.1f %%%% of (tet, iteration)" %% (f, nfev.mean(), one, 100 * slow.mean()))
    print("   slow %% per iteration: " + " ".join("%%.0f" %% (100 * slow[i].mean()) for i in range(iters)))
    q = lambda a: " ".join("%%.1e" %% v for v in np.quantile(a, [0.01, 0.05, 0.5, 0.95, 0.99])) if a.size else "-"
    print("   |g0| quantiles (1 5 50 95 99 %%): slow " + q(g0[slow]) + " | fast " + q(g0[~slow]))
    for thr in (1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2):
        pred = g0 < thr
        fast_cost = wave_max(np.where(pred, 0, nfev))
        sec = 0.0; nsec = 0
        for i in range(iters):
            ev = nfev[i][pred[i]]
            k = (len(ev) + 63) // 64
            nsec += k
            if k: sec += np.concatenate([ev, np.zeros(k * 64 - len(ev), dtype=ev.dtype)]).reshape(k, 64).max(axis=1).sum()
        sec /= iters * nw
        print("   thr %%.0e: predicted slow %%5.1f %%%%, missed %%5.2f %%%%, false alarms %%5.2f %%%%; wave-evaluations: one kernel %%.2f -> fast kernel %%.2f + compacted %%.2f (%%.1f %%%% of the waves again) = %%.2f" %% (
            thr, 100 * pred.mean(), 100 * (slow & ~pred).mean(), 100 * (~slow & pred).mean(), one, fast_cost, sec, 100.0 * nsec / (iters * nw), fast_cost + sec))
    # oracle bound: perfect knowledge
    fast_cost = wave_max(np.where(slow, 0, nfev))
    k = (slow.sum(axis=1) + 63) // 64
    print("   perfect prediction: fast kernel %%.2f + compacted %%.2f = %%.2f" %% (fast_cost, 20.0 * k.sum() / (iters * nw), fast_cost + 20.0 * k.sum() / (iters * nw)))
os.makedirs(os.path.join(%r, "gpurun_out"), exist_ok=True)
lg = np.clip(np.round(4 * np.log2(np.maximum(buf[:, :n, 0], 1e-30))), -127, 127).astype(np.int8)
np.savez_compressed(os.path.join(%r, "gpurun_out", "ls_trace.npz"), nfev=buf[:, :n, 1].astype(np.uint8), log2g0_x4=lg)
'''


def main():
    dims, frames = (32, 32, 163), 3
    for a in sys.argv[1:]:
        if a.startswith("dims="):
            dims = tuple(int(v) for v in a[5:].split("x"))
        if a.startswith("frames="):
            frames = int(a[7:])
    from __graft_entry__ import load_package
    pkg = load_package()
    out = os.path.join(ROOT, "admm-elastic-sca_amd", "_build", "libadmm_hip_prof.so")
    pkg._build.build(force=False, extra_hip_flags=["-DADMM_TET_PROFILE"], out=out, tag="_prof")
    env = dict(os.environ, ADMM_HIP_LIB=out)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, dims, frames, ROOT, ROOT)], env=env, capture_output=True, text=True)
    print(r.stdout)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)


if __name__ == "__main__":
    main()
