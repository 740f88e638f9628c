#!/usr/bin/env python3
"""Probe (GPU box): how the phases of one ADMM iteration of the 1M-tet bar scale with the CUs the solver's stream may use
(hipExtStreamCreateWithCUMask through ADMM_HIP_STREAM_CUMASK), and what a mask bit means on this part (interleaved over the
XCDs or XCD-major)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()


def mask(bits):
    v = 0
    for b in bits: v |= 1 << b
    return "%064x" % v


CASES = [("all 256", None),
         ("low 128 bits", mask(range(128))), ("even bits (128)", mask(range(0, 256, 2))), ("bits with (i//8)%2==0 (128)", mask([i for i in range(256) if (i // 8) % 2 == 0])),
         ("low 192 bits", mask(range(192))), ("3 of every 4 bits (192)", mask([i for i in range(256) if i % 4 != 3])),
         ("low 64 bits", mask(range(64))), ("every 4th bit (64)", mask(range(0, 256, 4))), ("bits i%32<8 (64)", mask([i for i in range(256) if i % 32 < 8])),
         ("low 32 bits", mask(range(32))), ("every 8th bit (32)", mask(range(0, 256, 8)))]
for name, m in CASES:
    if m is None: os.environ.pop("ADMM_HIP_STREAM_CUMASK", None)
    else: os.environ["ADMM_HIP_STREAM_CUMASK"] = m
    s = pkg.make_bar_system(32, 32, 163); s.initialize()
    s.step(20); s.sync()
    s.enable_timing(1)
    ph = dict(local_ms=0.0, rhs_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
    for _ in range(2):
        s.step(20); tm = s.timing()
        for k in ph: ph[k] += tm[k] / 40.0
    print("%-32s local %.3f  rhs %.3f  fwd %.3f  bwd %.3f  iteration %.3f ms" % (name, ph["local_ms"], ph["rhs_ms"], ph["solve_fwd_ms"], ph["solve_bwd_ms"], ph["total_ms"]), flush=True)
    del s
