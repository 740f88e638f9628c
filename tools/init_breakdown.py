#!/usr/bin/env python3
"""Where System::initialize goes (GPU box): python tools/init_breakdown.py [nx ny nz]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
dims = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 32, 163)
t0 = time.time()
s = pkg.make_bar_system(*dims, device_id=0)
t1 = time.time()
s.initialize()
t2 = time.time()
i = s.info()
print("mesh + add forces %.2f s; initialize %.2f s: order %.2f, symbolic %.2f, numeric %.3f, upload %.2f (host threads %d); nodes %d, nnz(L) %d" % (
    t1 - t0, t2 - t1, i["t_order_s"], i["t_symbolic_s"], i["t_numeric_s"], i["t_upload_s"], i["host_threads"], i["n_nodes"], i["nnz_L"]))
t3 = time.time(); s.recompute_weights(); t4 = time.time()
print("recompute_weights %.3f s (numeric %.3f)" % (t4 - t3, s.info()["t_numeric_s"]))
