import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from __graft_entry__ import load_package
pkg=load_package()
dims=tuple(int(a) for a in sys.argv[1:4])
t0=time.time()
s=pkg.make_bar_system(*dims, device_id=-1)
t1=time.time()
s.initialize()
inf=s.info()
t2=time.time()
print(dims,"tets",s.n_tets,"nodes",inf['n_nodes'],"nnzL",inf['nnz_L'],"panelMB",inf['panel_bytes']/1e6,"sn",inf['n_supernodes'],"levels",inf['n_levels'],"maxk",inf['max_super_cols'],"maxr",inf['max_super_rows'],
      "mesh %.2f t_order %.2f t_sym %.2f t_num %.2f init total %.2f"%(t1-t0,inf['t_order_s'],inf['t_symbolic_s'],inf['t_numeric_s'],t2-t1))
