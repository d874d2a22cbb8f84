import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysdc_amd.problems import heatNd_unforced
from pysdc_amd.transfer import mesh_to_mesh
pf = heatNd_unforced(nvars=(256,)*3, nu=0.1, freq=2)
pc = heatNd_unforced(nvars=(128,)*3, nu=0.1, freq=2)
T = mesh_to_mesh(pf, pc, dict(iorder=6, rorder=2, periodic=True))
F, G = pf.u_init, pc.u_init
F[:] = 1.0; G[:] = 1.0
for name, fn, arg in (('restrict', T.restrict, F), ('prolong', T.prolong, G)):
    for _ in range(3): fn(arg)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): fn(arg)
    torch.cuda.synchronize(); print(name, (time.perf_counter()-t)/50*1e6, 'us')
