"""host-side profile of the two-level Allen-Cahn MLSDC step (where does the non-GPU time go?)"""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
import numpy as np
from pysdc_amd.controller import controller_nonMPI
from pysdc_amd.problems import allencahn_imex
from pysdc_amd.sweepers import imex_1st_order
from pysdc_amd.transfer import mesh_to_mesh
import torch
n = 256
desc = dict(problem_class=allencahn_imex, problem_params=dict(nvars=[(n,)*3, (n//2,)*3], eps=0.04, radius=0.25, init_type='sphere'),
            sweeper_class=imex_1st_order, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU', QE='EE'),
            level_params=dict(dt=1e-3, restol=-1.0, nsweeps=1), step_params=dict(maxiter=4),
            space_transfer_class=mesh_to_mesh, space_transfer_params=dict(iorder=6, rorder=2, periodic=True))
C = controller_nonMPI(1, dict(logger_level=40), desc)
u0 = C.MS[0].levels[0].prob.u_exact(0.0)
uend, _ = C.run(u0, 0.0, 1e-3)
torch.cuda.synchronize()
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
uend, _ = C.run(uend, 1e-3, 4e-3)
torch.cuda.synchronize()
pr.disable()
print('3 steps', time.perf_counter() - t)
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
