"""two-level Allen-Cahn PFASST: controller_dist with thread ranks on ONE GPU (tests/_fake_dist.py) against controller_nonMPI
emulating the same ranks:   python scripts/tp_check_ac.py N RANKS [restol]"""
import os
import sys
import threading
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, nranks = int(sys.argv[1]), int(sys.argv[2])
restol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8
import numpy as np
import torch

import bench
from pysdc_amd.controller import controller_nonMPI, controller_dist
from pysdc_amd.sweepers import imex_1st_order
from pysdc_amd.stats import get_sorted
from pysdc_amd.transfer import mesh_to_mesh
from tests import _fake_dist as FD
from tests._cases import rel_err

desc = dict(problem_class=bench.allencahn_ref2d_extruded(),
            problem_params=dict(nvars=[(n,) * 3, (n // 2,) * 3], nu=2, eps=0.04, radius=0.25),
            sweeper_class=imex_1st_order, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU', QE='EE'),
            level_params=dict(dt=1e-3, restol=restol, nsweeps=1), step_params=dict(maxiter=50 if restol > 0 else 4),
            space_transfer_class=mesh_to_mesh, space_transfer_params=dict(iorder=6, rorder=2, periodic=True))
cpar = dict(logger_level=40, predict_type='pfasst_burnin')
Tend = 1e-3 * nranks
C = controller_nonMPI(nranks, cpar, desc)
ref, st = C.run(C.MS[0].levels[0].prob.u_exact(0.0), 0.0, Tend)
ref = ref.get()
nref = [v for _, v in get_sorted(st, type='niter', sortby='time')]
del C
torch.cuda.empty_cache()
world = FD.World(nranks)
out, its, errors = {}, {}, []


def rank_main(rank):
    try:
        FD.bind(world, rank)
        Cd = controller_dist(dict(cpar, comm_wire='shm'), desc, dist=FD)
        uend, stats = Cd.run(Cd.S.levels[0].prob.u_exact(0.0), 0.0, Tend)
        out[rank] = uend.get()
        its[rank] = [v for _, v in get_sorted(stats, type='niter')]
    except Exception:  # noqa: BLE001
        errors.append(traceback.format_exc())
        try:
            world.barrier.abort()
        except Exception:  # noqa: BLE001
            pass


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=900)
if errors:
    print(errors[0])
print(n, nranks, 'lazy', os.environ.get('SDC_LAZY_MIN_BYTES'), 'niter ref', nref, 'dist', [its.get(r) for r in range(nranks)],
      'rel err per rank:', [float('%.2e' % rel_err(out[r], ref)) for r in sorted(out)])
