"""controller_dist with thread ranks on ONE GPU (tests/_fake_dist.py) against controller_nonMPI emulating the same
ranks, at any size:   python scripts/tp_check.py N RANKS M [relay] [overlap]"""
import os
import sys
import threading
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, nranks, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4:
    os.environ['PYSDC_AMD_RELAY'] = sys.argv[4]
if len(sys.argv) > 5:
    os.environ['PYSDC_AMD_OVERLAP'] = sys.argv[5]
import numpy as np
import torch

DT = float(os.environ.get('TP_DT', '2e-3'))

from pysdc_amd.controller import controller_nonMPI, controller_dist
from pysdc_amd.synth import init_field
from tests import _fake_dist as FD
from tests._cases import rel_err
from tests.test_gpu_plugin import description_from

meta = dict(prob='heat_unforced', prob_params=dict(nvars=[n, n, n], nu=0.1, freq=2), sweeper='generic_implicit',
            sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='IE'), level_params=dict(dt=DT, restol=-1),
            maxiter=3, controller_params={}, t0=0.0, Tend=DT * (2 * nranks - nranks // 2))
u0h = init_field((n, n, n), 2, 1e-2, 3)
C = controller_nonMPI(nranks, dict(logger_level=40), description_from(meta))
P = C.MS[0].levels[0].prob
u0 = P.u_init
u0[:] = u0h
ref, _ = C.run(u0, meta['t0'], meta['Tend'])
ref = ref.get()
del C, P, u0
torch.cuda.empty_cache()
world = FD.World(nranks)
out, errors = {}, []


def rank_main(rank):
    try:
        FD.bind(world, rank)
        Cd = controller_dist(dict(logger_level=40, comm_wire='shm'), description_from(meta), dist=FD)
        v = Cd.S.levels[0].prob.u_init
        v[:] = u0h
        uend, stats = Cd.run(v, meta['t0'], meta['Tend'])
        out[rank] = uend.get()
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
    t.join(timeout=600)
if errors:
    print(errors[0])
print(n, nranks, M, os.environ.get('PYSDC_AMD_RELAY'), os.environ.get('PYSDC_AMD_OVERLAP'),
      'max|ref| %.3e' % float(np.max(np.abs(ref))), 'max|u0| %.3e' % float(np.max(np.abs(u0h))), 'rel err per rank:', [float('%.2e' % rel_err(out[r], ref)) for r in sorted(out)])
