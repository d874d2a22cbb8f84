import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import numpy as np
from pysdc_amd.controller import controller_nonMPI
from pysdc_amd.synth import init_field
from tests.test_gpu_plugin import description_from
n, nranks, M, dt = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 3, 5, 2e-3
meta = dict(prob='heat_unforced', prob_params=dict(nvars=[n, n, n], nu=0.1, freq=2), sweeper='generic_implicit',
            sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='IE'), level_params=dict(dt=dt, restol=-1),
            maxiter=6, controller_params={}, t0=0.0, Tend=dt * (2 * nranks - 1))
u0h = init_field((n, n, n), 2, 1e-2, 3)
C = controller_nonMPI(nranks, dict(logger_level=40), description_from(meta))
u0 = C.MS[0].levels[0].prob.u_init
u0[:] = u0h
ref, rstats = C.run(u0, meta['t0'], meta['Tend'])
print('ok', float(abs(ref)))
