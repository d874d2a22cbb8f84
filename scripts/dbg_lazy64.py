import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysdc_amd import lib as L
from tests import _gpu as G
from tests._cases import load_cases
case = load_cases('sweeps_big3d.npz')['cg64_heat3d_M5_IE']
meta = case['meta']
M, dt, t0 = len(case['coll_nodes']), meta['dt'], meta['t0']
pp = dict(meta['prob_params']); pp.pop('solver_type')
for variant in sys.argv[1:] or ['plain']:
    e = G.engine_for('heat_unforced', pp, M)
    e.set_virtual_sweeps(8)
    G.set_case_coeffs(e, case)
    e.upload(L.SLOT_U, 0, case['u0'])
    e.profile_enable(True)
    e.predict(t0, dt, 'spread')
    out = [e.residual(dt)[0]]
    for k in range(1, 4):
        e.sweep(t0, dt)
        out.append(e.residual(dt)[0])
        print('  sweep', k, sorted(kk for kk in e.profile_read()), flush=True)
        e.profile_enable(True)
        if variant == 'relres':
            e.residual(dt, 'full_rel')
        if variant in ('endpoint', 'download'):
            e.end_point(dt, False)
        if variant == 'download':
            e.download(L.SLOT_UEND)
    print(variant, os.environ.get('SDC_LAZY_MIN_BYTES'), ['%.6e' % v for v in out], 'ref', ['%.6e' % float(case[f'k{k}_res_full_abs']) for k in range(4)], flush=True)
    e.close()
