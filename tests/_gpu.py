"""GPU-side helpers: build a SweepEngine from a golden case / oracle problem description."""
import numpy as np

from pysdc_amd import lib as L
from pysdc_amd.engine import SweepEngine
from pysdc_amd.fd import periodic_operator_stencil, grid_1d


def norm_nvars(nv):
    return (nv,) if isinstance(nv, int) else tuple(nv)


def profile_for(nvars, freq):
    """prod_i sin(pi k_i x_i) on the reference's grid orientation (generic_ND_FD.py:171-180)."""
    ndim = len(nvars)
    freq = (freq,) * ndim if isinstance(freq, int) else tuple(freq)
    _, x = grid_1d(nvars[0], 'periodic')
    if ndim == 1:
        return np.sin(np.pi * freq[0] * x)
    if ndim == 2:
        return np.sin(np.pi * freq[0] * x[None, :]) * np.sin(np.pi * freq[1] * x[:, None])
    return (np.sin(np.pi * freq[0] * x[None, :, None]) * np.sin(np.pi * freq[1] * x[:, None, None])
            * np.sin(np.pi * freq[2] * x[None, None, :]))


def forcing_g(nu, freq, ndim, t):
    freq = (freq,) * ndim if isinstance(freq, int) else tuple(freq)
    return nu * np.pi**2 * sum([f**2 for f in freq]) * np.cos(t) - np.sin(t)


def engine_for(prob, pp, M):
    """engine configured for one of the oracle problem kinds (periodic only)."""
    nvars = norm_nvars(pp['nvars'])
    dx, _ = grid_1d(nvars[0], 'periodic')
    order = pp.get('order', 2)
    if prob == 'heat_unforced':
        e = SweepEngine(nvars, M, 1)
        e.set_stencil(0, *periodic_operator_stencil(2, order, 'center', dx, pp.get('nu', 0.1)))
    elif prob == 'heat_forced':
        e = SweepEngine(nvars, M, 2)
        e.set_stencil(0, *periodic_operator_stencil(2, order, 'center', dx, pp.get('nu', 0.1)))
        e.set_forcing_profile(profile_for(nvars, pp.get('freq', 2)))
    elif prob == 'advection':
        e = SweepEngine(nvars, M, 1)
        e.set_stencil(0, *periodic_operator_stencil(1, order, pp.get('stencil_type', 'center'), dx, -pp.get('c', 1.0)))
    elif prob == 'advdiff':
        e = SweepEngine(nvars, M, 2)
        e.set_stencil(0, *periodic_operator_stencil(2, order, 'center', dx, pp.get('nu', 0.02)))
        e.set_stencil(1, *periodic_operator_stencil(1, order, pp.get('stencil_type', 'center'), dx, -pp.get('c', 1.0)))
    else:
        raise ValueError(prob)
    return e


def set_case_coeffs(e, case, QI=None):
    e.set_coeffs(case['coll_Qmat'], case['coll_QI'] if QI is None else QI, case.get('coll_QE'),
                 case['coll_nodes'], case['coll_weights'])


def set_forcing_times(e, meta, t, dt, nodes):
    if meta['prob'] != 'heat_forced':
        return
    pp = meta['prob_params']
    nd = len(norm_nvars(pp['nvars']))
    ts = [t] + [t + dt * tau for tau in nodes]
    e.set_forcing_values([forcing_g(pp.get('nu', 0.1), pp.get('freq', 2), nd, s) for s in ts])
