"""Helpers shared by the CPU (oracle vs golden) and GPU (HIP vs oracle / golden) tests."""
import json
import os

import numpy as np

from oracle import sdc_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_cases(fname):
    z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
    cases = {}
    for key in z.files:
        name, field = key.split('/', 1)
        cases.setdefault(name, {})[field] = z[key]
    for c in cases.values():
        c['meta'] = json.loads(str(c['meta']))
    return cases


def make_oracle_problem(prob, pp):
    pp = dict(pp)
    for key in ('nvars', 'bc', 'freq'):      # (JSON has no tuples)
        if isinstance(pp.get(key), list):
            pp[key] = tuple(pp[key])
    if prob == 'heat_unforced':
        return O.HeatUnforced(**pp)
    if prob == 'heat_forced':
        return O.HeatForced(**pp)
    if prob == 'advection':
        return O.Advection(**pp)
    if prob == 'advdiff':
        return O.AdvectionDiffusionIMEX(**pp)
    if prob == 'vanderpol':
        return O.VanDerPol(**pp)
    if prob == 'ad1d_imex':
        return O.AdvDiff1DIMEX(**pp)
    if prob == 'ad1d_implicit':
        return O.AdvDiff1DImplicit(**pp)
    if prob == 'allencahn2d':
        return O.AllenCahn2D(**pp)
    if prob == 'allencahnNd':
        return O.AllenCahnND(**pp)
    raise ValueError(prob)


def make_oracle_coll(case, QI=None):
    return O.Coll(case['coll_nodes'], case['coll_weights'], case['coll_Qmat'],
                  case['coll_QI'] if QI is None else QI, case.get('coll_QE'),
                  right_is_node=bool(case['coll_right_is_node']), left_is_node=bool(case['coll_left_is_node']))


def fstack(f):
    return np.asarray(f)


def rel_err(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    scale = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / scale
