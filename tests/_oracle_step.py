"""TEST-ONLY: a Step implementation backed by the CPU oracle, with the attribute surface
pysdc_amd.controller.controller_dist uses.  It lets the one-time-step-per-rank controller (P2P ordering,
done-flag chain, block broadcast, iteration counting) run under gloo on CPU, where no GPU engine exists."""
import numpy as np
import torch
import torch.distributed as dist

from oracle import sdc_oracle as O
from pysdc_amd.level import StepParams, StepStatus, LevelParams, LevelStatus


class np_mesh(np.ndarray):
    def __new__(cls, init, val=0.0):
        if isinstance(init, np.ndarray):
            obj = np.array(init, dtype=float).view(cls)
        else:
            obj = np.full(init[0], float(val)).view(cls)
        return obj

    def as_torch(self):
        return torch.from_numpy(self.view(np.ndarray))

    def bcast(self, root=None, comm=None):
        dist.broadcast(self.as_torch(), src=root, group=comm)
        return self


class _Prob:
    def __init__(self, oprob):
        self.o = oprob
        self.init = (oprob.nvars, None, np.dtype('float64'))
        self.dtype_u = np_mesh

        self.dtype_f = np_mesh
        self.work_counters = getattr(oprob, 'work_counters', {})

    def eval_f(self, u, t):
        return np_mesh(self.o.eval_f(np.asarray(u), t))


class _Sweep:
    def __init__(self, lvl):
        self.l = lvl

    def predict(self):
        O.predict(self.l.o)
        self.l.status.unlocked = True
        self.l.status.updated = True

    def update_nodes(self):
        assert self.l.status.unlocked
        self.l.o.unlocked = True
        O.sweep(self.l.o)
        self.l.status.updated = True

    def compute_residual(self, stage=''):
        O.compute_residual(self.l.o)
        self.l.status.residual = self.l.o.status_residual

    def compute_end_point(self):
        O.compute_end_point(self.l.o)

    def integrate(self):
        return [np_mesh(x) for x in O.integrate(self.l.o)]

    @property
    def coll(self):
        return self.l.o.coll

    def updateVariableCoeffs(self, k):
        pass


class _UList:
    def __init__(self, lvl):
        self.l = lvl

    def __len__(self):
        return len(self.l.o.u)

    def __getitem__(self, m):
        v = self.l.o.u[m]
        return None if v is None else v.view(np_mesh)

    def __setitem__(self, m, value):
        self.l.o.u[m] = None if value is None else np.array(value, dtype=float)


class _FList(_UList):
    def __len__(self):
        return len(self.l.o.f)

    def __getitem__(self, m):
        v = self.l.o.f[m]
        return None if v is None else v.view(np_mesh)

    def __setitem__(self, m, value):
        self.l.o.f[m] = None if value is None else np.array(value, dtype=float)


class _TauList(_UList):
    def __len__(self):
        return len(self.l.o.tau)

    def __getitem__(self, m):
        v = self.l.o.tau[m]
        return None if v is None else v.view(np_mesh)

    def __setitem__(self, m, value):
        self.l.o.tau[m] = None if value is None else np.array(value, dtype=float)


class OracleMeshToMesh:
    """space_transfer_class for oracle-backed levels (same constructor signature as the product classes)."""

    def __init__(self, fine_prob, coarse_prob, params):
        self.T = O.MeshToMesh(fine_prob.o.nvars, coarse_prob.o.nvars, params.get('iorder', 2), params.get('rorder', 2),
                              periodic=params.get('periodic', True))

    def restrict(self, F):
        return np_mesh(self.T.restrict(np.asarray(F)))

    def prolong(self, G):
        return np_mesh(self.T.prolong(np.asarray(G)))


class OracleMeshToMeshFFT2D:
    """oracle-backed stand-in of mesh_to_mesh_fft2d"""

    def __init__(self, fine_prob, coarse_prob, params):
        self.T = O.MeshToMeshFFT2D(fine_prob.o.nvars, coarse_prob.o.nvars)

    def restrict(self, F):
        return np_mesh(self.T.restrict(np.asarray(F)))

    def prolong(self, G):
        return np_mesh(self.T.prolong(np.asarray(G)))


class OracleLevel:
    def __init__(self, olevel, level_params):
        self.o = olevel
        self.params = LevelParams(level_params)
        self._status = LevelStatus()
        self.prob = _Prob(olevel.prob)
        self.sweep = _Sweep(self)
        self.u = _UList(self)
        self.f = _FList(self)
        self.tau = _TauList(self)
        self.uold = [None] * len(olevel.u)
        self.fold = [None] * len(olevel.u)
        self.tag = None
        self.level_index = 0
        self.sweep.rank = 0

    class _St:
        pass

    @property
    def status(self):
        return self._status

    @property
    def time(self):
        return self.o.time

    @property
    def dt(self):
        return self.o.dt

    @property
    def uend(self):
        return None if self.o.uend is None else self.o.uend.view(np_mesh)

    def _touched(self, *a):
        pass

    def reset_level(self):
        self.o.reset()
        self.uold = [None] * len(self.o.u)
        self.fold = [None] * len(self.o.u)
        self._status = _TimeStatus(self.o)


class _TimeStatus(LevelStatus):
    """status whose ``time`` / ``sweep`` write through to the oracle level."""

    def __init__(self, o):
        object.__setattr__(self, '_o', o)
        super().__init__()

    def __setattr__(self, k, v):
        if k == 'time':
            self._o.time = v
        if k == 'sweep':
            self._o.sweep = v
        if k == 'unlocked':
            self._o.unlocked = bool(v)
        object.__setattr__(self, k, v)


class OracleStep:
    """description['oracle_level_factory'] -> callable (single level) or list of callables (finest first),
    each returning an oracle Level; level_params may be a dict or dict of per-level lists."""

    def __init__(self, description):
        from pysdc_amd.transfer import BaseTransfer

        self.params = StepParams(description.get('step_params', {}))
        self.status = StepStatus()
        fac = description['oracle_level_factory']
        facs = fac if isinstance(fac, list) else [fac]
        lp = description['level_params']
        self.levels = []
        self._tr = {}
        self.prev = None
        for l, f in enumerate(facs):
            ol = f()
            lpl = {k: (v[min(l, len(v) - 1)] if isinstance(v, list) else v) for k, v in lp.items()}
            L = OracleLevel(ol, lpl)
            L.level_index = l
            L._status = _TimeStatus(ol)
            self.levels.append(L)
            if l > 0:
                bt = BaseTransfer(self.levels[l - 1], L, description.get('base_transfer_params', {}),
                                  description['space_transfer_class'], description.get('space_transfer_params', {}))
                self.base_transfer = bt
                self._tr[(self.levels[l - 1], L)] = bt.restrict
                self._tr[(L, self.levels[l - 1])] = bt.prolong_f if bt.params.finter else bt.prolong

    def transfer(self, source, target):
        self._tr[(source, target)]()

    @property
    def dt(self):
        return self.levels[0].dt

    @property
    def time(self):
        return self.levels[0].time

    def reset_step(self):
        for L in self.levels:
            L.reset_level()

    def init_step(self, u0):
        self.levels[0].u[0] = u0
