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

    def eval_f(self, u, t):
        return self.o.eval_f(np.asarray(u), t)


class _Sweep:
    def __init__(self, lvl):
        self.l = lvl

    def predict(self):
        O.predict(self.l.o)

    def update_nodes(self):
        O.sweep(self.l.o)

    def compute_residual(self, stage=''):
        O.compute_residual(self.l.o)
        self.l.status.residual = self.l.o.status_residual

    def compute_end_point(self):
        O.compute_end_point(self.l.o)

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
    def __getitem__(self, m):
        return self.l.o.f[m]

    def __setitem__(self, m, value):
        self.l.o.f[m] = value


class OracleLevel:
    def __init__(self, olevel, level_params):
        self.o = olevel
        self.params = LevelParams(level_params)
        self._status = LevelStatus()
        self.prob = _Prob(olevel.prob)
        self.sweep = _Sweep(self)
        self.u = _UList(self)
        self.f = _FList(self)
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
        object.__setattr__(self, k, v)


class OracleStep:
    """description['oracle_level_factory']() -> oracle Level."""

    def __init__(self, description):
        self.params = StepParams(description.get('step_params', {}))
        self.status = StepStatus()
        ol = description['oracle_level_factory']()
        self.levels = [OracleLevel(ol, description['level_params'])]
        self.levels[0]._status = _TimeStatus(ol)
        self.prev = None

    @property
    def dt(self):
        return self.levels[0].dt

    @property
    def time(self):
        return self.levels[0].time

    def reset_step(self):
        self.levels[0].reset_level()

    def init_step(self, u0):
        self.levels[0].u[0] = u0
