"""Level and Step containers with the reference's surface (pySDC/core/level.py:42-191, pySDC/core/step.py:24-331)
on top of device slabs: ``L.u[m]``, ``L.f[m]``, ``L.tau[m]``, ``L.uend`` are views into the SweepEngine's
U / F / TAU / UEND slabs; assigning to them copies into the slab (SURVEY.md 8b "Level/data surface")."""
import logging
import os

import numpy as np

from pysdc_amd import lib as Lb
from pysdc_amd.engine import SweepEngine
from pysdc_amd.errors import ParameterError
from pysdc_amd.hip_mesh import hip_mesh, hip_imex_mesh


# The residual of the state a spread predictor leaves is evaluated when somebody reads L.status.residual (0 in the environment
# variable: when compute_residual is called, like every other residual)
LAZY_PREDICTOR_RESIDUAL = os.environ.get('PYSDC_AMD_LAZY_RESIDUAL', '1') != '0'


class _Frozen:
    def get(self, key, default=None):
        return self.__dict__.get(key, default)


class LevelParams(_Frozen):
    """pySDC/core/level.py:9-21."""

    def __init__(self, params):
        self.dt = None
        self.dt_initial = None
        self.restol = -1.0
        self.nsweeps = 1
        self.residual_type = 'full_abs'
        for k, v in params.items():
            setattr(self, k, v)
        self.dt_initial = self.dt * 1.0 if self.dt is not None else None


class LevelStatus(_Frozen):
    """pySDC/core/level.py:24-39.  ``residual`` may be handed a callable instead of a number: it is then evaluated when
    somebody reads the attribute (the residual of a predictor's state that no convergence test can depend on,
    sweepers.compute_residual) - readers never see anything but the number."""

    def __init__(self):
        self._residual = None
        self.unlocked = False
        self.updated = False
        self.time = None
        self.dt_new = None
        self.sweep = None

    @property
    def residual(self):
        r = self._residual
        if callable(r):
            if not getattr(r, 'queued', False):
                self._residual = None     # (a thunk that fails is not called again)
            # (a residual that is on its way stays where it is should collecting it fail: the engine's error surfaces again
            # for the next reader instead of a None that no comparison understands)
            r = self._residual = r()
        return r

    @residual.setter
    def residual(self, value):
        self._residual = value

    def residual_is_deferred(self):
        """a residual whose device work has NOT been queued yet (a thunk); one that is merely on its way
        (engine.ResidualFuture) is not deferred: nothing has to happen before the state it belongs to changes"""
        return callable(self._residual) and not getattr(self._residual, 'queued', False)

    def peek_residual(self):
        """what `residual` would return, without waiting for a residual that is on its way: the ResidualFuture itself in that
        case (hooks keep it and read the number when the statistics are collected)"""
        r = self._residual
        return r if getattr(r, 'queued', False) else self.residual

    def drop_deferred_residual(self):
        """the state the put-off residual belongs to is about to change (a sweep, a new predictor): nobody has asked"""
        if self.residual_is_deferred():
            self._residual = None

    def get(self, key, default=None):
        return self.residual if key == 'residual' else self.__dict__.get(key, default)

    def __getstate__(self):          # (copies / pickles carry the number, not a closure over the engine)
        d = dict(self.__dict__)
        d['_residual'] = self.residual
        return d


class SlabList:
    """list-like over the fields of one engine slab.  get -> view or None; set -> copy into the slab."""

    def __init__(self, level, slot, length, imex=False):
        self._L, self._slot, self._len, self._imex = level, slot, length, imex
        self._valid = [False] * length
        self._views = [None] * length

    def __len__(self):
        return self._len

    def _norm(self, m):
        if m < 0:
            m += self._len
        if not 0 <= m < self._len:
            raise IndexError(m)
        return m

    def _view(self, m):
        if self._views[m] is None:
            e = self._L.engine
            shape = self._L._field_shape()
            cb = lambda slot=self._slot, m=m: self._L._touched(slot, m)  # noqa: E731
            acc = None
            if self._slot in (Lb.SLOT_U, Lb.SLOT_F):  # node fields may be deferred (sdc_materialize)
                acc = lambda slot=self._slot, m=m: e.materialize(slot, m)  # noqa: E731
            off = 8 * self._L._view_offset()
            # (addresses are asked for when a view is first USED: U[1..M] and F[1..M] are blocks of their own that the engine
            # allocates on first touch - a hook that merely fetches L.u[m] must not cost 86 GB at 1024^3)
            at = lambda comp, slot=self._slot, m=m: e.ptr(slot, m, comp) + off  # noqa: E731
            if self._imex:
                self._views[m] = hip_imex_mesh.view(lambda: at(0), lambda: at(1), shape, keep=e, on_write=cb, on_access=acc)
            else:
                self._views[m] = hip_mesh.view(lambda: at(0), shape, keep=e, on_write=cb, on_access=acc)
        return self._views[m]

    def __getitem__(self, m):
        if isinstance(m, slice):
            return [self[i] for i in range(*m.indices(self._len))]
        m = self._norm(m)
        return self._view(m) if self._valid[m] else None

    def __setitem__(self, m, value):
        m = self._norm(m)
        if value is None:
            self._valid[m] = False
        else:
            if self._slot == Lb.SLOT_TAU and not any(self._valid):
                self._L._activate_tau()
            if self._slot == Lb.SLOT_U and m == 0:
                self._L.settle_residual()   # (a put-off residual belongs to the start value that is about to go)
            v = self._view(m)
            if value is not v:
                v[:] = value
            self._valid[m] = True
        if self._slot == Lb.SLOT_TAU and not any(self._valid):
            self._L.engine.set_tau_active(False)
        self._L._touched(self._slot, m)

    def __iter__(self):
        return (self[m] for m in range(self._len))

    def write(self, m, fill):
        """let ``fill(view)`` produce field m in place (instead of assigning a temporary, which copies)"""
        m = self._norm(m)
        v = self._view(m)
        fill(v)
        self._valid[m] = True
        self._L._touched(self._slot, m)

    def mark(self, ms, valid=True):
        for m in ms:
            self._valid[m] = valid

    def invalidate(self):
        self._valid = [False] * self._len


class DeviceBacked:
    """Device side of one level: the SweepEngine, the slab-backed ``u / f / tau`` lists, the end-value view and the
    caches the sweeper keeps.  ``Level`` below IS one; for a level object of another framework (the reference's own
    ``pySDC.core.level.Level``: plain Python lists, frozen attribute set, core/level.py:42-131) the sweeper keeps a
    ``ForeignLevelState`` beside it."""

    def _init_device_state(self):
        self._engine_obj = None
        self._u = self._f = self._tau = None
        self._uend_valid = False
        self._uend_view = None
        self._res_cache = None
        self.integrals_wanted = False   # set by a BaseTransfer whose fine level this is: compute_residual brings integrate() along
        # change counters of u[0] and f[0] (every path that writes them bumps): lets a transfer recognise a start value it has
        # restricted already (serial MLSDC restricts the same u[0] in every iteration of a step, core/base_transfer.py:113-118)
        self._tok_u0 = 0
        self._tok_f0 = 0

    def _bump(self, u0=False, f0=False):
        if u0:
            self._tok_u0 = getattr(self, '_tok_u0', 0) + 1
        if f0:
            self._tok_f0 = getattr(self, '_tok_f0', 0) + 1

    def settle_residual(self):
        """a residual that was put off (LevelStatus: evaluated when read) is evaluated NOW: the state it belongs to - u[0] in
        particular - is about to be replaced, and whoever reads it later must get the number the eager path stored"""
        st = getattr(self, 'status', None)
        if st is not None and hasattr(st, 'residual_is_deferred') and st.residual_is_deferred():
            st.residual  # noqa: B018  (the property evaluates and keeps the number)

    # what the two flavours provide
    def _db_prob(self):
        raise NotImplementedError

    def _db_sweep(self):
        raise NotImplementedError

    @property
    def engine(self):
        if self._engine_obj is None:
            P, M = self._db_prob(), self._db_sweep().coll.num_nodes
            nvars = getattr(P, 'nvars', None)
            if nvars is None:
                raise ParameterError('problem does not define nvars: cannot create device slabs')
            self._engine_obj = SweepEngine(getattr(P, 'engine_nvars', nvars), M, getattr(P, 'ncomp', 1))
            # a status object that can hold a put-off residual (the product's own Level): the engine may put the norm behind
            # the predictor's residual off until it is asked for
            st = getattr(self, 'status', None)
            if isinstance(st, LevelStatus) and LAZY_PREDICTOR_RESIDUAL:
                self._engine_obj.set_lazy_predictor_residual(True)
            P.bind_engine(self._engine_obj)
            self._db_sweep().push_coeffs(self._engine_obj)
        return self._engine_obj

    def _lists(self):
        if self._u is None:
            M = self._db_sweep().coll.num_nodes
            imex = getattr(self._db_prob(), 'ncomp', 1) == 2
            self._u = SlabList(self, Lb.SLOT_U, M + 1)
            self._f = SlabList(self, Lb.SLOT_F, M + 1, imex=imex)
            self._tau = SlabList(self, Lb.SLOT_TAU, M)

    def _uend_slab_view(self):
        # the end value alternates between two device buffers from step to step (sdc_advance): the view is rebuilt
        # whenever the address has moved; it reports writes (on_write), so the plain address is enough
        e = self.engine
        addr = e.uend_address() + 8 * self._view_offset()
        if self._uend_view is None or self._uend_view._p != addr:
            # (the engine may not have transformed the end value back yet: reading through the view makes it do so)
            self._uend_view = hip_mesh.view(addr, self._field_shape(), keep=e,
                                            on_write=lambda: self._touched(Lb.SLOT_UEND, 0),
                                            on_access=lambda: e.materialize(Lb.SLOT_UEND, 0))
        return self._uend_view

    def _field_shape(self):
        shape = self._db_prob().init[0]
        return (int(shape),) if np.isscalar(shape) else tuple(shape)

    def _view_offset(self):
        return int(getattr(self._db_prob(), 'view_offset', 0))

    def _activate_tau(self):
        e = self.engine
        e.set_tau_active(True)
        e.vec_fill(e.N * e.M, 0.0, e.ptr(Lb.SLOT_TAU, 0))

    def _touched(self, slot=None, m=None):
        """device state was written outside the engine's own sweep calls: drop the cached residual and tell
        the engine which cached transforms are stale (include/sdcmi.h: sdc_invalidate_spectra)."""
        self._res_cache = None
        self._bump(u0=slot is None or (slot == Lb.SLOT_U and m == 0), f0=slot is None or (slot == Lb.SLOT_F and m == 0))
        e = self._engine_obj
        if e is None:
            return
        if self._view_offset() and slot is not None:
            # dirichlet-zero: the interior was written through a view - rebuild the odd extension of that field
            n = self._field_shape()[0]
            comps = range(e.ncomp) if slot == Lb.SLOT_F else (0,)
            for comp in comps:
                Lb.check(e.lib.sdc_odd_mirror(e.ctx, e.ptr(slot, m if slot != Lb.SLOT_UEND else 0, comp), n), e.ctx)
        if slot == Lb.SLOT_U:
            e.invalidate_spectra(1 if m == 0 else 2)
        elif slot == Lb.SLOT_F:
            if m != 0:
                e.invalidate_spectra(4)
        elif slot == Lb.SLOT_TAU:
            # a changed FAS correction changes the residual and the gathered right-hand sides, whoever wrote it
            e.invalidate_spectra(16)
        elif slot == Lb.SLOT_UEND:
            e.invalidate_spectra(8)
        elif slot is None:
            e.invalidate_spectra(15)

    # hooks of the fused sweeper methods (the foreign flavour re-adopts the host's lists here)
    def sync_in(self):
        return self

    def publish_uend(self):
        self._uend_valid = True

    def publish_residual_norms(self, norms):
        self.residual = list(norms)

    def publish_residual_future(self, fut):
        """node norms that are on their way: `residual` (a property of Level) collects them when it is read"""
        self._residual_norms = fut


class ForeignLevelState(DeviceBacked):
    """Device state kept by a pysdc_amd sweeper for a level object that is NOT a pysdc_amd.level.Level - in
    particular the reference's ``pySDC.core.level.Level`` inside the reference's ``Step`` / controllers.

    That level owns plain lists ``u, f, tau`` which ``reset_level`` replaces at every block (core/level.py:110-131),
    and the controller / transfer classes assign owning datatype objects into them (``core/step.py:271``,
    ``controller_nonMPI.py:282-284``, ``core/base_transfer.py:93-251``).  ``sync_in`` - called first thing by every
    fused sweeper method - replaces a plain list it finds by the slab-backed list (ordinary attribute assignment,
    allowed on the frozen class because the attribute exists) after copying the entries the host has put there
    into the slab; from then on assignments through ``L.u[m] = x`` land in device slabs like on a product Level.
    ``L.uend`` receives an OWNING copy (stock hooks keep references to it, hooks/log_solution.py:34-60)."""

    def __init__(self, host, sweeper):
        self.host = host
        self._sweeper = sweeper
        self._init_device_state()

    def _db_prob(self):
        return self.host.prob

    def _db_sweep(self):
        return self._sweeper

    @property
    def u(self):
        return self.sync_in()._u

    @property
    def f(self):
        return self.sync_in()._f

    @property
    def tau(self):
        return self.sync_in()._tau

    def sync_in(self):
        H = self.host
        self._lists()
        for name, sl in (('u', self._u), ('f', self._f), ('tau', self._tau)):
            cur = getattr(H, name)
            if cur is sl:
                continue
            sl.invalidate()
            self._res_cache = None
            self._bump(True, True)
            if name == 'tau' and self._engine_obj is not None:
                self._engine_obj.set_tau_active(False)
            for m, v in enumerate(cur):
                if v is not None:
                    sl[m] = v
            setattr(H, name, sl)
        return self

    def publish_uend(self):
        self.host.uend = hip_mesh(self._uend_slab_view())

    def publish_residual_norms(self, norms):
        self.host.residual = list(norms)

    def publish_residual_future(self, fut):
        self.host.residual = list(fut.norms)


class Level(DeviceBacked):
    """pySDC/core/level.py:42-191."""

    def __init__(self, problem_class, problem_params, sweeper_class, sweeper_params, level_params, level_index):
        self.params = LevelParams(level_params)
        self.status = LevelStatus()
        self.__sweep = sweeper_class(sweeper_params, self)
        self.__prob = problem_class(**problem_params)
        self.level_index = level_index
        M = self.__sweep.coll.num_nodes
        self._init_device_state()
        self.uold = [None] * (M + 1)
        self.fold = [None] * (M + 1)
        self.u_avg = [None] * M
        self.residual = [None] * M
        self.increment = [None] * M
        self.__tag = None

    def _db_prob(self):
        return self.__prob

    def _db_sweep(self):
        return self.__sweep

    @property
    def residual(self):
        """node-wise max norms of the collocation residual (the reference keeps the M residual vectors here,
        core/sweeper.py:186-199; the engine reduces them on the way).  Norms that are still on their way
        (publish_residual_future) are collected now."""
        fut = self.__dict__.get('_residual_norms')
        if fut is not None:
            self.__dict__['_residual_norms'] = None
            self.__dict__['_residual_list'] = list(fut.norms)
        return self.__dict__.get('_residual_list')

    @residual.setter
    def residual(self, value):
        self.__dict__['_residual_norms'] = None
        self.__dict__['_residual_list'] = value

    # ---- device state ----------------------------------------------------------------------------------
    @property
    def u(self):
        self._lists()
        return self._u

    @property
    def f(self):
        self._lists()
        return self._f

    @property
    def tau(self):
        self._lists()
        return self._tau

    @property
    def uend(self):
        if not self._uend_valid:
            return None
        return self._uend_slab_view()

    @uend.setter
    def uend(self, value):
        if value is None:
            self._uend_valid = False
            return
        self._uend_valid = True
        v = self.uend
        if value is not v:
            v[:] = value

    def replace_u0(self, src):
        """u[0] <- src (a device field of this level's shape), e.g. the value received from the previous time slice;
        the engine updates the node norms of the residual on the way when it kept the residual fields"""
        if self._view_offset():
            self.u[0] = src
            return
        self.settle_residual()
        self.engine.replace_u0(src.ptr)
        self._u.mark([0])
        self._res_cache = None
        self._bump(u0=True)

    def received_u0(self):
        """the engine has taken a new u[0] from its communicator (sdc_comm_*: inbox -> sdc_replace_u0 inside the library)"""
        self._lists()
        self._u.mark([0])
        self._res_cache = None
        self._bump(u0=True)

    def refresh_f0(self):
        """f[0] = f(u[0]) after u[0] was replaced (controller_MPI.py:233, controller_nonMPI.py:284).  Nothing on
        the sweep path reads f[0]; an engine-backed problem evaluates it when it is asked for."""
        self._bump(f0=True)
        if getattr(self.prob, 'fused', False) and self._engine_obj is not None:
            Lb.check(self._engine_obj.lib.sdc_defer_f0(self._engine_obj.ctx), self._engine_obj.ctx)
            self._f.mark([0])
        else:
            self.f[0] = self.prob.eval_f(self.u[0], self.time)

    def advance(self):
        """next time step on this very level: u[0] <- uend inside the engine (what core/step.py:271 does with the
        end value of the previous block, controller_nonMPI.py:148); include/sdcmi.h: sdc_advance"""
        self.engine.advance()
        self._u.mark([0])
        self._res_cache = None
        self._bump(True, True)
        self._uend_valid = False  # (the end-value buffer of the finished step now holds this step's start value)

    def start_from_wire(self):
        """next block on a rank that RECEIVED the end value of the previous one as a spectrum (sdc_comm_bcast_end_spectrum): it
        becomes u[0] inside the engine, like advance() on the rank that owns it; include/sdcmi.h: sdc_start_from_spectrum"""
        self.engine.start_from_spectrum()
        self._u.mark([0])
        self._res_cache = None
        self._bump(True, True)
        self._uend_valid = False

    def reset_level(self, reset_status=True):
        """pySDC/core/level.py:110-131."""
        self._lists()
        M = self.__sweep.coll.num_nodes
        self.uend = None
        self._u.invalidate()
        self._f.invalidate()
        self._tau.invalidate()
        if self._engine_obj is not None:
            self._engine_obj.set_tau_active(False)
        self.uold = [None] * (M + 1)
        self.fold = [None] * (M + 1)
        self.u_avg = [None] * M
        self.residual = [None] * M
        self.increment = [None] * M
        self._res_cache = None
        self._bump(True, True)
        if reset_status:
            self.status = LevelStatus()

    # ---- the rest of the reference's surface -------------------------------------------------------------
    @property
    def sweep(self):
        return self.__sweep

    @property
    def prob(self):
        return self.__prob

    @property
    def time(self):
        return self.status.time

    @property
    def dt(self):
        return self.params.dt

    @property
    def tag(self):
        return self.__tag

    @tag.setter
    def tag(self, t):
        self.__tag = t


class StepParams(_Frozen):
    def __init__(self, params):
        self.maxiter = None
        for k, v in params.items():
            setattr(self, k, v)


class StepStatus(_Frozen):
    """pySDC/core/step.py:21-44."""

    def __init__(self):
        self.iter = None
        self.stage = None
        self.slot = None
        self.first = None
        self.last = None
        self.pred_cnt = None
        self.done = None
        self.force_done = None
        self.force_continue = False
        self.prev_done = None
        self.time_size = None
        self.diff_old_loc = None
        self.diff_first_loc = None
        self.restart = False


class Step:
    """pySDC/core/step.py:47-331: one time step with its level hierarchy (list-valued parameters = one entry per
    level, step.py:175-199) and the transfer operators between neighbouring levels (step.py:201-253)."""

    level_class = None  # the container of one level; None = pysdc_amd.level.Level (tests substitute a foreign one)

    def __init__(self, description):
        from pysdc_amd.transfer import BaseTransfer

        self.logger = logging.getLogger('step')
        self.params = StepParams(description.get('step_params', {}))
        self.status = StepStatus()
        self.levels = []
        self.__prev = None
        self.__next = None
        self.__transfer_dict = {}
        self.base_transfer = None
        descr = dict(description)
        for key in ['problem_class', 'sweeper_class', 'sweeper_params', 'level_params']:
            if key not in descr:
                raise ParameterError('need %s to instantiate step, only got %s' % (key, str(descr.keys())))
        descr['problem_params'] = descr.get('problem_params', {})
        descr['base_transfer_class'] = descr.get('base_transfer_class', BaseTransfer)
        descr['base_transfer_params'] = descr.get('base_transfer_params', {})
        descr['space_transfer_class'] = descr.get('space_transfer_class', {})
        descr['space_transfer_params'] = descr.get('space_transfer_params', {})
        descr.pop('step_params', None)
        descr.pop('step_class', None)
        descr_new = dict(descr)
        for key in ('problem_params', 'level_params', 'sweeper_params'):
            descr_new[key] = self._dict_to_list(descr[key])
        descr_list = self._dict_to_list(descr_new)
        if len(descr_list) > 1 and not descr_new['space_transfer_class']:
            raise ParameterError('need space_transfer_class to instantiate step, only got %s' % str(descr_new.keys()))
        for l, d in enumerate(descr_list):
            L = (self.level_class or Level)(d['problem_class'], dict(d['problem_params']), d['sweeper_class'],
                                            dict(d['sweeper_params']), dict(d['level_params']), l)
            self.levels.append(L)
            if l > 0:
                self.connect_levels(descr_new['base_transfer_class'], d['base_transfer_params'],
                                    d['space_transfer_class'], d['space_transfer_params'], self.levels[l - 1], L)

    @staticmethod
    def _dict_to_list(in_dict):
        max_val = 1
        for v in in_dict.values():
            if type(v) is list:
                max_val = max(max_val, len(v))
        ld = [{} for _ in range(max_val)]
        for d in range(len(ld)):
            for k, v in in_dict.items():
                ld[d][k] = v if type(v) is not list else v[min(d, len(v) - 1)]
        return ld

    def connect_levels(self, base_transfer_class, base_transfer_params, space_transfer_class, space_transfer_params,
                       fine_level, coarse_level):
        self.base_transfer = base_transfer_class(fine_level, coarse_level, base_transfer_params,
                                                 space_transfer_class, space_transfer_params)
        self.__transfer_dict[(fine_level, coarse_level)] = self.base_transfer.restrict
        if self.base_transfer.params.finter:
            self.__transfer_dict[(coarse_level, fine_level)] = self.base_transfer.prolong_f
        else:
            self.__transfer_dict[(coarse_level, fine_level)] = self.base_transfer.prolong

    def transfer(self, source, target):
        self.__transfer_dict[(source, target)]()

    @property
    def time(self):
        return self.levels[0].time

    @property
    def dt(self):
        return self.levels[0].dt

    @property
    def prev(self):
        return self.__prev

    @prev.setter
    def prev(self, p):
        self.__prev = p

    def reset_step(self):
        for l in self.levels:
            l.reset_level()

    def init_step(self, u0):
        """pySDC/core/step.py:256-271."""
        assert len(self.levels) >= 1
        assert len(self.levels[0].u) >= 1
        self.levels[0].u[0] = u0
