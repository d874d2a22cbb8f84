"""Sweepers with the reference's surface (SURVEY.md 8b "Sweeper surface"):
``predict, update_nodes, compute_residual, compute_end_point, integrate, updateVariableCoeffs,
get_Qdelta_implicit / _explicit, coll, params, QI, QE, parallelizable, genQI, genQE, level, rank``.

Mirrors  pySDC/core/sweeper.py:33-276, pySDC/implementations/sweeper_classes/generic_implicit.py:4-131 and
imex_1st_order.py:6-137.  When the problem is one the engine can sweep in one call (``prob.fused``: periodic
finite-difference operators) each method is ONE C-ABI call on the level's device slabs; otherwise the same
algorithm runs node by node on ``hip_mesh`` operations with ``prob.eval_f`` / ``prob.solve_system``."""
import logging
import os

import numpy as np

from pysdc_amd import lib as Lb
from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS
from pysdc_amd.errors import ParameterError


# compute_residual queues the residual and does not wait for it (include/sdcmi.h: sdc_residual_post) where the level's status
# object can hold a number that is on its way; PYSDC_AMD_QUEUED_RESIDUAL=0: the blocking call, as for foreign level objects
QUEUED_RESIDUALS = os.environ.get('PYSDC_AMD_QUEUED_RESIDUAL', '1') != '0'

# L.status.residual from the node-wise max norms, per residual_type (core/sweeper.py:200-215)
_RESIDUAL_REDUCERS = {
    'full_abs': lambda norms, L: max(norms),
    'last_abs': lambda norms, L: norms[-1],
    'full_rel': lambda norms, L: max(norms) / abs(L.u[0]),
    'last_rel': lambda norms, L: norms[-1] / abs(L.u[0]),
}

# every controller stage that asks for the residual after a sweep or a transfer (controller_nonMPI.py / controller_MPI.py)
_STAGES_AFTER_SWEEP = {'IT_CHECK', 'IT_FINE', 'IT_DOWN', 'IT_UP', 'IT_COARSE'}


class _Pars:
    """pySDC/core/sweeper.py:20-30."""

    def __init__(self, pars):
        self.do_coll_update = False
        self.initial_guess = 'spread'
        self.skip_residual_computation = ()  # stages in which compute_residual returns at once
        for k, v in pars.items():
            if k != 'collocation_class':
                setattr(self, k, v)

    def get(self, key, default=None):
        return self.__dict__.get(key, default)


def _aliases(cls):
    return {k for k, v in QDELTA_GENERATORS.items() if v is cls}


class Sweeper:
    def __init__(self, params, level):
        self.logger = logging.getLogger('sweeper')
        if 'num_nodes' not in params:
            msg = 'need num_nodes to instantiate step, only got %s' % str(params.keys())
            self.logger.error(msg)
            raise ParameterError(msg)
        if 'collocation_class' not in params:
            params['collocation_class'] = CollBase
        if params.get('initial_guess', 'spread') == 'random':
            params['random_seed'] = params.get('random_seed', 1984)
            self.rng = np.random.RandomState(params['random_seed'])
        self.params = _Pars(params)
        self.coll = params['collocation_class'](**params)
        if not (self.coll.right_is_node or self.params.do_coll_update):
            # (the end point is not a node: only the quadrature over all nodes reaches it, core/sweeper.py:82-87)
            self.params.do_coll_update = True
            self.logger.warning('we need to do a collocation update here, since the right end point is not a node. Changing this!')
        self.__level = level
        self.parallelizable = False
        for name in ['genQI', 'genQE']:
            if hasattr(self, name):
                delattr(self, name)

    # ---- Q-Delta matrices (pySDC/core/sweeper.py:97-123) --------------------------------------------------
    def buildGenerator(self, qdType):
        if qdType not in QDELTA_GENERATORS:
            raise ParameterError(f'unknown Q-Delta type {qdType!r}')
        return QDELTA_GENERATORS[qdType](qGen=self.coll.generator, tLeft=self.coll.tleft)

    def _qdelta(self, slot, qd_type, k, explicit):
        """(M+1) x (M+1) Q-Delta matrix with a zero first row, from the generator kept in `slot` ('genQI' / 'genQE'; rebuilt
        when another type is asked for).  Implicit: lower triangular, zero first column; explicit: strictly lower
        triangular with the node distances in the first column.  A diagonal result makes the sweeper parallelizable
        (pySDC/core/sweeper.py:97-123)."""
        gen = getattr(self, slot, None)
        if gen is None or qd_type not in _aliases(type(gen)) | {type(gen).__name__}:
            gen = self.buildGenerator(qd_type)
            setattr(self, slot, gen)
        full = np.zeros(self.coll.Qmat.shape, dtype=float)
        if explicit:
            full[1:, 1:], full[1:, 0] = gen.genCoeffs(k=k, dTau=True)
        else:
            full[1:, 1:] = gen.genCoeffs(k=k)
        above = np.triu(full, k=0 if explicit else 1)
        np.testing.assert_array_equal(above, np.zeros(full.shape),
                                      err_msg=('Strictly lower' if explicit else 'Lower') + ' triangular matrix expected!')
        if np.allclose(np.diag(np.diag(full)), full):
            self.parallelizable = True
        return full

    def get_Qdelta_implicit(self, qd_type, k=None):
        return self._qdelta('genQI', qd_type, k, explicit=False)

    def get_Qdelta_explicit(self, qd_type, k=None):
        return self._qdelta('genQE', qd_type, k, explicit=True)

    def updateVariableCoeffs(self, k):
        """pySDC/core/sweeper.py:262-276."""
        changed = False
        if hasattr(self, 'genQI') and self.genQI.isKDependent():
            self.QI = self.get_Qdelta_implicit(type(self.genQI).__name__, k=k)
            changed = True
        if hasattr(self, 'genQE') and self.genQE.isKDependent():
            self.QE = self.get_Qdelta_explicit(type(self.genQE).__name__, k=k)
            changed = True
        if changed and self._fused():
            self.push_coeffs(self._dev().engine)

    # ---- engine plumbing ----------------------------------------------------------------------------------
    imex = False
    _skip_sent = False
    _foreign = None

    def push_coeffs(self, engine):
        qe = getattr(self, 'QE', None)
        if engine.ncomp == 2 and qe is None:
            qe = np.zeros_like(self.coll.Qmat)
        engine.set_coeffs(self.coll.Qmat, self.QI, qe if engine.ncomp == 2 else None, self.coll.nodes,
                          self.coll.weights)

    def _dev(self):
        """the device state of this sweeper's level: the level itself when it is a pysdc_amd.level.Level, else a
        ForeignLevelState kept here for a level object of another framework (the reference's pySDC Level)"""
        from pysdc_amd.level import DeviceBacked, ForeignLevelState

        L = self.level
        if isinstance(L, DeviceBacked):
            return L
        if self._foreign is None or self._foreign.host is not L:
            self._foreign = ForeignLevelState(L, self)
        return self._foreign.sync_in()

    def _fused(self):
        L = self.level
        if L is None:
            return False
        P = L.prob
        if not getattr(P, 'fused', False):
            return False
        return (getattr(P, 'ncomp', 1) == 2) == self.imex

    def _node_times(self):
        L = self.level
        return [L.time] + [L.time + L.dt * tau for tau in self.coll.nodes]

    def _push_forcing(self):
        P = self.level.prob
        if hasattr(P, 'forcing_g'):
            self._dev().engine.set_forcing_values([float(P.forcing_g(t)) for t in self._node_times()])

    # ---- predict (pySDC/core/sweeper.py:125-162) ------------------------------------------------------------
    def predict(self):
        L = self.level
        P = L.prob
        M = self.coll.num_nodes
        guess = self.params.initial_guess
        if guess not in ('spread', 'copy', 'zero', 'random'):
            raise ParameterError(f'initial_guess option {guess} not implemented')
        if self._fused():
            D = self._dev()
            e = D.engine
            self._push_forcing()
            fu = ff = 0.0
            if guess == 'random':
                # the reference draws one scalar per node and field (sweeper.py:155-156); the engine fills all
                # nodes with one pair, so draw node by node on the generic path instead
                return self._predict_generic()
            if hasattr(L.status, 'drop_deferred_residual'):
                L.status.drop_deferred_residual()
            e.predict(L.time, L.dt, guess, fu, ff)
            D.u.mark(range(M + 1))
            D.f.mark(range(M + 1))
            D._res_cache = None
            if hasattr(D, '_bump'):
                D._bump(f0=True)
        else:
            return self._predict_generic()
        L.status.unlocked = True
        L.status.updated = True

    def _predict_generic(self):
        L = self.level
        P = L.prob
        L.f[0] = P.eval_f(L.u[0], L.time)
        for m in range(1, self.coll.num_nodes + 1):
            g = self.params.initial_guess
            if g == 'spread':
                L.u[m] = P.dtype_u(L.u[0])
                L.f[m] = P.eval_f(L.u[m], L.time + L.dt * self.coll.nodes[m - 1])
            elif g == 'copy':
                L.u[m] = P.dtype_u(L.u[0])
                L.f[m] = P.dtype_f(L.f[0])
            elif g == 'zero':
                L.u[m] = P.dtype_u(init=P.init, val=0.0)
                L.f[m] = P.dtype_f(init=P.init, val=0.0)
            elif g == 'random':
                L.u[m] = P.dtype_u(init=P.init, val=self.rng.rand(1)[0])
                L.f[m] = P.dtype_f(init=P.init, val=self.rng.rand(1)[0])
        L.status.unlocked = True
        L.status.updated = True

    # ---- residual (pySDC/core/sweeper.py:164-215) -----------------------------------------------------------
    def compute_residual(self, stage=''):
        L = self.level
        if stage in self.params.skip_residual_computation:
            L.status.residual = 0.0 if L.status.residual is None else L.status.residual
            return None
        rt = L.params.residual_type
        if rt not in Lb.RES_TYPES:
            raise ParameterError(
                f'residual_type = {rt} not implemented, choose full_abs, last_abs, full_rel or last_rel instead'
            )
        if self._fused():
            # the device state is unchanged since the last evaluation -> same value, no second pass
            # (the controller calls this twice per iteration)
            D = self._dev()

            def evaluate(D=D, rt=rt, dt=L.dt):
                res, norms = D.engine.residual(dt, rt)
                D._res_cache = ((rt, dt), res)
                D.publish_residual_norms(norms)  # node-wise max norms; the M residual vectors are not materialised
                return res

            own_status = hasattr(L.status, 'residual_is_deferred')   # (the product's LevelStatus: may hold what is not a number yet)
            if D._res_cache is not None and D._res_cache[0] == (rt, L.dt):
                L.status.residual = D._res_cache[1]
            elif (L.params.restol < 0 and own_status
                  and getattr(D.engine, 'residual_deferred', lambda: False)()):
                # the state a spread predictor left, its norm not computed yet (include/sdcmi.h: sdc_residual_deferred), and a
                # convergence test that cannot depend on the value (restol < 0): evaluated when somebody reads the attribute
                L.status.residual = evaluate
            elif QUEUED_RESIDUALS and own_status and hasattr(D.engine, 'residual_post') and hasattr(D, 'publish_residual_future'):
                # queued, not waited for (include/sdcmi.h: sdc_residual_post): the device finishes the number and the test
                # against restol and leaves both in pinned host memory; whoever reads L.status.residual, L.residual or the
                # convergence flag collects them there - a run with a fixed number of sweeps never waits
                me = None
                if (getattr(D, 'integrals_wanted', False) and not D._view_offset() and stage in ('IT_FINE', 'IT_DOWN', 'IT_UP')
                        and getattr(D.engine, 'residual_route', lambda dt: 3)(L.dt) == 3):
                    # a coarser level follows: its FAS correction starts from integrate() of THIS state (BaseTransfer.restrict) -
                    # and this residual is one pass over F in real space (the only route that can write the sums on its way)
                    me = self._integral_fields()
                fut = D.engine.residual_post(L.dt, rt, restol=L.params.restol,
                                             integrals=None if me is None else [x.ptr for x in me])
                D._res_cache = ((rt, L.dt), fut, me if (me is not None and D.engine.integrals_written) else None)
                D.publish_residual_future(fut)
                L.status.residual = fut
            else:
                L.status.residual = evaluate()
        else:
            # node by node on datatype operations (core/sweeper.py:186-215): residual_m = u0 + (Q F)_m - u_m (+ tau_m)
            vectors = self.integrate()
            for m, r in enumerate(vectors):
                r += L.u[0] - L.u[m + 1]
                if L.tau[m] is not None:
                    r += L.tau[m]
            L.residual = vectors
            norms = [abs(r) for r in vectors]
            L.status.residual = _RESIDUAL_REDUCERS[rt](norms, L)
        L.status.updated = False
        return None

    def compute_end_point(self):
        raise NotImplementedError('ERROR: sweeper has to implement compute_end_point(self)')

    def integrate(self):
        raise NotImplementedError('ERROR: sweeper has to implement integrate(self)')

    def update_nodes(self):
        raise NotImplementedError('ERROR: sweeper has to implement update_nodes(self)')

    def _integral_fields(self):
        """one buffer, the M integrals one behind the other (a transfer class can then restrict them together)"""
        from pysdc_amd.hip_mesh import hip_mesh

        D = self._dev()
        M, size = self.coll.num_nodes, D.engine.N
        buf = hip_mesh(((M * size,), None, np.dtype('float64')), val=None)
        return [hip_mesh.view(buf.ptr + 8 * k * size, D._field_shape(), keep=buf) for k in range(M)]

    def _integrate_fused(self):
        L = self.level
        D = self._dev()
        # the residual of this very state brought the quadrature sums along (compute_residual on a level whose restriction
        # asks for them next: same pass over F) - handed out once, the caller owns them like any result of integrate()
        cache = D._res_cache
        if cache is not None and len(cache) > 2 and cache[2] is not None and cache[0][1] == L.dt:
            D._res_cache = (cache[0], cache[1], None)
            return cache[2]
        me = self._integral_fields()
        D.engine.integrate(L.dt, [x.ptr for x in me])
        return me

    def _update_nodes_fused(self):
        L = self.level
        assert L.status.unlocked
        M = self.coll.num_nodes
        D = self._dev()
        if not all(D.u[m] is not None for m in range(M + 1)):
            raise ParameterError('update_nodes needs values at all nodes (predict first)')
        self._push_forcing()
        D.engine.set_unlocked(True)
        if hasattr(L.status, 'drop_deferred_residual'):
            L.status.drop_deferred_residual()   # (belongs to the state this sweep replaces)
        # skip_residual_computation covering every stage that follows a sweep: the engine then only moves the iterate
        skip = _STAGES_AFTER_SWEEP <= set(self.params.skip_residual_computation)
        if skip != self._skip_sent:
            D.engine.set_skip_residual(skip)
            self._skip_sent = skip
        D.engine.sweep(L.time, L.dt)
        D._res_cache = None
        L.status.updated = True

    def _end_point_fused(self):
        L = self.level
        dcu = not (self.coll.right_is_node and not self.params.do_coll_update)
        D = self._dev()
        D.engine.end_point(L.dt, dcu)
        D.publish_uend()

    @property
    def level(self):
        return self.__level

    @level.setter
    def level(self, L):
        # the reference asserts its own Level class here (core/sweeper.py:245-256); this sweeper serves any object with
        # that surface: a pysdc_amd.level.Level (device slabs built in) or the reference's Level (ForeignLevelState)
        assert all(hasattr(L, a) for a in ('u', 'f', 'tau', 'uend', 'status', 'params', 'prob', 'dt', 'time'))
        self.__level = L

    @property
    def rank(self):
        return 0


# ---- node-by-node algorithm on datatype operations (problems the engine cannot sweep in one call) ----------
def _rhs_sum(L, j, imex):
    """f_j as one field: impl + expl for IMEX right-hand sides"""
    return L.f[j].impl + L.f[j].expl if imex else L.f[j]


def _quadrature(sw, weights_of_row):
    """[dt * sum_j row[j] * f_j for each row]: rows of Qmat give integrate(), the weights give the end point"""
    L = sw.level
    P = L.prob
    out = []
    for row in weights_of_row:
        acc = P.dtype_u(P.init, val=0.0)
        for j in range(1, sw.coll.num_nodes + 1):
            acc += L.dt * row[j] * _rhs_sum(L, j, sw.imex)
        out.append(acc)
    return out


def _weighted_f(sw, m, j):
    """dt * (QI[m][j] f_impl_j (+ QE[m][j] f_expl_j))"""
    L = sw.level
    if sw.imex:
        return L.dt * (sw.QI[m, j] * L.f[j].impl + sw.QE[m, j] * L.f[j].expl)
    return L.dt * sw.QI[m, j] * L.f[j]


def _sweep_nodes(sw):
    """one sweep, node after node (generic_implicit.py:51-103, imex_1st_order.py:57-108): known terms first
    (u0 + dt (Q - QDelta) F^k + tau), then for each node the lower-triangular part with the NEW f values, the
    implicit solve and the evaluation of f.  Only generic_implicit skips the solve when its factor is zero."""
    L = sw.level
    P = L.prob
    assert L.status.unlocked
    M = sw.coll.num_nodes
    known = _quadrature(sw, [sw.coll.Qmat[m] for m in range(1, M + 1)])
    for m in range(M):
        for j in range(1, M + 1):
            known[m] -= _weighted_f(sw, m + 1, j)
        known[m] += L.u[0]
        if L.tau[m] is not None:
            known[m] += L.tau[m]
    for m in range(M):
        rhs = P.dtype_u(known[m])
        for j in range(1, m + 1):
            rhs += _weighted_f(sw, m + 1, j)
        factor = L.dt * sw.QI[m + 1, m + 1]
        t_node = L.time + L.dt * sw.coll.nodes[m]
        if factor == 0 and not sw.imex:
            L.u[m + 1] = rhs
        else:
            L.u[m + 1] = P.solve_system(rhs, factor, L.u[m + 1], t_node)
        L.f[m + 1] = P.eval_f(L.u[m + 1], t_node)
    L.status.updated = True


def _end_point(sw):
    """generic_implicit.py:105-131 / imex_1st_order.py:110-137"""
    L = sw.level
    P = L.prob
    if sw.coll.right_is_node and not sw.params.do_coll_update:
        L.uend = P.dtype_u(L.u[-1])
        return
    weights = np.concatenate([[0.0], sw.coll.weights])
    uend = P.dtype_u(L.u[0])
    uend += _quadrature(sw, [weights])[0]
    if L.tau[-1] is not None:
        uend += L.tau[-1]
    L.uend = uend


class generic_implicit(Sweeper):
    """generic_implicit.py:4-131."""

    def __init__(self, params, level):
        if 'QI' not in params:
            params['QI'] = 'IE'
        super().__init__(params, level)
        self.QI = self.get_Qdelta_implicit(qd_type=self.params.QI)

    def integrate(self):
        if self._fused() and not self._dev()._view_offset():  # (odd-extended engine fields: use the views)
            return self._integrate_fused()
        return _quadrature(self, [self.coll.Qmat[m] for m in range(1, self.coll.num_nodes + 1)])

    def update_nodes(self):
        return self._update_nodes_fused() if self._fused() else _sweep_nodes(self)

    def compute_end_point(self):
        return self._end_point_fused() if self._fused() else _end_point(self)


class imex_1st_order(Sweeper):
    """imex_1st_order.py:6-137."""

    imex = True

    def __init__(self, params, level):
        if 'QI' not in params:
            params['QI'] = 'IE'
        if 'QE' not in params:
            params['QE'] = 'EE'
        super().__init__(params, level)
        self.QI = self.get_Qdelta_implicit(qd_type=self.params.QI)
        self.QE = self.get_Qdelta_explicit(qd_type=self.params.QE)

    def integrate(self):
        if self._fused() and not self._dev()._view_offset():  # (odd-extended engine fields: use the views)
            return self._integrate_fused()
        return _quadrature(self, [self.coll.Qmat[m] for m in range(1, self.coll.num_nodes + 1)])

    def update_nodes(self):
        return self._update_nodes_fused() if self._fused() else _sweep_nodes(self)

    def compute_end_point(self):
        return self._end_point_fused() if self._fused() else _end_point(self)

    def get_sweeper_mats(self):
        return self.QE[1:, 1:], self.QI[1:, 1:], self.coll.Qmat[1:, 1:]
