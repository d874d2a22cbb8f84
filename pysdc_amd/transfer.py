"""Space-time transfer between two levels (FAS restriction / coarse-grid correction).

``BaseTransfer`` restates pySDC/core/base_transfer.py:36-251 against the duck-typed Level surface, so the same
host logic drives device levels (``pysdc_amd.level.Level``) and, in the CPU tests, oracle-backed levels.
``mesh_to_mesh`` is the device version of pySDC/implementations/transfer_classes/TransferMesh.py:9-218 for
periodic, equidistant, nested grids (coarsening factor 2 per axis): the interpolation matrices of
pySDC/helpers/transfer_helper.py:153-186 are applied matrix-free by ``sdc_transfer_prolong / _restrict``."""
import ctypes as C
import logging

import numpy as np

from pysdc_amd import lib as Lb
from pysdc_amd.coeffs import LagrangeApproximation
from pysdc_amd.errors import TransferError, UnlockError
from pysdc_amd.hip_mesh import hip_mesh, hip_imex_mesh


class _BasePars:
    """pySDC/core/base_transfer.py:17-27."""

    def __init__(self, pars):
        self.finter = False
        for k, v in pars.items():
            setattr(self, k, v)


class BaseTransfer:
    def __init__(self, fine_level, coarse_level, base_transfer_params, space_transfer_class, space_transfer_params):
        self.params = _BasePars(base_transfer_params)
        self.logger = logging.getLogger('transfer')
        self.fine = fine_level
        self.coarse = coarse_level
        fine_grid = self.fine.sweep.coll.nodes
        coarse_grid = self.coarse.sweep.coll.nodes
        if len(fine_grid) == len(coarse_grid):
            self.Pcoll = np.eye(len(fine_grid))
            self.Rcoll = np.eye(len(fine_grid))
        else:
            self.Pcoll = self.get_transfer_matrix_Q(fine_grid, coarse_grid)
            self.Rcoll = self.get_transfer_matrix_Q(coarse_grid, fine_grid)
        self.space_transfer = space_transfer_class(
            fine_prob=self.fine.prob, coarse_prob=self.coarse.prob, params=space_transfer_params
        )

    @staticmethod
    def get_transfer_matrix_Q(f_nodes, c_nodes):
        return LagrangeApproximation(c_nodes).getInterpolationMatrix(f_nodes)

    def restrict(self):
        """base_transfer.py:93-168."""
        F, G = self.fine, self.coarse
        PG = G.prob
        SF, SG = F.sweep, G.sweep
        if not F.status.unlocked:
            raise UnlockError('fine level is still locked, cannot use data from there')
        tmp_u = []
        for m in range(1, SF.coll.num_nodes + 1):
            tmp_u.append(self.space_transfer.restrict(F.u[m]))
        G.u[0] = self.space_transfer.restrict(F.u[0])
        for n in range(1, SG.coll.num_nodes + 1):
            G.u[n] = self.Rcoll[n - 1, 0] * tmp_u[0]
            for m in range(1, SF.coll.num_nodes):
                G.u[n] += self.Rcoll[n - 1, m] * tmp_u[m]
        G.f[0] = PG.eval_f(G.u[0], G.time)
        for m in range(1, SG.coll.num_nodes + 1):
            G.f[m] = PG.eval_f(G.u[m], G.time + G.dt * SG.coll.nodes[m - 1])
        tauG = G.sweep.integrate()
        tauF = F.sweep.integrate()
        tmp_tau = []
        for m in range(SF.coll.num_nodes):
            tmp_tau.append(self.space_transfer.restrict(tauF[m]))
        tauFG = []
        for n in range(1, SG.coll.num_nodes + 1):
            tauFG.append(self.Rcoll[n - 1, 0] * tmp_tau[0])
            for m in range(1, SF.coll.num_nodes):
                tauFG[-1] += self.Rcoll[n - 1, m] * tmp_tau[m]
        for m in range(SG.coll.num_nodes):
            G.tau[m] = tauFG[m] - tauG[m]
        if F.tau[0] is not None:
            tmp_tau = []
            for m in range(SF.coll.num_nodes):
                tmp_tau.append(self.space_transfer.restrict(F.tau[m]))
            for n in range(SG.coll.num_nodes):
                for m in range(SF.coll.num_nodes):
                    G.tau[n] += self.Rcoll[n, m] * tmp_tau[m]
        for m in range(1, SG.coll.num_nodes + 1):
            G.uold[m] = PG.dtype_u(G.u[m])
            G.fold[m] = PG.dtype_f(G.f[m])
        G.status.unlocked = True
        return None

    def prolong(self):
        """base_transfer.py:170-207."""
        F, G = self.fine, self.coarse
        PF = F.prob
        SF, SG = F.sweep, G.sweep
        if not G.status.unlocked:
            raise UnlockError('coarse level is still locked, cannot use data from there')
        tmp_u = []
        for m in range(1, SG.coll.num_nodes + 1):
            tmp_u.append(self.space_transfer.prolong(G.u[m] - G.uold[m]))
        for n in range(1, SF.coll.num_nodes + 1):
            for m in range(SG.coll.num_nodes):
                F.u[n] += self.Pcoll[n - 1, m] * tmp_u[m]
        for m in range(1, SF.coll.num_nodes + 1):
            F.f[m] = PF.eval_f(F.u[m], F.time + F.dt * SF.coll.nodes[m - 1])
        return None

    def prolong_f(self):
        """base_transfer.py:209-251."""
        F, G = self.fine, self.coarse
        SF, SG = F.sweep, G.sweep
        if not G.status.unlocked:
            raise UnlockError('coarse level is still locked, cannot use data from there')
        tmp_u, tmp_f = [], []
        for m in range(1, SG.coll.num_nodes + 1):
            tmp_u.append(self.space_transfer.prolong(G.u[m] - G.uold[m]))
            tmp_f.append(self.space_transfer.prolong(G.f[m] - G.fold[m]))
        for n in range(1, SF.coll.num_nodes + 1):
            for m in range(SG.coll.num_nodes):
                F.u[n] += self.Pcoll[n - 1, m] * tmp_u[m]
                F.f[n] += self.Pcoll[n - 1, m] * tmp_f[m]
        return None


class _SpacePars:
    """pySDC/core/space_transfer.py:8-19."""

    def __init__(self, pars):
        self.periodic = False
        self.equidist_nested = True
        self.iorder = 2
        self.rorder = 2
        for k, v in pars.items():
            setattr(self, k, v)


def midpoint_weights(k):
    """Lagrange weights of the k nearest coarse points for the fine point half way between the two middle ones
    (the rows of interpolation_matrix_1d for odd fine indices, transfer_helper.py:160-186)."""
    if k == 0:
        return np.zeros(0)
    t = np.arange(k, dtype=float) - k / 2 + 1  # coarse offsets relative to i; the fine point sits at 0.5
    w = np.ones(k)
    for j in range(k):
        for l in range(k):
            if l != j:
                w[j] *= (0.5 - t[l]) / (t[j] - t[l])
    return w


class mesh_to_mesh:
    def __init__(self, fine_prob, coarse_prob, params):
        self.params = _SpacePars(params)
        self.logger = logging.getLogger('space-transfer')
        self.fine_prob, self.coarse_prob = fine_prob, coarse_prob
        if self.params.rorder % 2 != 0:
            raise TransferError('Need even order for restriction')
        if self.params.iorder % 2 != 0:
            raise TransferError('Need even order for interpolation')
        nf, nc = fine_prob.nvars, coarse_prob.nvars
        if type(nf) is tuple:
            if type(nc) is not tuple:
                raise TransferError('nvars parameter of coarse problem needs to be a tuple')
            if not len(nf) == len(nc):
                raise TransferError('nvars parameter of fine and coarse level needs to have the same length')
        else:
            raise TransferError('unknow type of nvars for transfer, got %s' % (nf,))
        self.identity = nf == nc
        if not self.identity:
            if not self.params.periodic or not self.params.equidist_nested:
                raise TransferError('the MI355X transfer kernels implement periodic, equidistant, nested grids')
            if any(f != 2 * c for f, c in zip(nf, nc)):
                raise TransferError(f'need coarsening by a factor of 2 per axis, got {nf} -> {nc}')
        self.ndim, self.nc = len(nc), nc[0]
        self.wi = np.ascontiguousarray(midpoint_weights(self.params.iorder))
        self.wr = np.ascontiguousarray(midpoint_weights(self.params.rorder))

    def _apply(self, fn, k, w, src, dst):
        wp = w.ctypes.data_as(C.POINTER(C.c_double)) if k > 0 else None
        Lb.check(fn(None, self.ndim, self.nc, k, wp, src.ptr, dst.ptr), None)

    def _restrict(self, fine, coarse):
        if self.identity:
            coarse[:] = fine
        else:
            self._apply(Lb.load().sdc_transfer_restrict, self.params.rorder, self.wr, fine, coarse)

    def _prolong(self, coarse, fine):
        if self.identity:
            fine[:] = coarse
        else:
            self._apply(Lb.load().sdc_transfer_prolong, self.params.iorder, self.wi, coarse, fine)

    def restrict(self, F):
        """TransferMesh.py:148-183."""
        if isinstance(F, hip_imex_mesh):
            G = hip_imex_mesh(self.coarse_prob.init)
            self._restrict(F.impl, G.impl)
            self._restrict(F.expl, G.expl)
        elif isinstance(F, hip_mesh):
            G = hip_mesh(self.coarse_prob.init)
            self._restrict(F, G)
        else:
            raise TransferError('Wrong data type for restriction, got %s' % type(F))
        return G

    def prolong(self, G):
        """TransferMesh.py:185-218."""
        if isinstance(G, hip_imex_mesh):
            F = hip_imex_mesh(self.fine_prob.init)
            self._prolong(G.impl, F.impl)
            self._prolong(G.expl, F.expl)
        elif isinstance(G, hip_mesh):
            F = hip_mesh(self.fine_prob.init)
            self._prolong(G, F)
        else:
            raise TransferError('Wrong data type for prolongation, got %s' % type(G))
        return F
