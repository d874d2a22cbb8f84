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


def _continue_periodic_array(arr, nn):
    """helpers/transfer_helper.py:72-92."""
    nn = np.asarray(nn)
    d_nn = nn[1:] - nn[:-1]
    if np.all(d_nn == np.ones(nn.shape[0] - 1)):
        return arr[nn]
    cont_arr = [arr[nn[0]]]
    shift = 0.0
    for n, d in zip(nn[1:], d_nn):
        if d != 1:
            shift = -1
        cont_arr.append(arr[n] + shift)
    return np.asarray(cont_arr)


def interpolation_matrix_1d(fine_grid, coarse_grid, k=2):
    """Dense (n_fine x n_coarse) interpolation matrix for periodic, equidistant, nested grids: the same
    construction, row by row, as helpers/transfer_helper.py:153-186 (including its behaviour when the k nearest
    neighbours wrap around a very small coarse grid), so the device operator equals the reference's matrix."""
    from scipy.interpolate import BarycentricInterpolator

    M = np.zeros((fine_grid.size, coarse_grid.size))
    for i, p in enumerate(fine_grid):
        if i % 2 == 0:
            M[i, int(i / 2)] = 1.0
            continue
        if k == 0:
            continue
        nn = []
        cpos, offset = int(i / 2), int(k / 2)
        for j in range(k):
            nn.append(cpos - offset + 1 + j)
            if nn[-1] < 0:
                nn[-1] += coarse_grid.size
            elif nn[-1] > coarse_grid.size - 1:
                nn[-1] -= coarse_grid.size
        nn = sorted(nn)
        cont_arr = np.array(_continue_periodic_array(coarse_grid, nn), dtype=float)
        if p > np.mean(fine_grid) and not (cont_arr[0] <= p <= cont_arr[-1]):
            cont_arr += 1
        one = np.asarray([1.0] + [0.0] * (k - 1))
        with np.errstate(divide='ignore'):
            M[i, nn] = np.asarray([BarycentricInterpolator(cont_arr, np.roll(one, l))(p) for l in range(k)])
    return M


def _row_tables(M):
    """fixed-width (idx, w) rows of a dense 1-D operator, zero-padded."""
    width = max(1, int(np.max(np.count_nonzero(M, axis=1))))
    idx = np.zeros((M.shape[0], width), dtype=np.int32)
    w = np.zeros((M.shape[0], width))
    for i in range(M.shape[0]):
        cols = np.nonzero(M[i])[0]
        idx[i, : len(cols)] = cols
        w[i, : len(cols)] = M[i, cols]
    return idx, w, width


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
        self.ndim, self.nc, self.nf = len(nc), nc[0], nf[0]
        if not self.identity:
            import torch

            fine_grid = np.array([j * fine_prob.dx for j in range(nf[0])])
            coarse_grid = np.array([j * coarse_prob.dx for j in range(nc[0])])
            P = interpolation_matrix_1d(fine_grid, coarse_grid, k=self.params.iorder)
            restr_factor = 0.5 if self.params.rorder > 0 else 1.0
            Pr = P if self.params.iorder == self.params.rorder else interpolation_matrix_1d(
                fine_grid, coarse_grid, k=self.params.rorder)
            R = restr_factor * Pr.T
            self._tab = {}
            for key, Mx in (('P', P), ('R', R)):
                idx, w, width = _row_tables(Mx)
                self._tab[key] = (torch.from_numpy(idx).cuda(), torch.from_numpy(w).cuda(), width, Mx.shape)

    def _apply(self, key, src, dst):
        idx, w, width, (n_out, n_in) = self._tab[key]
        Lb.check(Lb.load().sdc_transfer_apply(None, self.ndim, n_out, n_in, width, idx.data_ptr(), w.data_ptr(),
                                              src.ptr, dst.ptr), None)

    def _restrict(self, fine, coarse):
        if self.identity:
            coarse[:] = fine
        else:
            self._apply('R', fine, coarse)

    def _prolong(self, coarse, fine):
        if self.identity:
            fine[:] = coarse
        else:
            self._apply('P', coarse, fine)

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
