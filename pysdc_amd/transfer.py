"""Space-time transfer between two levels (FAS restriction / coarse-grid correction).

``BaseTransfer`` restates pySDC/core/base_transfer.py:36-251 against the duck-typed Level surface, so the same
host logic drives device levels (``pysdc_amd.level.Level``) and, in the CPU tests, oracle-backed levels.
``mesh_to_mesh`` is the device version of pySDC/implementations/transfer_classes/TransferMesh.py:9-218 for
periodic, equidistant, nested grids (coarsening factor 2 per axis): the interpolation matrices of
pySDC/helpers/transfer_helper.py:153-186 are applied matrix-free by ``sdc_transfer_prolong / _restrict``."""
import logging

import numpy as np

from pysdc_amd import lib as Lb
from pysdc_amd.coeffs import LagrangeApproximation
from pysdc_amd.errors import TransferError, UnlockError
from pysdc_amd.hip_mesh import device_buffer, hip_mesh, hip_imex_mesh


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
        if hasattr(self.fine, 'integrals_wanted'):
            # restrict() starts from integrate() of the fine level's state (core/base_transfer.py:120-127): its residual, which
            # the controller asks for just before, can bring those sums along in the same pass over F
            self.fine.integrals_wanted = True

    @staticmethod
    def get_transfer_matrix_Q(f_nodes, c_nodes):
        return LagrangeApproximation(c_nodes).getInterpolationMatrix(f_nodes)

    # ---- helpers: node-space combinations with a small dense matrix ------------------------------------------
    @staticmethod
    def _add_scaled(acc, coeff, field):
        """acc += coeff * field; terms with a zero coefficient leave acc as it is (x + 0*y = x for finite y), which is
        most of them when both levels share their nodes (Pcoll = Rcoll = identity)"""
        if coeff == 0.0:
            return acc
        if coeff == 1.0:
            acc += field
        elif hasattr(acc, 'iaxpy'):
            acc.iaxpy(coeff, field)
        else:
            acc += coeff * field
        return acc

    @classmethod
    def _mix(cls, matrix_row, fields):
        """sum_m matrix_row[m] * fields[m], accumulated left to right like the reference's loops.  Terms with a zero
        coefficient are left out (x + 0*y = x for finite y); a row that is a unit vector - both levels share their
        nodes - IS one of the fields (the callers only read the result)."""
        terms = [(c, f) for c, f in zip(matrix_row, fields) if c != 0.0]
        if not terms:
            return 0.0 * fields[0]
        (c0, f0), rest = terms[0], terms[1:]
        if not rest and c0 == 1.0:
            return f0
        acc = c0 * f0
        for coeff, field in rest:
            acc = cls._add_scaled(acc, coeff, field)
        return acc

    @staticmethod
    def _refresh_f(L, n, t):
        """L.f[n] = f(L.u[n], t); problems that can evaluate into an existing field fill the slab view directly"""
        into = getattr(L.prob, 'eval_f_into', None)
        if into is not None and hasattr(L.f, 'write'):
            L.f.write(n, lambda view: into(L.u[n], t, view))
        else:
            L.f[n] = L.prob.eval_f(L.u[n], t)

    @classmethod
    def _refresh_f_nodes(cls, L, nodes):
        """L.f[n] = f(L.u[n], t_n) for every n in nodes (core/base_transfer.py:207-213, :141-147): one batched evaluation when
        the problem offers it and the right-hand sides live in slabs, node by node otherwise"""
        times = {n: (L.time if n == 0 else L.time + L.dt * L.sweep.coll.nodes[n - 1]) for n in nodes}
        many = getattr(L.prob, 'eval_f_into_many', None)
        if many is not None and hasattr(L.f, 'write') and len(nodes) > 1:
            views = {}
            for n in nodes:      # (write() hands out the slab view and marks the entry valid)
                L.f.write(n, lambda view, n=n: views.__setitem__(n, view))
            many([L.u[n] for n in nodes], [times[n] for n in nodes], [views[n] for n in nodes])
            return
        for n in nodes:
            cls._refresh_f(L, n, times[n])

    def _to_coarse_nodes(self, fine_fields):
        """space restriction of every fine-node field followed by the node restriction Rcoll."""
        many = getattr(self.space_transfer, 'restrict_many', None)
        in_space = many(list(fine_fields)) if many else [self.space_transfer.restrict(x) for x in fine_fields]
        return [self._mix(self.Rcoll[n], in_space) for n in range(self.coarse.sweep.coll.num_nodes)]

    # ---- batched device paths ------------------------------------------------------------------------------------------
    # With the same nodes on both levels (Pcoll = Rcoll = identity), plain meshes in slabs and the table-driven space
    # transfer, the node loops of restrict / prolong collapse into a few launches over fields that lie one behind the other:
    # u[0..M] restricted straight into the coarse U slab, tau = R(Q_F f_F) - Q_G f_G as one difference into the TAU slab,
    # the coarse-grid correction as one difference, one batched prolongation and one in-place addition over U[1..M].  Same
    # arithmetic as the node-by-node code below (which remains for every other case), ~30 launches and the copies of the
    # temporaries fewer per iteration.
    def _batched(self):
        sp = self.space_transfer
        fine, coarse = self.fine, self.coarse
        def fused_sweeper(L):   # the sweeper's own integrate() hands out ONE buffer only on its fused path
            probe = getattr(L.sweep, '_fused', None)
            return callable(probe) and bool(probe())

        return (self.Pcoll.shape[0] == self.Pcoll.shape[1] and np.array_equal(self.Pcoll, np.eye(self.Pcoll.shape[0]))
                and np.array_equal(self.Rcoll, np.eye(self.Rcoll.shape[0]))
                and type(sp) is mesh_to_mesh and not sp.identity
                and all(hasattr(L, 'engine') and hasattr(L, '_u') and not L._view_offset()
                        and getattr(L.prob, 'dtype_u', None) is hip_mesh and getattr(L.prob, 'fused', False)
                        and fused_sweeper(L) for L in (fine, coarse))
                and type(fine).__module__ == type(coarse).__module__ == 'pysdc_amd.level')

    @staticmethod
    def _one_buffer(fields, nbytes):
        """are these M fields one behind the other in ONE allocation (what Sweeper._integrate_fused returns)?  A sweeper
        class that overrides integrate() may hand out separately allocated meshes: those take the node-by-node path"""
        try:
            base = fields[0].ptr
            return all(f.ptr == base + nbytes * k for k, f in enumerate(fields))
        except (AttributeError, IndexError, TypeError):
            return False

    def _space_batch(self, key, nfields, src_ptr, dst_ptr, accumulate=False, src_minus=None, dst_minus=None):
        """nfields fields through the space transfer; src_minus: the difference src - src_minus is transferred, dst_minus:
        result - dst_minus is stored (the differences of core/base_transfer.py:120-147 / :196-205 without fields and launches
        of their own where the tables have the nested structure, sdc_transfer_apply_nested)"""
        sp = self.space_transfer
        idx, w, width, (n_out, n_in) = sp._tab[key]
        if sp._nested.get(key):
            Lb.check(Lb.load().sdc_transfer_apply_nested(None, nfields, sp.ndim, n_out, n_in, width, idx.ptr, w.ptr, src_ptr,
                                                         src_minus, dst_ptr, dst_minus, int(accumulate)), None)
            return
        assert src_minus is None and dst_minus is None
        Lb.check(Lb.load().sdc_transfer_apply_batch_acc(None, nfields, sp.ndim, n_out, n_in, width, idx.ptr, w.ptr, src_ptr,
                                                        dst_ptr, int(accumulate)), None)

    def _restrict_batched(self):
        fine, coarse = self.fine, self.coarse
        M = fine.sweep.coll.num_nodes
        ef, ec = fine.engine, coarse.engine
        nc = ec.N
        ef.ptr(Lb.SLOT_U, 1)                      # (node fields that were put off are stored now)
        ec.ptr(Lb.SLOT_U, 1)
        coarse._lists()
        # (u[0] and U[1..M] are blocks of their own inside the engines: two calls, the node fields as one batch)
        # The start value: serial MLSDC restricts the SAME u[0] in every iteration of a step (core/base_transfer.py:113-118
        # restricts and re-evaluates it each time).  When neither the fine u[0] nor the coarse u[0] / f[0] have been written
        # since this transfer made them - change counters of both levels - and f does not depend on t, they are what this
        # call would produce bit for bit: left where they are, the evaluation counted.
        seen = getattr(self, '_u0_seen', None)
        fresh = (getattr(coarse.prob, 'rhs_autonomous', False) and seen is not None
                 and seen == (fine._tok_u0, coarse._tok_u0, coarse._tok_f0)
                 and coarse.u[0] is not None and coarse.f[0] is not None)
        if not fresh:
            self._space_batch('R', 1, ef.ptr(Lb.SLOT_U, 0), ec.ptr(Lb.SLOT_U, 0))
        self._space_batch('R', M, ef.ptr(Lb.SLOT_U, 1), ec.ptr(Lb.SLOT_U, 1))
        coarse._u.mark(range(M + 1))
        if not fresh:
            coarse._touched(Lb.SLOT_U, 0)
        coarse._touched(Lb.SLOT_U, 1)
        if fresh:
            if 'rhs' in getattr(coarse.prob, 'work_counters', {}):
                coarse.prob.work_counters['rhs']()
        else:
            self._refresh_f(coarse, 0, coarse.time)
            self._u0_seen = (fine._tok_u0, coarse._tok_u0, coarse._tok_f0)
        self._refresh_f_nodes(coarse, list(range(1, M + 1)))
        quad_coarse = coarse.sweep.integrate()    # M fields, one behind the other (checked: a user's integrate() need not)
        quad_fine = fine.sweep.integrate()
        if not self._one_buffer(quad_coarse, 8 * nc):
            packed = hip_mesh(((M * nc,), None, np.dtype('float64')), val=None)
            for k in range(M):
                dst = hip_mesh.view(packed.ptr + 8 * k * nc, (nc,), keep=packed)
                dst._axpby(1.0, hip_mesh.view(quad_coarse[k].ptr, (nc,), keep=quad_coarse[k]), 0.0, None, dst)
            quad_coarse = [packed]
        if coarse.tau[0] is None:                 # (sets the slab up and tells the engine)
            coarse._activate_tau()
        tau = hip_mesh.view(ec.ptr(Lb.SLOT_TAU, 0), (M * nc,), keep=ec)
        qc = hip_mesh.view(quad_coarse[0].ptr, (M * nc,), keep=quad_coarse[0])
        if self._one_buffer(quad_fine, 8 * ef.N) and self.space_transfer._nested.get('R'):
            # tau = R(Q_F f_F) - Q_G f_G leaves the restriction's launch as that difference
            self._space_batch('R', M, quad_fine[0].ptr, tau.ptr, dst_minus=qc.ptr)
        else:
            on_coarse = hip_mesh(((M * nc,), None, np.dtype('float64')), val=None)
            if self._one_buffer(quad_fine, 8 * ef.N):
                self._space_batch('R', M, quad_fine[0].ptr, on_coarse.ptr)
            else:
                for k in range(M):
                    self._space_batch('R', 1, quad_fine[k].ptr, on_coarse.ptr + 8 * k * nc)
            tau._axpby(1.0, on_coarse, -1.0, qc, tau)
        if fine.tau[0] is not None:               # a correction the fine level itself received from above travels down too
            carried = hip_mesh(((M * nc,), None, np.dtype('float64')), val=None)
            self._space_batch('R', M, ef.ptr(Lb.SLOT_TAU, 0), carried.ptr)
            tau._axpby(1.0, tau, 1.0, carried, tau)
        coarse._tau.mark(range(M))
        coarse._touched(Lb.SLOT_TAU, 0)
        # the snapshot the coarse-grid correction is measured against: one buffer each for u and f
        imex = getattr(coarse.prob, 'ncomp', 1) == 2
        uold = hip_mesh(((M * nc,), None, np.dtype('float64')), val=None)
        uold._axpby(1.0, hip_mesh.view(ec.ptr(Lb.SLOT_U, 1), (M * nc,), keep=ec), 0.0, None, uold)
        shape = coarse._field_shape()
        ncomp = getattr(coarse.prob, 'ncomp', 1)
        fold = None
        if coarse.prob.dtype_f in (hip_mesh, hip_imex_mesh) and all(coarse.f[n] is not None for n in range(1, M + 1)):
            # f[1..M] lie one behind the other ([node][component][point]): ONE copy for the snapshot of all of them
            fold = hip_mesh(((M * ncomp * nc,), None, np.dtype('float64')), val=None)
            fold._axpby(1.0, hip_mesh.view(ec.ptr(Lb.SLOT_F, 1), (M * ncomp * nc,), keep=ec), 0.0, None, fold)
        for n in range(1, M + 1):
            coarse.uold[n] = hip_mesh.view(uold.ptr + 8 * (n - 1) * nc, shape, keep=uold)
            if fold is None:
                coarse.fold[n] = coarse.prob.dtype_f(coarse.f[n])
            elif imex:
                at = fold.ptr + 8 * (n - 1) * ncomp * nc
                coarse.fold[n] = hip_imex_mesh.view(at, at + 8 * nc, shape, keep=fold)
            else:
                coarse.fold[n] = hip_mesh.view(fold.ptr + 8 * (n - 1) * nc, shape, keep=fold)
        self._uold_batch = uold
        coarse.status.unlocked = True

    def _prolong_batched(self):
        fine, coarse = self.fine, self.coarse
        M = fine.sweep.coll.num_nodes
        ef, ec = fine.engine, coarse.engine
        nc, nf = ec.N, ef.N
        uold = getattr(self, '_uold_batch', None)
        if uold is None or any(coarse.uold[n] is None or coarse.uold[n].ptr != uold.ptr + 8 * (n - 1) * nc for n in range(1, M + 1)):
            return False
        if self.space_transfer._nested.get('P'):
            # u_F[m] += P (u_G[m] - uold_G[m]): the difference is formed as the coarse tile is staged, the sum as the fine
            # tile is stored - one launch, no field for either
            self._space_batch('P', M, ec.ptr(Lb.SLOT_U, 1), ef.ptr(Lb.SLOT_U, 1), accumulate=True, src_minus=uold.ptr)
        else:
            diff = hip_mesh(((M * nc,), None, np.dtype('float64')), val=None)
            diff._axpby(1.0, hip_mesh.view(ec.ptr(Lb.SLOT_U, 1), (M * nc,), keep=ec), -1.0, uold, diff)
            # u_F[m] += P diff[m]: the last pass of the prolongation adds its result to the node values where they lie
            self._space_batch('P', M, diff.ptr, ef.ptr(Lb.SLOT_U, 1), accumulate=True)
        fine._touched(Lb.SLOT_U, 1)
        self._refresh_f_nodes(fine, list(range(1, M + 1)))
        return True

    def restrict(self):
        """FAS restriction (base_transfer.py:93-168): coarse node values R u_F, coarse right-hand sides
        re-evaluated there, tau = R(Q_F f_F) - Q_G f_G (+ R tau_F), and the snapshot uold / fold that the
        coarse-grid correction is measured against."""
        fine, coarse = self.fine, self.coarse
        if not fine.status.unlocked:
            raise UnlockError('fine level is still locked, cannot use data from there')
        Mf, Mc = fine.sweep.coll.num_nodes, coarse.sweep.coll.num_nodes
        cprob = coarse.prob
        self._uold_batch = None
        if self._batched() and all(fine.u[m] is not None for m in range(Mf + 1)):
            return self._restrict_batched()

        coarse.u[0] = self.space_transfer.restrict(fine.u[0])
        for n, value in enumerate(self._to_coarse_nodes([fine.u[m] for m in range(1, Mf + 1)]), start=1):
            coarse.u[n] = value
        self._refresh_f(coarse, 0, coarse.time)
        for n in range(1, Mc + 1):
            self._refresh_f(coarse, n, coarse.time + coarse.dt * coarse.sweep.coll.nodes[n - 1])

        quad_coarse = coarse.sweep.integrate()
        quad_fine_on_coarse = self._to_coarse_nodes(fine.sweep.integrate())
        for n in range(Mc):
            coarse.tau[n] = quad_fine_on_coarse[n] - quad_coarse[n]
        if fine.tau[0] is not None:  # a correction the fine level itself received from above travels down too
            carried = [self.space_transfer.restrict(fine.tau[m]) for m in range(Mf)]
            for n in range(Mc):
                for m in range(Mf):
                    coarse.tau[n] = self._add_scaled(coarse.tau[n], self.Rcoll[n, m], carried[m])

        for n in range(1, Mc + 1):
            coarse.uold[n] = cprob.dtype_u(coarse.u[n])
            coarse.fold[n] = cprob.dtype_f(coarse.f[n])
        coarse.status.unlocked = True

    def _coarse_correction(self, new_fields, old_fields):
        Mc = self.coarse.sweep.coll.num_nodes
        many = getattr(self.space_transfer, 'prolong_many', None)
        if many and all(type(new_fields[n]) is hip_mesh for n in range(1, Mc + 1)):
            # the differences go into one buffer, one behind the other, so that they are prolonged together
            size, shape = new_fields[1].size, new_fields[1].shape
            buf = hip_mesh(((Mc * size,), None, np.dtype('float64')), val=None)
            diffs = [hip_mesh.view(buf.ptr + 8 * k * size, shape, keep=buf) for k in range(Mc)]
            for k in range(Mc):
                diffs[k]._axpby(1.0, new_fields[k + 1], -1.0, old_fields[k + 1], diffs[k])
            return many(diffs)
        return [self.space_transfer.prolong(new_fields[n] - old_fields[n]) for n in range(1, Mc + 1)]

    def prolong(self):
        """coarse-grid correction (base_transfer.py:170-207): u_F += P (u_G - uold_G), then f re-evaluated."""
        fine, coarse = self.fine, self.coarse
        if not coarse.status.unlocked:
            raise UnlockError('coarse level is still locked, cannot use data from there')
        Mf, Mc = fine.sweep.coll.num_nodes, coarse.sweep.coll.num_nodes
        if self._batched() and self._prolong_batched():
            return
        delta = self._coarse_correction(coarse.u, coarse.uold)
        for n in range(1, Mf + 1):
            for m in range(Mc):
                fine.u[n] = self._add_scaled(fine.u[n], self.Pcoll[n - 1, m], delta[m])
        self._refresh_f_nodes(fine, list(range(1, Mf + 1)))

    def prolong_f(self):
        """variant that also interpolates the change of f instead of re-evaluating it (base_transfer.py:209-251)."""
        fine, coarse = self.fine, self.coarse
        if not coarse.status.unlocked:
            raise UnlockError('coarse level is still locked, cannot use data from there')
        Mf, Mc = fine.sweep.coll.num_nodes, coarse.sweep.coll.num_nodes
        du = self._coarse_correction(coarse.u, coarse.uold)
        df = self._coarse_correction(coarse.f, coarse.fold)
        for n in range(1, Mf + 1):
            for m in range(Mc):
                fine.u[n] = self._add_scaled(fine.u[n], self.Pcoll[n - 1, m], du[m])
                fine.f[n] = self._add_scaled(fine.f[n], self.Pcoll[n - 1, m], df[m])


class _SpacePars:
    """pySDC/core/space_transfer.py:8-19."""

    def __init__(self, pars):
        self.periodic = False
        self.equidist_nested = True
        self.iorder = 2
        self.rorder = 2
        for k, v in pars.items():
            setattr(self, k, v)


def _unwrapped_abscissae(grid, cols):
    """the coarse abscissae of the (ascending) column set `cols`, continued over the periodic seam: columns are
    neighbours on the circle, so wherever two consecutive entries are not adjacent indices the set has wrapped and every
    abscissa from there on belongs one period (1.0) to the left (what helpers/transfer_helper.py:72-92 computes)"""
    cols = np.asarray(cols)
    wrapped = np.concatenate(([False], np.cumsum(np.diff(cols) != 1) > 0))
    return grid[cols] - wrapped.astype(float)


def interpolation_matrix_1d(fine_grid, coarse_grid, k=2):
    """Dense (n_fine x n_coarse) interpolation matrix for periodic, equidistant, nested grids - the reference's matrix
    (helpers/transfer_helper.py:153-186), including its behaviour when the k nearest neighbours wrap around a very small
    coarse grid: even fine points coincide with a coarse point; an odd one, between the coarse points c and c + 1, gets the
    Lagrange weights of the k columns c - k/2 + 1 .. c + k/2 taken modulo the grid and SORTED, evaluated on abscissae made
    monotone across the seam."""
    from scipy.interpolate import BarycentricInterpolator

    nc = coarse_grid.size
    M = np.zeros((fine_grid.size, nc))
    M[np.arange(0, fine_grid.size, 2), np.arange(0, fine_grid.size, 2) // 2] = 1.0
    if k == 0:
        return M
    centre = np.mean(fine_grid)
    basis = np.eye(k)
    for i in range(1, fine_grid.size, 2):
        p = fine_grid[i]
        first = i // 2 - k // 2 + 1
        cols = sorted(q + nc if q < 0 else (q - nc if q > nc - 1 else q) for q in range(first, first + k))
        xs = _unwrapped_abscissae(coarse_grid, cols)
        if p > centre and not (xs[0] <= p <= xs[-1]):     # the point lies in the period the abscissae were moved out of
            xs = xs + 1
        with np.errstate(divide='ignore'):
            M[i, cols] = [BarycentricInterpolator(xs, basis[l])(p) for l in range(k)]
    return M


def interpolation_matrix_1d_bounded(fine_grid, coarse_grid, k=2):
    """Dense (n_fine x n_coarse) interpolation matrix between nested non-periodic grids (n_fine = 2 n_coarse + 1,
    zero boundary values outside): helpers/transfer_helper.py:206-244 with pad = 1 - fine points with an odd index
    coincide with a coarse point, the others interpolate k points of the coarse grid extended by its mirror
    image at either end (border_padding, :250-273); the two padding columns are cut off again."""
    from scipy.interpolate import BarycentricInterpolator

    nc = coarse_grid.size
    ext = np.empty(nc + 2)
    ext[1:-1] = coarse_grid
    ext[0] = 2 * coarse_grid[0] - coarse_grid[1]
    ext[-1] = 2 * coarse_grid[-1] - coarse_grid[-2]
    M = np.zeros((fine_grid.size, nc + 2))
    one = np.asarray([1.0] + [0.0] * (k - 1))
    for i, p in enumerate(fine_grid):
        if i % 2 == 1:
            M[i, (i - 1) // 2 + 1] = 1.0
            continue
        first = i // 2 - k // 2 + 1
        nn = sorted(q + k if q < 0 else (q - k if q > nc + 1 else q) for q in range(first, first + k))
        with np.errstate(divide='ignore'):
            M[i, nn] = np.asarray([BarycentricInterpolator(ext[nn], np.roll(one, l))(p) for l in range(k)])
    return M[:, 1:-1]


def _row_tables(M):
    """fixed-width (idx, w) rows of a dense 1-D operator, zero-padded, stored entry-major ([width][rows]: what
    sdc_transfer_apply takes).  Padded entries repeat the row's first column (weight 0): a valid, nearby address."""
    width = max(1, int(np.max(np.count_nonzero(M, axis=1))))
    idx = np.zeros((M.shape[0], width), dtype=np.int32)
    w = np.zeros((M.shape[0], width))
    for i in range(M.shape[0]):
        cols = np.nonzero(M[i])[0]
        idx[i, :] = cols[0] if len(cols) else 0
        idx[i, : len(cols)] = cols
        w[i, : len(cols)] = M[i, cols]
    return np.ascontiguousarray(idx.T), np.ascontiguousarray(w.T), width


def _rows_in_window(M, first):
    """do the non-zero columns of every row r of the dense 1-D operator M lie in first(r, width) .. + width - 1 (modulo the
    number of columns), width = the widest row?"""
    n_rows, n_cols = M.shape
    width = max(1, int(np.max(np.count_nonzero(M, axis=1))))
    if n_cols < 2 * width:
        return False
    for r in range(n_rows):
        cols = np.nonzero(M[r])[0]
        if len(cols) and np.any((cols - first(r, width)) % n_cols >= width):
            return False
    return True


class mesh_to_mesh:
    _nested = {}

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
        periodic = bool(self.params.periodic)
        if not self.identity:
            if not self.params.equidist_nested:
                raise TransferError('the MI355X transfer kernels implement equidistant, nested grids')
            if periodic and any(f != 2 * c for f, c in zip(nf, nc)):
                raise TransferError(f'need coarsening by a factor of 2 per axis, got {nf} -> {nc}')
            if not periodic and (len(nf) != 1 or nf[0] != 2 * nc[0] + 1):
                raise TransferError('non-periodic transfer is built for 1-D grids with n_fine = 2 n_coarse + 1 '
                                    f'(dirichlet-zero), got {nf} -> {nc}')
        self.ndim, self.nc, self.nf = len(nc), nc[0], nf[0]
        if not self.identity:
            first = 0 if periodic else 1  # TransferMesh.py:62-67: dirichlet grids start at dx
            fine_grid = np.array([(j + first) * fine_prob.dx for j in range(nf[0])])
            coarse_grid = np.array([(j + first) * coarse_prob.dx for j in range(nc[0])])
            build = interpolation_matrix_1d if periodic else interpolation_matrix_1d_bounded
            P = build(fine_grid, coarse_grid, k=self.params.iorder)
            restr_factor = 0.5 if self.params.rorder > 0 else 1.0
            Pr = P if self.params.iorder == self.params.rorder else build(
                fine_grid, coarse_grid, k=self.params.rorder)
            R = restr_factor * Pr.T
            self._tab = {}
            for key, Mx in (('P', P), ('R', R)):
                idx, w, width = _row_tables(Mx)
                self._tab[key] = (device_buffer(idx), device_buffer(w), width, Mx.shape)
            # nested periodic grids in 3-D: do all entries of every row lie in that row's window?  (what the one-launch
            # transfers of sdc_transfer_apply_nested assume; everything else takes the separable passes)
            self._nested = {'P': periodic and self.ndim == 3 and _rows_in_window(P, lambda r, wd: r // 2 - wd // 2 + 1),
                            'R': periodic and self.ndim == 3 and _rows_in_window(R, lambda r, wd: 2 * r - 1)}

    def _apply(self, key, src, dst):
        idx, w, width, (n_out, n_in) = self._tab[key]
        if self._nested.get(key):
            Lb.check(Lb.load().sdc_transfer_apply_nested(None, 1, self.ndim, n_out, n_in, width, idx.ptr, w.ptr, src.ptr, None,
                                                         dst.ptr, None, 0), None)
            return
        Lb.check(Lb.load().sdc_transfer_apply(None, self.ndim, n_out, n_in, width, idx.ptr, w.ptr, src.ptr, dst.ptr), None)

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

    def _many(self, key, fields, out_init):
        """several plain fields that lie one behind the other in memory (slab views U[1..M], a batch of integrals): one
        launch per axis for all of them; the results are views into one buffer.  None when that does not apply."""
        if self.identity or len(fields) < 2 or not all(type(f) is hip_mesh for f in fields):
            return None
        step = 8 * fields[0].size
        ptrs = [f.ptr for f in fields]
        if any(p != ptrs[0] + k * step for k, p in enumerate(ptrs)):
            return None
        idx, w, width, (n_out, n_in) = self._tab[key]
        shape = out_init[0] if not np.isscalar(out_init[0]) else (int(out_init[0]),)
        osize = int(np.prod(shape))
        buf = hip_mesh(((len(fields) * osize,), None, np.dtype('float64')), val=None)
        if self._nested.get(key):
            Lb.check(Lb.load().sdc_transfer_apply_nested(None, len(fields), self.ndim, n_out, n_in, width, idx.ptr, w.ptr,
                                                         ptrs[0], None, buf.ptr, None, 0), None)
        else:
            Lb.check(Lb.load().sdc_transfer_apply_batch(None, len(fields), self.ndim, n_out, n_in, width, idx.ptr, w.ptr,
                                                        ptrs[0], buf.ptr), None)
        return [hip_mesh.view(buf.ptr + 8 * k * osize, shape, keep=buf) for k in range(len(fields))]

    def restrict_many(self, fields):
        out = self._many('R', fields, self.coarse_prob.init)
        return out if out is not None else [self.restrict(f) for f in fields]

    def prolong_many(self, fields):
        out = self._many('P', fields, self.fine_prob.init)
        return out if out is not None else [self.prolong(f) for f in fields]

    def restrict(self, F):
        """TransferMesh.py:148-183."""
        if isinstance(F, hip_imex_mesh):
            G = hip_imex_mesh(self.coarse_prob.init, val=None)
            self._restrict(F.impl, G.impl)
            self._restrict(F.expl, G.expl)
        elif isinstance(F, hip_mesh):
            G = hip_mesh(self.coarse_prob.init, val=None)
            self._restrict(F, G)
        else:
            raise TransferError('Wrong data type for restriction, got %s' % type(F))
        return G

    def prolong(self, G):
        """TransferMesh.py:185-218."""
        if isinstance(G, hip_imex_mesh):
            F = hip_imex_mesh(self.fine_prob.init, val=None)
            self._prolong(G.impl, F.impl)
            self._prolong(G.expl, F.expl)
        elif isinstance(G, hip_mesh):
            F = hip_mesh(self.fine_prob.init, val=None)
            self._prolong(G, F)
        else:
            raise TransferError('Wrong data type for prolongation, got %s' % type(G))
        return F


class _fourier_transfer:
    """Common part of the two FFT-based space transfers for periodic grids: restriction by injection
    (``F[::ratio]`` per axis), prolongation by copying the coarse spectrum into the low modes of a fine spectrum
    (``sdc_fft_prolong``; the transforms run in the two levels' engines)."""

    ndim = None

    def __init__(self, fine_prob, coarse_prob, params):
        self.params = _SpacePars(params)
        self.logger = logging.getLogger('space-transfer')
        self.fine_prob, self.coarse_prob = fine_prob, coarse_prob
        nf, nc = self._shape(fine_prob.nvars), self._shape(coarse_prob.nvars)
        if len(nf) != self.ndim or len(nc) != self.ndim:
            raise TransferError(f'{type(self).__name__} transfers between {self.ndim}-D grids, got {nf} -> {nc}')
        if len(set(nf)) != 1 or len(set(nc)) != 1:
            raise TransferError(f'need square grids, got {nf} -> {nc}')
        if getattr(fine_prob, 'view_offset', 0) or getattr(coarse_prob, 'view_offset', 0):
            raise TransferError('Fourier transfer needs periodic problems on both levels')
        self.nf, self.nc = nf[0], nc[0]
        self.ratio = int(self.nf / self.nc)
        if self.ratio * self.nc != self.nf:
            raise TransferError(f'fine grid is not a multiple of the coarse grid: {nf} -> {nc}')
        self._tab = None

    @staticmethod
    def _shape(nvars):
        return (int(nvars),) if np.isscalar(nvars) else tuple(int(v) for v in nvars)

    def _factor(self):
        raise NotImplementedError

    def _inject(self, fine, coarse):
        if self._tab is None:
            idx = (np.arange(self.nc, dtype=np.int32) * self.ratio).reshape(self.nc, 1)
            self._tab = (device_buffer(idx), device_buffer(np.ones((self.nc, 1))))
        idx, w = self._tab
        Lb.check(Lb.load().sdc_transfer_apply(None, self.ndim, self.nc, self.nf, 1, idx.ptr, w.ptr, fine.ptr, coarse.ptr),
                 None)

    def _pad(self, coarse, fine):
        ec, ef = self.coarse_prob.engine, self.fine_prob.engine
        Lb.check(ef.lib.sdc_fft_prolong(ec.ctx, ef.ctx, coarse.ptr, fine.ptr, float(self._factor())), ef.ctx)

    def restrict(self, F):
        if isinstance(F, hip_imex_mesh):
            G = hip_imex_mesh(self.coarse_prob.init, val=None)
            self._inject(F.impl, G.impl)
            self._inject(F.expl, G.expl)
        elif isinstance(F, hip_mesh):
            G = hip_mesh(self.coarse_prob.init, val=None)
            self._inject(F, G)
        else:
            raise TransferError('Unknown data type, got %s' % type(F))
        return G

    def prolong(self, G):
        if isinstance(G, hip_imex_mesh):
            F = hip_imex_mesh(self.fine_prob.init, val=None)
            self._pad(G.impl, F.impl)
            self._pad(G.expl, F.expl)
        elif isinstance(G, hip_mesh):
            F = hip_mesh(self.fine_prob.init, val=None)
            self._pad(G, F)
        else:
            raise TransferError('Unknown data type, got %s' % type(G))
        return F


class mesh_to_mesh_fft(_fourier_transfer):
    """pySDC/implementations/transfer_classes/TransferMesh_FFT.py:5-57 (1-D, rfft / irfft, factor = ratio)."""

    ndim = 1

    def _factor(self):
        return self.ratio


class mesh_to_mesh_fft2d(_fourier_transfer):
    """pySDC/implementations/transfer_classes/TransferMesh_FFT2D.py:8-77 (2-D, fft2 / real part of ifft2, the
    reference's factor 2 * ratio).  imex data is transferred component by component (the reference's imex branch
    of prolong cannot run: it multiplies the shape tuple by the communicator, :96)."""

    ndim = 2

    def _factor(self):
        return self.ratio * 2


class mesh_to_mesh_fft3d(_fourier_transfer):
    """The same in 3-D: restriction by injection, prolongation by copying the eight corner blocks of fftn(G) into a zero fine
    spectrum and taking the real part of its inverse transform.  The reference's Fourier transfer in three dimensions
    (transfer_classes/TransferMesh_MPIFFT.py:51-136, `fft_to_fft`) delegates the padded transform to mpi4py_fft, which is
    not available here; this class extends TransferMesh_FFT2D.py:58-77 by one axis instead - a field that does not depend on
    one axis is prolonged plane by plane exactly as mesh_to_mesh_fft2d does it, including its factor (2 * ratio there, hence
    2 * ratio^2 here; both are the interpolating ratio^ndim for ratio 2)."""

    ndim = 3

    def _factor(self):
        return 2 * self.ratio**2
