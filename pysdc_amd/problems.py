"""Problem classes with the reference's plug-in surface (SURVEY.md 8b "Problem surface"):
``init, dtype_u, dtype_f, u_init, f_init, eval_f(u, t), solve_system(rhs, factor, u0, t), u_exact(t),
work_counters, nvars, dx, get_default_sweeper_class()`` - computing on the MI355X through libsdcmi.

Mirrors (behaviour, parameter names, error messages):
  GenericNDimFinDiff  /root/reference/pySDC/implementations/problem_classes/generic_ND_FD.py:17-264
  heatNd_unforced/_forced  .../HeatEquation_ND_FD.py:9-230
  advectionNd         .../AdvectionEquation_ND_FD.py:8-132
plus ``advectiondiffusionNd_imex`` (BASELINE config 3; the reference has no ND finite-difference IMEX
advection-diffusion class)."""
import logging

import numpy as np

from pysdc_amd import fd
from pysdc_amd import lib as L
from pysdc_amd.errors import ProblemError, ReadOnlyError
from pysdc_amd.hip_mesh import hip_mesh, hip_imex_mesh


class WorkCounter:
    """pySDC/core/problem.py:16-40."""

    def __init__(self):
        self.niter = 0

    def __call__(self, *args, **kwargs):
        self.niter += 1

    def decrement(self):
        self.niter -= 1

    def __str__(self):
        return f'{self.niter}'


class _Params(dict):
    __getattr__ = dict.get


class Problem:
    """pySDC/core/problem.py:43-215 (interface part)."""

    dtype_u = hip_mesh
    dtype_f = hip_mesh
    logger = logging.getLogger('problem')
    fused = False  # set by classes whose sweep can run as one fused engine call

    def __init__(self, init):
        object.__setattr__(self, '_readonly', set())
        self.work_counters = {}
        self.init = init
        self.params = _Params()
        self._engine = None

    def _makeAttributeAndRegister(self, *names, localVars=None, readOnly=False):
        for name in names:
            object.__setattr__(self, name, localVars[name])
            self.params[name] = localVars[name]
            if readOnly:
                self._readonly.add(name)

    def __setattr__(self, name, value):
        if name in getattr(self, '_readonly', ()):
            raise ReadOnlyError(name)
        object.__setattr__(self, name, value)

    @property
    def u_init(self):
        return self.dtype_u(self.init)

    @property
    def f_init(self):
        return self.dtype_f(self.init)

    # outputs of eval_f / solve_system: every element is written by the engine, no zero fill needed
    def _out_u(self):
        return self.dtype_u(self.init, val=None)

    def _out_f(self):
        return self.dtype_f(self.init, val=None)

    def eval_f(self, u, t):
        raise NotImplementedError('ERROR: problem has to implement eval_f(self, u, t)')

    def solve_system(self, rhs, factor, u0, t):
        raise NotImplementedError('ERROR: problem has to implement solve_system(self, rhs, factor, u0, t)')

    def u_exact(self, t):
        raise NotImplementedError('ERROR: problem has to implement u_exact(self, t)')

    # ---- output through the stock file hook (core/problem.py:84-97; hooks/log_solution.py:207-282 LogToFile) ---------
    # The hook asks the problem for an output file and for the host form of a solution.  The file format is pySDC's own
    # (helpers/fieldsIO.py), so it comes from the pySDC installation the hook itself comes from; the solution is mirrored
    # to host memory here - one device-to-host copy per logged time, nothing on the sweep path.
    def setUpFieldsIO(self):
        pass

    def getOutputFile(self, fileName):
        from pySDC.helpers.fieldsIO import Rectilinear, Scalar  # (wherever the stock hook is importable, this is too)

        shape = self.init[0]
        shape = (int(shape),) if np.isscalar(shape) else tuple(shape)
        grid = getattr(self, 'xvalues', None)
        if grid is not None and all(len(grid) == n for n in shape):
            out = Rectilinear(np.float64, fileName=fileName)
            out.setHeader(nVar=1, coords=[np.asarray(grid, dtype=float)] * len(shape))
        else:  # no grid to describe (ensembles of scalar equations): a flat record per time
            out = Scalar(np.float64, fileName=fileName)
            out.setHeader(nVar=int(np.prod(shape)))
        out.initialize()
        return out

    def processSolutionForOutput(self, u):
        host = u.get() if hasattr(u, 'get') else np.asarray(u)
        grid = getattr(self, 'xvalues', None)
        if grid is not None and all(len(grid) == n for n in host.shape):
            return np.ascontiguousarray(host, dtype=np.float64)[None, ...]   # (nVar, *grid)
        return np.ascontiguousarray(host, dtype=np.float64).reshape(-1)

    # ---- engine binding: the level's SweepEngine also serves eval_f / solve_system ------------------------
    ncomp = 1

    def bind_engine(self, engine):
        self._engine = engine
        self.configure_engine(engine)

    def configure_engine(self, engine):
        pass

    @property
    def engine(self):
        if self._engine is None:
            from pysdc_amd.engine import SweepEngine

            self.bind_engine(SweepEngine(getattr(self, 'engine_nvars', self.nvars), 1, self.ncomp))
        return self._engine


def _as_axis_tuple(value, what, ndim=None):
    """int -> one entry per axis, tuple -> itself; anything else is a ProblemError (messages as in generic_ND_FD.py:99-132,
    they are observable)"""
    if type(value) is int:
        return (value,) * (ndim or 1)
    if type(value) is tuple:
        return value
    raise ProblemError(f'{what} should be either tuple or int')


# what each boundary type asks of the number of points per axis: (predicate on nvar that must hold, message)
_POINTS_RULE = {
    'periodic': (lambda nv: nv % 2 == 0, 'the setup requires nvars = 2^p per dimension'),
    'dirichlet-zero': (lambda nv: nv % 2 == 1, 'setup requires nvars = 2^p - 1'),
}


def _grid_spec(nvars, freq, bc):
    """grid description of the finite-difference problems, normalised: (nvars tuple, freq tuple, bc).  Own restatement of
    the checks the reference makes on its constructor arguments (generic_ND_FD.py:99-132): up to three equal axes, one
    frequency per axis, even frequencies and point counts on periodic grids, odd point counts with dirichlet-zero; a 1-D
    grid with freq = -1 (Gaussian start value) is periodic whatever bc says."""
    grid = _as_axis_tuple(nvars, 'nvars')
    if type(freq) not in (int, tuple):
        raise ProblemError('freq should be either tuple or int')
    if len(grid) > 3:
        raise ProblemError(f'can work with up to three dimensions, got {len(grid)}')
    modes = _as_axis_tuple(freq, 'freq', len(grid))
    if len(modes) != len(grid):
        raise ProblemError(f'len(freq)={len(modes)}, different to ndim={len(grid)}')
    if len(grid) == 1 and -1 in modes[:1]:
        bc = 'periodic'
    elif bc == 'periodic' and any(f % 2 for f in modes):
        raise ProblemError('need even number of frequencies due to periodic BCs')
    ok, msg = _POINTS_RULE.get(bc, (lambda nv: True, ''))
    if not all(ok(nv) for nv in grid):
        raise ProblemError(msg)
    if len(set(grid)) > 1:
        raise ProblemError('need a square domain, got %s' % (grid,))
    return grid, modes, bc


class GenericNDimFinDiff(Problem):
    """du/dt = A u, A a periodic finite-difference operator applied matrix-free on the device."""

    fused = True

    def __init__(self, nvars=512, coeff=1.0, derivative=1, freq=2, stencil_type='center', order=2, lintol=1e-12,
                 liniter=10000, solver_type='direct', bc='periodic', bcParams=None, use_bcParams=False):
        nvars, freq, bc = _grid_spec(nvars, freq, bc)
        ndim = len(nvars)
        # Boundaries: 'periodic'; or Dirichlet / Neumann ends in any mix - one string for both ends ('dirichlet-zero',
        # 'dirichlet', 'neumann', 'neumann-zero', ...: an end is what its string contains, helpers/problem_helper.py:157) or a
        # pair such as ('dirichlet', 'neumann') (generic_ND_FD.py:50-54).  bcParams is accepted and - exactly like the
        # reference, whose constructor neither passes it on nor keeps the boundary vector (generic_ND_FD.py:140-148:
        # `self.A, _ = get_finite_difference_matrix(..., bc=bc)`) - has no effect on the operator: boundary values and
        # derivatives are zero.  The helper underneath (pysdc_amd.fd.bounded_operator_rows) implements all of them.
        # An unknown boundary type ends in get_1d_grid's NotImplementedError, as there (helpers/problem_helper.py:267).
        if solver_type not in ('direct', 'CG', 'GMRES'):
            raise ProblemError(f'solver type "{solver_type}" not known in generic advection-diffusion implementation!')
        super().__init__(init=(nvars[0] if ndim == 1 else nvars, None, np.dtype('float64')))
        dx, xvalues = fd.grid_1d(size=nvars[0], bc=bc, left_boundary=0.0, right_boundary=1.0)
        self._stencil = fd.periodic_operator_stencil(derivative, order, stencil_type, dx, coeff)
        self.xvalues = xvalues
        # dirichlet-zero: the engine works on the odd extension of length 2(n+1) per axis.  1-D: the interior is a
        # contiguous part of it, so level fields ARE views into extended slab fields (include/sdcmi.h: sdc_odd_mirror).
        # 2-D / 3-D: the interior is strided inside the extension, so fields stay compact and are packed into / extracted
        # from extension-sized scratch around eval_f and solve_system (sdc_odd_extend / sdc_odd_extract); the sweep then
        # runs node by node on datatype operations (fused = False).
        # every other bounded grid - dirichlet-zero with stencils of order >= 4, Neumann ends, mixed ends, one-sided
        # stencils: the reference rewrites the rows next to the boundary (helpers/problem_helper.py:143-224: shifted
        # one-sided stencils, a Neumann end eliminated through a one-sided first derivative) - a non-symmetric banded matrix per axis.  Fields stay compact (the
        # first n^ndim values of slab fields sized (n+1)^ndim), eval_f applies the row table axis by axis, the solve is
        # GMRES (to round-off for 'direct', the user's tolerance and counts for 'GMRES' / 'CG'), sweeps run node by node.
        bounded = bc != 'periodic'
        odd_extension = bc == 'dirichlet-zero' and derivative == 2 and stencil_type == 'center' and order == 2
        self.banded = bounded and not odd_extension
        self.view_offset = 1 if (odd_extension and ndim == 1) else 0
        self.odd_nd = odd_extension and ndim > 1
        self.engine_nvars = (2 * (nvars[0] + 1),) * ndim if odd_extension else nvars
        if self.banded:
            self.engine_nvars = (nvars[0] + nvars[0] % 2,) * ndim    # (slab fields: an even number of points per axis, >= n)
            # use_bcParams (an extension, default off = the reference's behaviour described above): the rows are built WITH
            # bcParams and the boundary vector is kept: f(u) = coeff (D u + b), (I - factor coeff D) u = rhs + factor coeff b.
            # 1-D only - in more dimensions the reference's b covers two corners of the grid (its own TODO, :226)
            if use_bcParams and ndim > 1:
                raise NotImplementedError('boundary data (bcParams) are defined for one dimension (helpers/problem_helper.py:226)')
            rows, bvec = fd.bounded_operator_rows(derivative, order, stencil_type, dx, nvars[0], bc,
                                                  bcParams if use_bcParams else None)
            self._rows = fd.rows_to_table(rows, coeff)
            self._bvec_host = coeff * bvec if (use_bcParams and np.any(bvec)) else None
        # banded levels sweep inside the engine (sdc_sweep: one gather for all nodes, then per node right-hand side, GMRES /
        # CG solve with the old node value as the guess, operator application [+ forcing profile] - the reference's node loop
        # as device launches without a host round trip per datatype operation)
        # (odd extensions in 2-D / 3-D sweep inside the engine too since round 4: sdc_set_odd_interior makes sdc_eval_f /
        # sdc_solve take the compact interior fields and go through the extension themselves)
        self._scratch = None
        self._makeAttributeAndRegister('nvars', 'stencil_type', 'order', 'bc', localVars=locals(), readOnly=True)
        self._makeAttributeAndRegister('freq', 'lintol', 'liniter', 'solver_type', localVars=locals())
        # 'direct' and 'CG' (generic_ND_FD.py:238-262) end in the SAME system
        # (I - factor*A) u = rhs.  'direct' is the exact solve in Fourier space; 'CG' runs the reference's conjugate
        # gradients on the device (x0 = previous node value, rtol = lintol, iterations counted like the reference's
        # callback, generic_ND_FD.py:158-159,252-260).
        if solver_type in ('CG', 'GMRES'):
            self.work_counters[solver_type] = _DeviceCounter(self, solver_type)

    @property
    def ndim(self):
        return len(self.nvars)

    @property
    def dx(self):
        return self.xvalues[1] - self.xvalues[0]

    @property
    def grids(self):
        x = self.xvalues
        if self.ndim == 1:
            return x
        if self.ndim == 2:
            return x[None, :], x[:, None]
        return x[None, :, None], x[:, None, None], x[None, None, :]

    @classmethod
    def get_default_sweeper_class(cls):
        from pysdc_amd.sweepers import generic_implicit

        return generic_implicit

    def configure_engine(self, engine):
        if self.banded:
            engine.set_banded_operator(*self._rows)
        else:
            engine.set_stencil(0, *self._stencil)
        if self.odd_nd:
            engine.set_odd_interior(self.nvars[0])
        if self.solver_type in ('CG', 'GMRES'):
            engine.set_solver(self.solver_type, self.lintol, self.liniter)

    # ---- odd-extension staging for fields that are not slab views (dirichlet-zero) ----------------------------
    def _ext(self, k):
        if self._scratch is None:
            n2 = int(np.prod(self.engine_nvars))
            self._scratch = [hip_mesh(((n2,), None, np.dtype('float64'))) for _ in range(3)]
        return self._scratch[k]

    def _stage_in(self, u, k):
        """pointer the engine can read: the field itself (periodic, banded, compact interior of an odd extension in 2-D / 3-D
        - the engine packs and extracts those itself) or, in 1-D, its odd extension in scratch k"""
        if self.banded or self.odd_nd:
            return u.ptr
        if not self.view_offset:
            return u.ptr
        e = self._ext(k)
        n = self.nvars[0]
        self.engine.vec_copy(n, u.ptr, e.ptr + 8)
        L.check(self.engine.lib.sdc_odd_mirror(self.engine.ctx, e.ptr, n), self.engine.ctx)
        return e.ptr

    def _stage_out(self, k, dst):
        if self.view_offset:
            self.engine.vec_copy(self.nvars[0], self._ext(k).ptr + 8, dst.ptr)

    def _out_ptr(self, k, dst):
        return self._ext(k).ptr if self.view_offset else dst.ptr   # (banded, odd extension in 2-D / 3-D: in place, nothing to stage)

    def _boundary_data(self):
        """coeff * b on the device (use_bcParams), or None"""
        if getattr(self, '_bvec_host', None) is None:
            return None
        if getattr(self, '_bvec_dev', None) is None:
            self._bvec_dev = self._from_host(self._bvec_host.reshape(self.nvars))
        return self._bvec_dev

    def eval_f(self, u, t):
        f = self._out_f()
        self.engine.eval_f(self._stage_in(u, 0), 0.0, self._out_ptr(1, f))
        self._stage_out(1, f)
        b = self._boundary_data()
        if b is not None:
            f._axpby(1.0, f, 1.0, b, f)
        return f

    def solve_system(self, rhs, factor, u0, t):
        sol = self._out_u()
        guess = self._stage_in(u0, 2) if self.solver_type in ('CG', 'GMRES') and u0 is not None else None
        b = self._boundary_data()
        if b is not None:
            shifted = self._out_u()
            shifted._axpby(1.0, rhs, float(factor), b, shifted)
            rhs = shifted
        self.engine.solve(self._stage_in(rhs, 0), float(factor), self._out_ptr(1, sol), guess)
        self._stage_out(1, sol)
        return sol

    def _from_host(self, values):
        sol = self._out_u()
        sol[:] = values
        return sol


class heatNd_unforced(GenericNDimFinDiff):
    def __init__(self, nvars=512, nu=0.1, freq=2, stencil_type='center', order=2, lintol=1e-12, liniter=10000,
                 solver_type='direct', bc='periodic', sigma=6e-2):
        super().__init__(nvars, nu, 2, freq, stencil_type, order, lintol, liniter, solver_type, bc)
        self._makeAttributeAndRegister('nu', localVars=locals(), readOnly=True)
        self._makeAttributeAndRegister('sigma', localVars=locals())

    def u_exact(self, t, **kwargs):
        """HeatEquation_ND_FD.py:84-132 verbatim in its arithmetic, including the 3-D expression whose middle
        term of rho has no '/ dx**2' (SURVEY.md F8); evaluated on the host, returned on the device."""
        ndim, freq, nu, sigma, dx = self.ndim, self.freq, self.nu, self.sigma, self.dx
        if ndim == 1:
            x = self.grids
            rho = (2.0 - 2.0 * np.cos(np.pi * freq[0] * dx)) / dx**2
            if freq[0] > 0:
                sol = np.sin(np.pi * freq[0] * x) * np.exp(-t * nu * rho)
            else:
                sol = np.exp(-0.5 * ((x - 0.5) / sigma) ** 2) * np.exp(-t * nu * rho)
        elif ndim == 2:
            rho = (2.0 - 2.0 * np.cos(np.pi * freq[0] * dx)) / dx**2 + (
                2.0 - 2.0 * np.cos(np.pi * freq[1] * dx)
            ) / dx**2
            x, y = self.grids
            sol = np.sin(np.pi * freq[0] * x) * np.sin(np.pi * freq[1] * y) * np.exp(-t * nu * rho)
        else:
            rho = (
                (2.0 - 2.0 * np.cos(np.pi * freq[0] * dx)) / dx**2
                + (2.0 - 2.0 * np.cos(np.pi * freq[1] * dx))
                + (2.0 - 2.0 * np.cos(np.pi * freq[2] * dx)) / dx**2
            )
            x, y, z = self.grids
            sol = (np.sin(np.pi * freq[0] * x) * np.sin(np.pi * freq[1] * y) * np.sin(np.pi * freq[2] * z)
                   * np.exp(-t * nu * rho))
        return self._from_host(sol)


class heatNd_forced(heatNd_unforced):
    dtype_f = hip_imex_mesh
    ncomp = 2
    expl_kind = L.EXPL_FORCING

    def _profile(self):
        g = self.grids
        if self.ndim == 1:
            return np.sin(np.pi * self.freq[0] * g)
        p = np.sin(np.pi * self.freq[0] * g[0])
        for i in range(1, self.ndim):
            p = p * np.sin(np.pi * self.freq[i] * g[i])
        return np.broadcast_to(p, self.nvars)

    def forcing_g(self, t):
        """time factor of the forcing term, HeatEquation_ND_FD.py:176-204."""
        return self.nu * np.pi**2 * sum([freq**2 for freq in self.freq]) * np.cos(t) - np.sin(t)

    def configure_engine(self, engine):
        super().configure_engine(engine)
        if self.view_offset or self.odd_nd:  # the sine profile continued over the odd extension is the same sine
            x = np.arange(self.engine_nvars[0]) * self.dx   # extension index e <-> x = e dx (interior point i is e = i + 1)
            if self.ndim == 1:
                g = [x]
            elif self.ndim == 2:
                g = [x[None, :], x[:, None]]                 # orientation of self.grids (generic_ND_FD.py:172-180)
            else:
                g = [x[None, :, None], x[:, None, None], x[None, None, :]]
            p = np.sin(np.pi * self.freq[0] * g[0])
            for i in range(1, self.ndim):
                p = p * np.sin(np.pi * self.freq[i] * g[i])
            engine.set_forcing_profile(np.broadcast_to(p, self.engine_nvars))
        elif self.banded:       # compact fields: the profile's n^ndim values lead the engine's (slab-sized) field
            flat = np.zeros(int(np.prod(self.engine_nvars)))
            flat[: int(np.prod(self.nvars))] = np.asarray(self._profile(), dtype=float).reshape(-1)
            engine.set_forcing_profile(flat.reshape(self.engine_nvars))
        else:
            engine.set_forcing_profile(self._profile())

    @classmethod
    def get_default_sweeper_class(cls):
        from pysdc_amd.sweepers import imex_1st_order

        return imex_1st_order

    def eval_f(self, u, t):
        f = self._out_f()
        if self.banded:   # the operator by its row table; the forcing is profile(x) * g(t) (HeatEquation_ND_FD.py:162-204)
            self.engine.eval_f(u.ptr, 0.0, f.impl.ptr)
            if getattr(self, '_profile_dev', None) is None:
                self._profile_dev = self._from_host(self._profile())
            f.expl._axpby(float(self.forcing_g(t)), self._profile_dev, 0.0, self._profile_dev, f.expl)
            return f
        if self.odd_nd:   # (compact fields in and out: the engine goes through the extension itself)
            self.engine.eval_f(u.ptr, float(self.forcing_g(t)), f.impl.ptr, f.expl.ptr)
        elif self.view_offset:
            self.engine.eval_f(self._stage_in(u, 0), float(self.forcing_g(t)), self._ext(1).ptr, self._ext(2).ptr)
            n = self.nvars[0]
            self.engine.vec_copy(n, self._ext(1).ptr + 8, f.impl.ptr)
            self.engine.vec_copy(n, self._ext(2).ptr + 8, f.expl.ptr)
        else:
            self.engine.eval_f(u.ptr, float(self.forcing_g(t)), f.impl.ptr, f.expl.ptr)
        return f

    def u_exact(self, t):
        return self._from_host(self._profile() * np.cos(t))


class advectionNd(GenericNDimFinDiff):
    def __init__(self, nvars=512, c=1.0, freq=2, stencil_type='center', order=2, lintol=1e-12, liniter=10000,
                 solver_type='direct', bc='periodic', sigma=6e-2):
        super().__init__(nvars, -c, 1, freq, stencil_type, order, lintol, liniter, solver_type, bc)
        self._makeAttributeAndRegister('c', localVars=locals(), readOnly=True)
        self._makeAttributeAndRegister('sigma', localVars=locals())

    def u_exact(self, t, **kwargs):
        ndim, freq, c, sigma = self.ndim, self.freq, self.c, self.sigma
        if ndim == 1:
            x = self.grids
            if freq[0] >= 0:
                sol = np.sin(np.pi * freq[0] * (x - c * t))
            else:
                sol = np.exp(-0.5 * (((x - (c * t)) % 1.0 - 0.5) / sigma) ** 2)
        elif ndim == 2:
            x, y = self.grids
            sol = np.sin(np.pi * freq[0] * (x - c * t)) * np.sin(np.pi * freq[1] * (y - c * t))
        else:
            x, y, z = self.grids
            sol = (np.sin(np.pi * freq[0] * (x - c * t)) * np.sin(np.pi * freq[1] * (y - c * t))
                   * np.sin(np.pi * freq[2] * (z - c * t)))
        return self._from_host(np.broadcast_to(sol, self.nvars))


class advectiondiffusionNd_imex(GenericNDimFinDiff):
    """u_t = nu * Laplace(u) - c * sum_d du/dx_d, periodic: diffusion implicit (solved in Fourier space),
    advection explicit; both halves use the reference's stencil generator.  BASELINE config 3."""

    dtype_f = hip_imex_mesh
    ncomp = 2
    expl_kind = L.EXPL_STENCIL

    def __init__(self, nvars=512, nu=0.02, c=1.0, freq=2, stencil_type='center', order=2, lintol=1e-12,
                 liniter=10000, solver_type='direct', bc='periodic'):
        super().__init__(nvars, nu, 2, freq, 'center', order, lintol, liniter, solver_type, bc)
        self._stencil_expl = fd.periodic_operator_stencil(1, order, stencil_type, self.dx, -c)
        self._makeAttributeAndRegister('nu', 'c', localVars=locals(), readOnly=True)

    def configure_engine(self, engine):
        super().configure_engine(engine)
        engine.set_stencil(1, *self._stencil_expl)

    @classmethod
    def get_default_sweeper_class(cls):
        from pysdc_amd.sweepers import imex_1st_order

        return imex_1st_order

    def eval_f(self, u, t):
        f = self._out_f()
        self.engine.eval_f(u.ptr, 0.0, f.impl.ptr, f.expl.ptr)
        return f

    def u_exact(self, t):
        """sin modes advected with speed c and damped by the *discrete* symbols of both operators."""
        g = self.grids if self.ndim > 1 else (self.grids,)
        sol = 1.0
        for i in range(self.ndim):
            sol = sol * np.sin(np.pi * self.freq[i] * (g[i] - self.c * t))
        rho = sum((2.0 - 2.0 * np.cos(np.pi * f * self.dx)) / self.dx**2 for f in self.freq)
        return self._from_host(np.broadcast_to(sol * np.exp(-t * self.nu * rho), self.nvars))


class _SumCounter:
    """counter that reads as a device counter plus a host-side count"""

    def __init__(self, base, extra):
        self._base, self._extra = base, extra

    @property
    def niter(self):
        return self._base.niter + self._extra()

    def __call__(self, *args, **kwargs):
        pass

    def __str__(self):
        return f'{self.niter}'


class _DeviceCounter:
    """work counter whose value lives on the device (summed over trajectories); reading synchronises."""

    def __init__(self, prob, key):
        self._p, self._k = prob, key

    @property
    def niter(self):
        return self._p.engine.work_counters()[self._k]

    def __call__(self, *args, **kwargs):  # counting happens in the kernels
        pass

    def __str__(self):
        return f'{self.niter}'


class vanderpol_ensemble(Problem):
    """``ntraj`` independent van der Pol oscillators  x1' = x2,  x2' = mu (1 - x1^2) x2 - x1  as one level with
    state ``[x1[ntraj], x2[ntraj]]`` (shape (2, ntraj)).  Per trajectory it is the reference's ``vanderpol``
    (Van_der_Pol_implicit.py:10-201): same right-hand side, same Newton iteration with the closed-form 2x2
    inverse, same stopping rule ``max|g| < newton_tol`` / NaN / ``newton_maxiter`` and the same failure
    behaviour (``ProblemError``).  BASELINE config 4."""

    fused = True

    def __init__(self, ntraj=1, u0=None, mu=5.0, newton_maxiter=100, newton_tol=1e-9, stop_at_nan=True,
                 crash_at_maxiter=True, relative_tolerance=False, block_solver='closed_form'):
        if relative_tolerance:
            raise ProblemError('relative_tolerance is not available in the ensemble kernels')
        if not (stop_at_nan and crash_at_maxiter):
            raise ProblemError('the ensemble kernels always report Newton failures (stop_at_nan / crash_at_maxiter)')
        if block_solver not in ('closed_form', 'mfma'):
            raise ProblemError(f"block_solver must be 'closed_form' or 'mfma', got {block_solver!r}")
        ntraj = int(ntraj)
        if u0 is None:
            u0 = (2.0, 0.0)
        u0 = np.asarray(u0, dtype=float)
        if u0.shape == (2,):
            u0 = np.repeat(u0[:, None], ntraj, axis=1)
        if u0.shape != (2, ntraj):
            raise ProblemError(f'u0 must have shape (2,) or (2, {ntraj}), got {u0.shape}')
        super().__init__(init=((2, ntraj), None, np.dtype('float64')))
        nvars = (2 * ntraj,)
        self._makeAttributeAndRegister('nvars', 'ntraj', localVars=locals(), readOnly=True)
        self._makeAttributeAndRegister('mu', 'newton_maxiter', 'newton_tol', 'stop_at_nan', 'crash_at_maxiter',
                                       'relative_tolerance', 'block_solver', localVars=locals())
        self._u0 = u0
        self.work_counters['newton'] = _DeviceCounter(self, 'newton')
        self.work_counters['rhs'] = _DeviceCounter(self, 'rhs')
        # every Newton step is one Jacobian solve (Van_der_Pol_implicit.py:171), plus the direct calls
        self._jac_calls = 0
        self.work_counters['jacobian_solves'] = _SumCounter(self.work_counters['newton'], lambda: self._jac_calls)

    @property
    def u0(self):
        return self._u0

    @classmethod
    def get_default_sweeper_class(cls):
        from pysdc_amd.sweepers import generic_implicit

        return generic_implicit

    def configure_engine(self, engine):
        engine.set_problem_vdp(self.mu, self.newton_tol, self.newton_maxiter)
        engine.set_vdp_block_solver(self.block_solver)

    def u_exact(self, t, u_init=None, t_init=None):
        if t > 0.0:
            raise NotImplementedError('the ensemble has no reference solution for t > 0 (the reference integrates '
                                      'with SciPy, Van_der_Pol_implicit.py:96-103)')
        me = self.u_init
        me[:] = self._u0
        return me

    def eval_f(self, u, t):
        f = self._out_f()
        self.engine.eval_f(u.ptr, 0.0, f.ptr)
        return f

    def solve_system(self, rhs, dt, u0, t):
        sol = self._out_u()
        self.engine.solve(rhs.ptr, float(dt), sol.ptr, guess_ptr=u0.ptr)
        return sol

    def solve_jacobian(self, rhs, dt, u, **kwargs):
        """Van_der_Pol_implicit.py:190-201 for every trajectory of the ensemble"""
        out = self._out_u()
        e = self.engine
        L.check(e.lib.sdc_solve_jacobian(e.ctx, rhs.ptr, float(dt), u.ptr, out.ptr), e.ctx)
        self._jac_calls += self.ntraj
        return out


class _SpectralLaplacianIMEX(Problem):
    """shared part of the pseudo-spectral Allen-Cahn problems: implicit Laplacian with symbol -(2 pi k / L)^2
    applied / inverted through the engine's FFT pipeline, explicit pointwise reaction term."""

    dtype_f = hip_imex_mesh
    ncomp = 2
    fused = True  # the engine sweeps node by node on the device (nonlinear explicit part: sdc_sweep -> sweep_nodewise)
    rhs_autonomous = True   # eval_f(u, t) does not depend on t (a transfer may keep f(u[0]) of a start value it has seen)

    def _symbol(self):
        n = self.nvars[0]
        k = np.fft.fftfreq(n, 1.0 / n)
        return -((2 * np.pi / self.L * k) ** 2) + 0j

    @classmethod
    def get_default_sweeper_class(cls):
        from pysdc_amd.sweepers import imex_1st_order

        return imex_1st_order

    def eval_f(self, u, t):
        f = self._out_f()
        self.engine.eval_f(u.ptr, 0.0, f.impl.ptr, f.expl.ptr)
        if 'rhs' in self.work_counters:
            self.work_counters['rhs']()
        return f

    def eval_f_into(self, u, t, out):
        """eval_f with the result written into an existing imex field (a slab view): no temporary, no copy"""
        self.engine.eval_f(u.ptr, 0.0, out.impl.ptr, out.expl.ptr)
        if 'rhs' in self.work_counters:
            self.work_counters['rhs']()

    def eval_f_into_many(self, us, ts, outs):
        """eval_f_into for several fields at once: ONE transform round trip for all of them (the node loop after a
        prolongation, core/base_transfer.py:207-213); the right-hand side does not depend on t"""
        self.engine.eval_f_many([u.ptr for u in us], [o.impl.ptr for o in outs], [o.expl.ptr for o in outs])
        if 'rhs' in self.work_counters:
            for _ in us:
                self.work_counters['rhs']()

    def solve_system(self, rhs, factor, u0, t):
        me = self._out_u()
        self.engine.solve(rhs.ptr, float(factor), me.ptr)
        return me

    def _from_host(self, values):
        sol = self._out_u()
        sol[:] = values
        return sol


class allencahn2d_imex(_SpectralLaplacianIMEX):
    """pySDC/implementations/problem_classes/AllenCahn_2D_FFT.py:11-200: u_t = Laplace(u) + 1/eps^2 u (1 - u^nu) on
    [-L/2, L/2]^2, periodic, pseudo-spectral."""

    def __init__(self, nvars=None, nu=2, eps=0.04, radius=0.25, L=1.0, init_type='circle'):
        if nvars is None:
            nvars = (128, 128)
        if len(nvars) != 2:
            raise ProblemError('this is a 2d example, got %s' % (nvars,))
        if nvars[0] != nvars[1]:
            raise ProblemError('need a square domain, got %s' % (nvars,))
        if nvars[0] % 2 != 0:
            raise ProblemError('the setup requires nvars = 2^p per dimension')
        super().__init__(init=(tuple(nvars), None, np.dtype('float64')))
        nvars = tuple(nvars)
        self._makeAttributeAndRegister('nvars', 'nu', 'eps', 'radius', 'L', 'init_type', localVars=locals(),
                                       readOnly=True)
        self.dx = self.L / self.nvars[0]
        self.xvalues = np.array([i * self.dx - self.L / 2.0 for i in range(self.nvars[0])])

    def configure_engine(self, engine):
        engine.set_symbol(0, self._symbol())
        engine.set_reaction(1, 1.0 / self.eps**2 if self.eps > 0 else 0.0, 0.0, int(self.nu))

    def u_exact(self, t, u_init=None, t_init=None):
        if t != 0:
            raise NotImplementedError('reference solutions for t > 0 come from SciPy in the reference '
                                      '(AllenCahn_2D_FFT.py:192-198)')
        if self.init_type == 'circle':
            xv, yv = np.meshgrid(self.xvalues, self.xvalues, indexing='ij')
            me = np.tanh((self.radius - np.sqrt(xv**2 + yv**2)) / (np.sqrt(2) * self.eps))
        elif self.init_type == 'checkerboard':
            xv, yv = np.meshgrid(self.xvalues, self.xvalues)
            me = np.sin(2.0 * np.pi * xv) * np.sin(2.0 * np.pi * yv)
        else:
            raise NotImplementedError('type of initial value not implemented, got %s' % self.init_type)
        return self._from_host(me)


class allencahn_imex(_SpectralLaplacianIMEX):
    """2-D / 3-D Allen-Cahn of pySDC/implementations/problem_classes/AllenCahn_MPIFFT.py:11-160 on top of
    generic_MPIFFT_Laplacian.py:17-211 (real-space variant, ``spectral=False``):  u_t = Laplace(u)
    - 2/eps^2 u (1-u)(1-2u) - 6 dw u (1-u) on [0, L)^d.  The reference distributes the FFT with mpi4py-fft; here
    one GPU holds the field (time-parallel PFASST uses one GPU per time-slice, BASELINE config 5).  In 3-D the
    reference's 'circle' uses the first two coordinates only (a cylinder, AllenCahn_MPIFFT.py:133-139); the
    explicit 'sphere' option is this build's own."""

    def __init__(self, nvars=None, eps=0.04, radius=0.25, dw=0.0, init_type='circle', L=1.0, spectral=False):
        if nvars is None:
            nvars = (128, 128)
        if not (isinstance(nvars, tuple) and 1 < len(nvars) <= 3):
            raise ProblemError('Need two or three dimensions, got %s' % (nvars,))
        if len(set(nvars)) != 1 or nvars[0] % 2 != 0:
            raise ProblemError('need a square domain with an even number of points, got %s' % (nvars,))
        if spectral:
            raise ProblemError('the engine keeps the state in real space (spectral=False)')
        super().__init__(init=(nvars, None, np.dtype('float64')))
        self._makeAttributeAndRegister('nvars', 'eps', 'radius', 'dw', 'init_type', 'L', 'spectral',
                                       localVars=locals(), readOnly=True)
        self.ndim_ = len(nvars)
        self.dx = self.L / nvars[0]
        self.work_counters['rhs'] = WorkCounter()

    @property
    def ndim(self):
        return self.ndim_

    def configure_engine(self, engine):
        engine.set_symbol(0, self._symbol())
        engine.set_reaction(2, -2.0 / self.eps**2 if self.eps > 0 else 0.0, 6.0 * self.dw, 2)

    def u_exact(self, t, **kwargs):
        assert t == 0, 'ERROR: u_exact only valid for t=0'
        n = self.nvars[0]
        x = np.arange(n) * self.L / n
        X = np.meshgrid(*([x] * self.ndim_), indexing='ij', sparse=True)
        if self.init_type == 'circle':
            r2 = (X[0] - 0.5) ** 2 + (X[1] - 0.5) ** 2
        elif self.init_type == 'sphere':
            r2 = sum((Xi - 0.5) ** 2 for Xi in X)
        else:
            raise NotImplementedError(f'init_type {self.init_type!r}')
        me = 0.5 * (1.0 + np.tanh((self.radius - np.sqrt(r2)) / (np.sqrt(2) * self.eps)))
        return self._from_host(np.broadcast_to(me, self.nvars))


class advectiondiffusion1d_imex(Problem):
    """pySDC/implementations/problem_classes/AdvectionDiffusionEquation_1D_FFT.py:9-164:  u_t = -c u_x + nu u_xx on
    [-L/2, L/2), periodic, pseudo-spectral; diffusion implicit, advection explicit.  The reference works on the half spectrum
    of numpy's rfft / irfft; the engine transforms the promoted complex line, so both operators are handed over as symbols on
    all n modes (mode n - k the conjugate of mode k) - irfft's rule that the imaginary part a Nyquist entry produces is dropped
    is the engine's "real part" (include/sdcmi.h: sdc_set_symbol).  SURVEY 2.1: the IMEX parity anchor."""

    dtype_u = hip_mesh
    dtype_f = hip_imex_mesh
    ncomp = 2
    fused = True            # sdc_sweep runs the node loop on the device (sweep_nodewise: solve, A u and B u from one transform)
    rhs_autonomous = True

    def __init__(self, nvars=256, c=1.0, freq=-1, nu=0.02, L=1.0):
        super().__init__(init=(nvars, None, np.dtype('float64')))
        self._makeAttributeAndRegister('nvars', 'c', 'freq', 'nu', 'L', localVars=locals(), readOnly=True)
        if (self.nvars) % 2 != 0:
            raise ProblemError('setup requires nvars = 2^p')
        if not L_fft_ok(int(nvars)):
            raise ProblemError(f'the line transforms of the engine take n = 2^p (<= 2048), 3 * 2^p (24 .. 768) and 5 * 2^p (40 .. 640), '
                               f'got nvars = {nvars}')
        self.xvalues = np.array([i * self.L / self.nvars - self.L / 2.0 for i in range(self.nvars)])
        kx = np.zeros(self.init[0] // 2 + 1)
        for i in range(0, len(kx)):
            kx[i] = 2 * np.pi / self.L * i
        self.ddx = kx * 1j
        self.lap = -(kx**2)
        self.work_counters['rhs'] = WorkCounter()
        self.engine_nvars = (int(nvars),)

    def _full(self, half):
        """a symbol given on the modes 0 .. n/2 of rfft continued to all n modes of the complex transform"""
        n = int(self.nvars)
        full = np.empty(n, dtype=complex)
        full[: n // 2 + 1] = half
        full[n // 2 + 1:] = np.conj(half[1: n // 2][::-1])
        return full

    def configure_engine(self, engine):
        engine.set_symbol(0, self._full(self.nu * self.lap + 0j))
        engine.set_symbol(1, self._full(-self.c * self.ddx))

    @classmethod
    def get_default_sweeper_class(cls):
        from pysdc_amd.sweepers import imex_1st_order

        return imex_1st_order

    def eval_f(self, u, t):
        f = self._out_f()
        self.engine.eval_f(u.ptr, 0.0, f.impl.ptr, f.expl.ptr)
        self.work_counters['rhs']()
        return f

    def solve_system(self, rhs, factor, u0, t):
        me = self._out_u()
        self.engine.solve(rhs.ptr, float(factor), me.ptr)
        return me

    def _from_host(self, values):
        sol = self._out_u()
        sol[:] = values
        return sol

    def u_exact(self, t):
        """AdvectionDiffusionEquation_1D_FFT.py:130-164 (host arithmetic of the reference, result on the device)"""
        me = np.zeros(self.nvars)
        if self.freq > 0:
            omega = 2.0 * np.pi * self.freq
            me[:] = np.sin(omega * (self.xvalues - self.c * t)) * np.exp(-t * self.nu * omega**2)
        elif self.freq == 0:
            np.random.seed(1)
            me[:] = np.random.rand(self.nvars)
        else:
            t00 = 0.08
            if self.nu > 0:
                nbox = int(np.ceil(np.sqrt(4.0 * self.nu * (t00 + t) * 37.0 / (self.L**2))))
                for k in range(-nbox, nbox + 1):
                    for i in range(self.init[0]):
                        x = self.xvalues[i] - self.c * t + k * self.L
                        me[i] += np.sqrt(t00) / np.sqrt(t00 + t) * np.exp(-(x**2) / (4.0 * self.nu * (t00 + t)))
            else:
                raise ProblemError('There is no exact solution implemented for negative frequency and negative nu!')
        return self._from_host(me)


class advectiondiffusion1d_implicit(advectiondiffusion1d_imex):
    """AdvectionDiffusionEquation_1D_FFT.py:167-238: the same equation with both parts implicit - one complex symbol
    nu lap - c ddx that is applied (eval_f) and inverted (solve_system).  Like the reference's eval_f (:203, a statement
    without a call) the 'rhs' counter is never incremented."""

    dtype_f = hip_mesh
    ncomp = 1

    def configure_engine(self, engine):
        engine.set_symbol(0, self._full(self.nu * self.lap - self.c * self.ddx))

    @classmethod
    def get_default_sweeper_class(cls):
        from pysdc_amd.sweepers import generic_implicit

        return generic_implicit

    def eval_f(self, u, t):
        f = self._out_f()
        self.engine.eval_f(u.ptr, 0.0, f.ptr)
        return f


def L_fft_ok(n):
    """line lengths the engine's Stockham kernels take (csrc/fft.hpp: fft_length_ok), up to 2048"""
    if n < 2 or n > 2048 or n % 15 == 0:
        return False
    odd = 3 if n % 3 == 0 else (5 if n % 5 == 0 else 1)
    m = n // odd
    return (m & (m - 1)) == 0 and (odd != 3 or 24 <= n <= 768) and (odd != 5 or 40 <= n <= 640)
