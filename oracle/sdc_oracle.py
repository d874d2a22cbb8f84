"""CPU ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy/SciPy restatement of the reference's SDC sweep path (pySDC v5.6, /root/reference).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (pysdc_amd/) never does and fails loudly without its HIP library.

Parity status: PINNED against the reference itself for everything below the coefficient
level - tests/golden/*.npz were produced by importing /root/reference in the build container
(tests/golden/gen_golden.py) and tests/test_oracle_golden.py replays them through this module.
Quadrature / Q-Delta coefficients come from the third-party ``qmat`` (absent here): the golden
files carry the coefficient matrices that were used, and this module takes coefficient matrices
as inputs, so it does not depend on how they were generated ("parity unpinned" applies to
pysdc_amd/coeffs.py only).

Every function cites the reference lines it follows (paths relative to /root/reference/pySDC).
State is kept as Python lists of ndarrays exactly like the reference's Level (core/level.py:96-106)
so that the arithmetic order - and hence rounding - is the reference's.
"""

import math

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import cg, gmres, spsolve
from scipy.special import factorial


# ----------------------------------------------------------------------------------------------
# finite differences (helpers/problem_helper.py)
# ----------------------------------------------------------------------------------------------
def fd_steps(derivative, order, stencil_type):
    """Stencil offsets; helpers/problem_helper.py:5-39."""
    if stencil_type == 'center':
        n = order + derivative - (derivative + 1) % 2 // 1
        return n, np.arange(n) - n // 2
    if stencil_type == 'forward':
        n = order + derivative
        return n, np.arange(n)
    if stencil_type == 'backward':
        n = order + derivative
        return n, -np.arange(n)
    if stencil_type == 'upwind':
        n = order + derivative
        if n <= 3:
            return fd_steps(derivative, order, 'backward')
        return n, np.append(-np.arange(n - 1)[::-1], [1])
    raise ValueError(f'unknown stencil type {stencil_type}')


def fd_stencil(derivative, order=None, stencil_type=None, steps=None):
    """Weights from the Taylor-matrix solve; helpers/problem_helper.py:42-80."""
    if steps is not None:
        n = len(steps)
    else:
        n, steps = fd_steps(derivative, order, stencil_type)
    steps = np.asarray(steps)
    T = np.zeros((n, n))
    idx = np.arange(n)
    inv_facs = 1.0 / factorial(idx)
    for i in range(n):
        T[i, :] = steps ** idx[i] * inv_facs[i]
    rhs = np.zeros(n)
    rhs[derivative] = 1.0
    coeff = np.linalg.solve(T, rhs)
    return coeff[np.argsort(steps)], np.sort(steps)


def fd_grid(size, bc):
    """dx and 1-D grid; helpers/problem_helper.py:245-269."""
    if bc == 'periodic':
        dx = 1.0 / size
        return dx, np.array([0.0 + dx * i for i in range(size)])
    if 'dirichlet' in bc or 'neumann' in bc:
        dx = 1.0 / (size + 1)
        return dx, np.array([0.0 + dx * (i + 1) for i in range(size)])
    raise NotImplementedError(bc)


def fd_matrix_and_vector(derivative, order, stencil_type, dx, size, dim, bc, bc_params=None):
    """Sparse FD operator and the boundary vector b; helpers/problem_helper.py:83-242 in full: periodic wrap-around,
    Dirichlet and Neumann rows on either side (bc a string or a pair), shifted one-sided stencils or reduced order next to
    the boundary (`reduce`), boundary values / derivatives `val`, `neumann_bc_order`.  b is set at the FLAT indices of the
    1-D boundary rows whatever `dim` is (:204,224 - the reference's own TODO at :226), and that is restated as it stands."""
    coeff, steps = fd_stencil(derivative, order, stencil_type)
    if type(bc) is not tuple:
        assert type(bc) == str, 'Please pass BCs as string or tuple of strings'
        bc = (bc, bc)
    bc_params = bc_params if bc_params is not None else {}
    if type(bc_params) is not list:
        bc_params = [bc_params, bc_params]
    b = np.zeros(size**dim)
    if bc[0] == 'periodic':
        assert bc[1] == 'periodic'
        A1 = 0 * sp.eye(size, format='csc')
        for i in steps:
            A1 += coeff[i] * sp.eye(size, k=steps[i])
            if steps[i] > 0:
                A1 += coeff[i] * sp.eye(size, k=-size + steps[i])
            if steps[i] < 0:
                A1 += coeff[i] * sp.eye(size, k=size + steps[i])
    else:
        A1 = sp.diags(coeff, steps, shape=(size, size), format='lil')
        defaults = {'val': 0.0, 'neumann_bc_order': order, 'reduce': False}
        for side in (0, 1):
            assert 'neumann' in bc[side] or 'dirichlet' in bc[side], f'unknown BC type : {bc[side]}'
            bc_params[side] = {**defaults, **bc_params[side]}
            par = bc_params[side].copy()
            val, reduce, n_order = par.pop('val'), par.pop('reduce'), par.pop('neumann_bc_order')
            assert len(par) == 0, f'unused BCs parameters : {par}'
            width = -min(steps) if side == 0 else max(steps)
            for i in range(width):
                line = i if side == 0 else -i - 1
                keep = slice(1, None) if side == 0 else slice(None, -1)
                edge = 0 if side == 0 else -1
                if reduce:
                    b_coeff, b_steps = fd_stencil(derivative, order=2 * (i + 1), stencil_type='center')
                else:
                    b_steps = (
                        np.arange(-(i + 1), order + derivative - (i + 1))
                        if side == 0
                        else np.arange(-(order + derivative) + (i + 2), (i + 2))
                    )
                    b_coeff, b_steps = fd_stencil(derivative, steps=b_steps)
                cols = slice(None, len(b_coeff) - 1) if side == 0 else slice(-len(b_coeff) + 1, None)
                A1[line, :] = 0
                A1[line, cols] = b_coeff[keep]
                if 'dirichlet' in bc[side]:
                    b[line] = val * b_coeff[edge]
                elif 'neumann' in bc[side]:
                    n_coeff, n_steps = fd_stencil(1, order=n_order, stencil_type='forward' if side == 0 else 'backward')
                    cols = slice(None, len(n_coeff) - 1) if side == 0 else slice(-len(n_coeff) + 1, None)
                    A1[line, cols] -= b_coeff[edge] / n_coeff[edge] * n_coeff[keep]
                    b[line] = val * b_coeff[edge] / n_coeff[edge] * dx
    A1 = A1.tocsc()
    if dim == 1:
        A = A1
    elif dim == 2:
        A = sp.kron(A1, sp.eye(size)) + sp.kron(sp.eye(size), A1)
    elif dim == 3:
        A = (
            sp.kron(A1, sp.eye(size**2))
            + sp.kron(sp.eye(size**2), A1)
            + sp.kron(sp.kron(sp.eye(size), A1), sp.eye(size))
        )
    else:
        raise NotImplementedError(dim)
    A /= dx**derivative
    b /= dx**derivative
    return A, b


def fd_matrix(derivative, order, stencil_type, dx, size, dim, bc):
    """the operator alone, the way generic_ND_FD.py:140-148 asks for it: no bc_params are passed on, b is dropped"""
    return fd_matrix_and_vector(derivative, order, stencil_type, dx, size, dim, bc)[0]


class _Counter:
    """work counter with the call protocol scipy callbacks need; core/problem.py:16-40."""

    def __init__(self):
        self.niter = 0

    def __call__(self, *args, **kwargs):
        self.niter += 1


# ----------------------------------------------------------------------------------------------
# problems
# ----------------------------------------------------------------------------------------------
class FDProblem:
    """du/dt = A u on a periodic grid or between Dirichlet / Neumann boundaries (bc a string or a pair of strings);
    implementations/problem_classes/generic_ND_FD.py:84-159 (setup), :188-206 (eval_f),
    :208-264 (solve_system)."""

    imex = False

    def __init__(
        self,
        nvars=512,
        coeff=1.0,
        derivative=1,
        freq=2,
        stencil_type='center',
        order=2,
        lintol=1e-12,
        liniter=10000,
        solver_type='direct',
        bc='periodic',
    ):
        if isinstance(nvars, int):
            nvars = (nvars,)
        nvars = tuple(nvars)
        ndim = len(nvars)
        if isinstance(freq, int):
            freq = (freq,) * ndim
        self.nvars, self.ndim, self.freq = nvars, ndim, tuple(freq)
        self.order, self.bc, self.stencil_type = order, bc, stencil_type
        self.lintol, self.liniter, self.solver_type = lintol, liniter, solver_type
        self.dx, self.xvalues = fd_grid(nvars[0], bc)
        self.A = fd_matrix(derivative, order, stencil_type, self.dx, nvars[0], ndim, bc)
        self.A *= coeff
        self.Id = sp.eye(int(np.prod(nvars)), format='csc')
        self.work_counters = {}
        if solver_type != 'direct':
            self.work_counters[solver_type] = _Counter()

    @property
    def grids(self):
        """generic_ND_FD.py:171-180."""
        x = self.xvalues
        if self.ndim == 1:
            return x
        if self.ndim == 2:
            return x[None, :], x[:, None]
        return x[None, :, None], x[:, None, None], x[None, None, :]

    def u_init(self):
        return np.zeros(self.nvars)

    def f_init(self):
        return np.zeros(self.nvars)

    def eval_f(self, u, t):
        f = self.f_init()
        f[:] = self.A.dot(u.flatten()).reshape(self.nvars)
        return f

    def solve_system(self, rhs, factor, u0, t):
        sol = self.u_init()
        if self.solver_type == 'direct':
            sol[:] = spsolve(self.Id - factor * self.A, rhs.flatten()).reshape(self.nvars)
        elif self.solver_type == 'CG':
            sol[:] = cg(
                self.Id - factor * self.A,
                rhs.flatten(),
                x0=u0.flatten(),
                rtol=self.lintol,
                maxiter=self.liniter,
                atol=0,
                callback=self.work_counters['CG'],
            )[0].reshape(self.nvars)
        elif self.solver_type == 'GMRES':
            sol[:] = gmres(
                self.Id - factor * self.A,
                rhs.flatten(),
                x0=u0.flatten(),
                rtol=self.lintol,
                maxiter=self.liniter,
                atol=0,
                callback=self.work_counters['GMRES'],
                callback_type='legacy',
            )[0].reshape(self.nvars)
        else:
            raise ValueError(self.solver_type)
        return sol


class HeatUnforced(FDProblem):
    """implementations/problem_classes/HeatEquation_ND_FD.py:64-132 (u_exact incl. the 3-D
    quirk of :119-123: the middle term of rho lacks '/ dx**2')."""

    def __init__(self, nvars=512, nu=0.1, freq=2, stencil_type='center', order=2, lintol=1e-12,
                 liniter=10000, solver_type='direct', bc='periodic', sigma=6e-2):
        super().__init__(nvars, nu, 2, freq, stencil_type, order, lintol, liniter, solver_type, bc)
        self.nu, self.sigma = nu, sigma

    def u_exact(self, t):
        ndim, freq, nu, dx = self.ndim, self.freq, self.nu, self.dx
        sol = self.u_init()
        if ndim == 1:
            x = self.grids
            rho = (2.0 - 2.0 * np.cos(np.pi * freq[0] * dx)) / dx**2
            if freq[0] > 0:
                sol[:] = np.sin(np.pi * freq[0] * x) * np.exp(-t * nu * rho)
            elif freq[0] == -1:
                sol[:] = np.exp(-0.5 * ((x - 0.5) / self.sigma) ** 2) * np.exp(-t * nu * rho)
        elif ndim == 2:
            rho = (2.0 - 2.0 * np.cos(np.pi * freq[0] * dx)) / dx**2 + (
                2.0 - 2.0 * np.cos(np.pi * freq[1] * dx)
            ) / dx**2
            x, y = self.grids
            sol[:] = np.sin(np.pi * freq[0] * x) * np.sin(np.pi * freq[1] * y) * np.exp(-t * nu * rho)
        else:
            rho = (
                (2.0 - 2.0 * np.cos(np.pi * freq[0] * dx)) / dx**2
                + (2.0 - 2.0 * np.cos(np.pi * freq[1] * dx))
                + (2.0 - 2.0 * np.cos(np.pi * freq[2] * dx)) / dx**2
            )
            x, y, z = self.grids
            sol[:] = (
                np.sin(np.pi * freq[0] * x)
                * np.sin(np.pi * freq[1] * y)
                * np.sin(np.pi * freq[2] * z)
                * np.exp(-t * nu * rho)
            )
        return sol


class HeatForced(HeatUnforced):
    """IMEX heat equation with analytic forcing; HeatEquation_ND_FD.py:135-230.
    f is an array of shape (2, *nvars): f[0] = impl, f[1] = expl (datatype_classes/mesh.py:166-173)."""

    imex = True

    def f_init(self):
        return np.zeros((2,) + self.nvars)

    def _profile(self):
        g = self.grids
        if self.ndim == 1:
            return np.sin(np.pi * self.freq[0] * g)
        p = np.sin(np.pi * self.freq[0] * g[0])
        for i in range(1, self.ndim):
            p = p * np.sin(np.pi * self.freq[i] * g[i])
        return p

    def eval_f(self, u, t):
        f = self.f_init()
        f[0][:] = self.A.dot(u.flatten()).reshape(self.nvars)
        f[1][:] = self._profile() * (
            self.nu * np.pi**2 * sum([fr**2 for fr in self.freq]) * np.cos(t) - np.sin(t)
        )
        return f

    def u_exact(self, t):
        sol = self.u_init()
        sol[:] = self._profile() * np.cos(t)
        return sol


class Advection(FDProblem):
    """implementations/problem_classes/AdvectionEquation_ND_FD.py:68-132."""

    def __init__(self, nvars=512, c=1.0, freq=2, stencil_type='center', order=2, lintol=1e-12,
                 liniter=10000, solver_type='direct', bc='periodic', sigma=6e-2):
        super().__init__(nvars, -c, 1, freq, stencil_type, order, lintol, liniter, solver_type, bc)
        self.c, self.sigma = c, sigma

    def u_exact(self, t):
        sol = self.u_init()
        c, freq = self.c, self.freq
        if self.ndim == 1:
            x = self.grids
            if freq[0] >= 0:
                sol[:] = np.sin(np.pi * freq[0] * (x - c * t))
            elif freq[0] == -1:
                sol[:] = np.exp(-0.5 * (((x - (c * t)) % 1.0 - 0.5) / self.sigma) ** 2)
        elif self.ndim == 2:
            x, y = self.grids
            sol[:] = np.sin(np.pi * freq[0] * (x - c * t)) * np.sin(np.pi * freq[1] * (y - c * t))
        else:
            x, y, z = self.grids
            sol[:] = (
                np.sin(np.pi * freq[0] * (x - c * t))
                * np.sin(np.pi * freq[1] * (y - c * t))
                * np.sin(np.pi * freq[2] * (z - c * t))
            )
        return sol


class AdvectionDiffusionIMEX:
    """Composite ND finite-difference advection-diffusion (BASELINE config 3):
    impl = nu * Laplacian (a HeatUnforced instance: eval_f / solve_system),
    expl = -c * gradient-sum (an Advection instance: eval_f).  The reference has no such class;
    both halves delegate to restatements of reference classes."""

    imex = True

    def __init__(self, nvars, nu=0.02, c=1.0, freq=2, order=2, stencil_type='center', lintol=1e-12,
                 liniter=10000, solver_type='direct', bc='periodic'):
        self.diff = HeatUnforced(nvars, nu, freq, 'center', order, lintol, liniter, solver_type, bc)
        self.adv = Advection(nvars, c, freq, stencil_type, order, lintol, liniter, 'direct', bc)
        self.nvars, self.ndim = self.diff.nvars, self.diff.ndim
        self.work_counters = self.diff.work_counters
        self.nu, self.c = nu, c

    def u_init(self):
        return np.zeros(self.nvars)

    def f_init(self):
        return np.zeros((2,) + self.nvars)

    def eval_f(self, u, t):
        f = self.f_init()
        f[0][:] = self.diff.eval_f(u, t)
        f[1][:] = self.adv.eval_f(u, t)
        return f

    def solve_system(self, rhs, factor, u0, t):
        return self.diff.solve_system(rhs, factor, u0, t)


class VanDerPol:
    """implementations/problem_classes/Van_der_Pol_implicit.py:106-129 (eval_f), :131-188
    (Newton solve_system), :190-201 (closed-form 2x2 Jacobian inverse)."""

    imex = False

    def __init__(self, u0=(2.0, 0.0), mu=5.0, newton_maxiter=100, newton_tol=1e-9,
                 stop_at_nan=True, crash_at_maxiter=True, relative_tolerance=False):
        self.u0 = np.asarray(u0, dtype=float)
        self.mu, self.newton_maxiter, self.newton_tol = mu, newton_maxiter, newton_tol
        self.stop_at_nan, self.crash_at_maxiter = stop_at_nan, crash_at_maxiter
        self.relative_tolerance = relative_tolerance
        self.nvars = (2,)
        self.work_counters = {'newton': _Counter(), 'jacobian_solves': _Counter(), 'rhs': _Counter()}

    def u_init(self):
        return np.zeros(2)

    def f_init(self):
        return np.zeros(2)

    def eval_f(self, u, t):
        x1, x2 = u[0], u[1]
        f = self.f_init()
        f[0] = x2
        f[1] = self.mu * (1 - x1**2) * x2 - x1
        self.work_counters['rhs']()
        return f

    def solve_system(self, rhs, dt, u0, t):
        mu = self.mu
        u = np.array(u0, dtype=float)
        x1, x2 = u[0], u[1]
        n = 0
        res = 99
        while n < self.newton_maxiter:
            g = np.array([x1 - dt * x2 - rhs[0], x2 - dt * (mu * (1 - x1**2) * x2 - x1) - rhs[1]])
            res = np.linalg.norm(g, np.inf) / (abs(float(np.max(np.abs(u)))) if self.relative_tolerance else 1.0)
            if res < self.newton_tol or np.isnan(res):
                break
            c = 1.0 / (-2 * dt**2 * mu * x1 * x2 - dt**2 - 1 + dt * mu * (1 - x1**2))
            dg = c * np.array([[dt * mu * (1 - x1**2) - 1, -dt], [2 * dt * mu * x1 * x2 + dt, -1]])
            self.work_counters['jacobian_solves']()
            u -= np.dot(dg, g)
            x1, x2 = u[0], u[1]
            n += 1
            self.work_counters['newton']()
        if np.isnan(res) and self.stop_at_nan:
            raise RuntimeError('Newton got nan after %i iterations, aborting...' % n)
        if n == self.newton_maxiter and self.crash_at_maxiter:
            raise RuntimeError('Newton did not converge after %i iterations, error is %s' % (n, res))
        return u

    def solve_jacobian(self, rhs, dt, u):
        """Van_der_Pol_implicit.py:190-201: (dg/du)^{-1} rhs with the closed-form 2x2 inverse"""
        mu, u1, u2 = self.mu, u[0], u[1]
        c = 1.0 / (-2 * dt**2 * mu * u1 * u2 - dt**2 - 1 + dt * mu * (1 - u1**2))
        dg = c * np.array([[dt * mu * (1 - u1**2) - 1, -dt], [2 * dt * mu * u1 * u2 + dt, -1]])
        self.work_counters['jacobian_solves']()
        return np.dot(dg, rhs)


# ----------------------------------------------------------------------------------------------
# level state and sweepers
# ----------------------------------------------------------------------------------------------
class Coll:
    """coefficient bundle: nodes[M], weights[M], Qmat[(M+1)^2], QI, QE ((M+1)^2, pySDC layout:
    zero first row; core/collocation.py:88-97, core/sweeper.py:100-123)."""

    def __init__(self, nodes, weights, Qmat, QI, QE=None, right_is_node=True, left_is_node=False):
        self.nodes = np.asarray(nodes, dtype=float)
        self.weights = np.asarray(weights, dtype=float)
        self.Qmat = np.asarray(Qmat, dtype=float)
        self.QI = np.asarray(QI, dtype=float)
        self.QE = None if QE is None else np.asarray(QE, dtype=float)
        self.num_nodes = len(self.nodes)
        self.right_is_node = right_is_node
        self.left_is_node = left_is_node


class Level:
    """containers of core/level.py:96-106 plus the status fields the path reads/writes
    (core/level.py:24-39) and the params (core/level.py:9-21)."""

    def __init__(self, prob, coll, dt, restol=-1.0, nsweeps=1, residual_type='full_abs'):
        M = coll.num_nodes
        self.prob, self.coll = prob, coll
        self.dt, self.restol, self.nsweeps, self.residual_type = dt, restol, nsweeps, residual_type
        self.reset()
        self.time = None

    def reset(self):
        M = self.coll.num_nodes
        self.u = [None] * (M + 1)
        self.f = [None] * (M + 1)
        self.uold = [None] * (M + 1)
        self.fold = [None] * (M + 1)
        self.tau = [None] * M
        self.residual = [None] * M
        self.uend = None
        self.status_residual = None
        self.unlocked = False
        self.updated = False
        self.sweep = None
        self.tag = None


def _fsum(L, j):
    """impl + expl for IMEX, plain f otherwise."""
    return L.f[j][0] + L.f[j][1] if L.prob.imex else L.f[j]


def predict(L, initial_guess='spread', rng=None):
    """core/sweeper.py:125-162."""
    P, c = L.prob, L.coll
    L.f[0] = P.eval_f(L.u[0], L.time)
    for m in range(1, c.num_nodes + 1):
        if initial_guess == 'spread':
            L.u[m] = np.array(L.u[0])
            L.f[m] = P.eval_f(L.u[m], L.time + L.dt * c.nodes[m - 1])
        elif initial_guess == 'copy':
            L.u[m] = np.array(L.u[0])
            L.f[m] = np.array(L.f[0])
        elif initial_guess == 'zero':
            L.u[m] = P.u_init()
            L.f[m] = P.f_init()
        elif initial_guess == 'random':
            L.u[m] = P.u_init() + rng.rand(1)[0]
            L.f[m] = P.f_init() + rng.rand(1)[0]
        else:
            raise ValueError(initial_guess)
    L.unlocked = True
    L.updated = True


def integrate(L):
    """Q.F; generic_implicit.py:29-49 / imex_1st_order.py:37-55 (left-to-right accumulation)."""
    c = L.coll
    me = []
    for m in range(1, c.num_nodes + 1):
        me.append(L.prob.u_init())
        for j in range(1, c.num_nodes + 1):
            me[-1] += L.dt * c.Qmat[m, j] * _fsum(L, j)
    return me


def sweep_generic_implicit(L):
    """generic_implicit.py:51-103."""
    P, c = L.prob, L.coll
    assert L.unlocked
    M = c.num_nodes
    integral = integrate(L)
    for m in range(M):
        for j in range(1, M + 1):
            integral[m] -= L.dt * c.QI[m + 1, j] * L.f[j]
        integral[m] += L.u[0]
        if L.tau[m] is not None:
            integral[m] += L.tau[m]
    for m in range(M):
        rhs = np.array(integral[m])
        for j in range(1, m + 1):
            rhs += L.dt * c.QI[m + 1, j] * L.f[j]
        alpha = L.dt * c.QI[m + 1, m + 1]
        if alpha == 0:
            L.u[m + 1] = rhs
        else:
            L.u[m + 1] = P.solve_system(rhs, alpha, L.u[m + 1], L.time + L.dt * c.nodes[m])
        L.f[m + 1] = P.eval_f(L.u[m + 1], L.time + L.dt * c.nodes[m])
    L.updated = True


def sweep_imex(L):
    """imex_1st_order.py:57-108 (no alpha == 0 shortcut)."""
    P, c = L.prob, L.coll
    assert L.unlocked
    M = c.num_nodes
    integral = integrate(L)
    for m in range(M):
        for j in range(1, M + 1):
            integral[m] -= L.dt * (c.QI[m + 1, j] * L.f[j][0] + c.QE[m + 1, j] * L.f[j][1])
        integral[m] += L.u[0]
        if L.tau[m] is not None:
            integral[m] += L.tau[m]
    for m in range(M):
        rhs = np.array(integral[m])
        for j in range(1, m + 1):
            rhs += L.dt * (c.QI[m + 1, j] * L.f[j][0] + c.QE[m + 1, j] * L.f[j][1])
        L.u[m + 1] = P.solve_system(rhs, L.dt * c.QI[m + 1, m + 1], L.u[m + 1], L.time + L.dt * c.nodes[m])
        L.f[m + 1] = P.eval_f(L.u[m + 1], L.time + L.dt * c.nodes[m])
    L.updated = True


def sweep(L):
    return sweep_imex(L) if L.prob.imex else sweep_generic_implicit(L)


def vabs(x):
    """mesh.__abs__: global max norm as Python float; datatype_classes/mesh.py:65-83."""
    return float(np.max(np.abs(x)))


def compute_residual(L, stage='', skip=()):
    """core/sweeper.py:164-215; skip = the sweeper parameter skip_residual_computation (:176-179: in the listed stages
    the residual keeps its value, 0.0 if there is none yet)."""
    if stage in skip:
        L.status_residual = 0.0 if L.status_residual is None else L.status_residual
        return None
    c = L.coll
    res_norm = []
    L.residual = integrate(L)
    for m in range(c.num_nodes):
        L.residual[m] += L.u[0] - L.u[m + 1]
        if L.tau[m] is not None:
            L.residual[m] += L.tau[m]
        res_norm.append(vabs(L.residual[m]))
    if L.residual_type == 'full_abs':
        L.status_residual = max(res_norm)
    elif L.residual_type == 'last_abs':
        L.status_residual = res_norm[-1]
    elif L.residual_type == 'full_rel':
        L.status_residual = max(res_norm) / vabs(L.u[0])
    elif L.residual_type == 'last_rel':
        L.status_residual = res_norm[-1] / vabs(L.u[0])
    else:
        raise ValueError(L.residual_type)
    L.updated = False
    return res_norm


def compute_end_point(L, do_coll_update=False):
    """generic_implicit.py:105-131 / imex_1st_order.py:110-137."""
    c = L.coll
    if c.right_is_node and not do_coll_update:
        L.uend = np.array(L.u[-1])
    else:
        L.uend = np.array(L.u[0])
        for m in range(c.num_nodes):
            L.uend += L.dt * c.weights[m] * _fsum(L, m + 1)
        if L.tau[-1] is not None:
            L.uend += L.tau[-1]


# ----------------------------------------------------------------------------------------------
# controller loop: single level SDC / multi-step SDC (controller_nonMPI.py)
# ----------------------------------------------------------------------------------------------
class _Step:
    def __init__(self, L, maxiter):
        self.L, self.maxiter = L, maxiter
        self.iter, self.done, self.prev_done, self.stage = 0, False, False, 'SPREAD'
        self.first = self.last = False
        self.prev = None
        self.slot = 0


def _converged(S):
    """check_convergence.py:72-82 (residual / maxiter rule)."""
    L = S.L
    iter_converged = S.iter >= S.maxiter
    res_converged = L.status_residual <= L.restol and (S.iter > 0 or L.sweep > 0)
    return iter_converged or res_converged


def run_sdc(make_level, u0, t0, Tend, num_procs=1, maxiter=50, mssdc_jac=True, do_coll_update=False,
            initial_guess='spread', on_sweep=None, skip_residual_computation=()):
    """Single-level SDC (num_procs=1) or multi-step SDC (num_procs>1) exactly as the serial
    controller stages them: controller_nonMPI.py:85-167 (time loop, 10*eps guard), :180-224
    (restart_block), :226-295 (send/recv = compute_end_point + copy + f[0] re-evaluation),
    :334-356 (spread), :479-543 (it_check), :545-582 (it_fine), :636-666 (it_coarse, single level).

    ``make_level()`` returns a fresh oracle Level.  Returns (uend, stats) with
    stats = dict(niter=[(time, niter)], residuals=[(time, [res after each iteration])])."""
    MS = [_Step(make_level(), maxiter) for _ in range(num_procs)]
    nsweeps = MS[0].L.nsweeps
    eps = np.finfo(float).eps
    slots = list(range(num_procs))
    time = [t0 + sum(MS[j].L.dt for j in range(p)) for p in slots]
    active = [time[p] < Tend - 10 * eps for p in slots]
    if not any(active):
        raise RuntimeError('Nothing to do, check t0, dt and Tend.')
    active_slots = [p for p in slots if active[p]]
    stats = {'niter': [], 'residuals': []}

    def restart_block(active_slots, time, u0):
        for j, p in enumerate(active_slots):
            S = MS[p]
            S.slot = p
            S.prev = MS[active_slots[j - 1]]
            S.L.reset()
            S.first = j == 0
            S.last = j == len(active_slots) - 1
            S.L.u[0] = np.array(u0)
            S.done = S.prev_done = False
            S.iter = 0
            S.stage = 'SPREAD'
            S.L.sweep = 1
            S.reshist = []
        for p in active_slots:
            MS[p].L.time = time[p]

    def send(S):
        if not S.last:
            compute_end_point(S.L, do_coll_update)
            S.L.tag = (0, S.iter, S.slot)

    def recv(S):
        if not S.prev_done and not S.first:
            src = S.prev.L
            if src.tag != (0, S.iter, S.prev.slot):
                raise RuntimeError('source and target tag are not the same, got %s and %s'
                                   % (src.tag, (0, S.iter, S.prev.slot)))
            S.L.u[0] = np.array(src.uend)
            S.L.f[0] = S.L.prob.eval_f(S.L.u[0], S.L.time)

    def one_sweep(S, stage):
        sweep(S.L)
        compute_residual(S.L, stage, skip_residual_computation)
        if on_sweep is not None:
            on_sweep(S)

    def pfasst(running_all):
        running = [S for S in running_all if S.stage != 'DONE']
        stage = running[0].stage
        assert all(S.stage == stage for S in running), 'not all stages are equal'
        if stage == 'SPREAD':
            for S in running:
                predict(S.L, initial_guess)
                S.stage = 'IT_CHECK'
        elif stage == 'IT_CHECK':
            for S in running:
                send(S)
                recv(S)
                compute_residual(S.L, 'IT_CHECK', skip_residual_computation)
            for S in running:
                if S.iter > 0:
                    S.reshist.append(S.L.status_residual)
                S.done = _converged(S)
            for S in running:
                if not S.first:
                    S.prev_done = S.prev.done
                    S.done = S.done and S.prev_done
                if not S.done:
                    S.iter += 1
                    if len(running) == 1 or mssdc_jac:
                        S.stage = 'IT_FINE'
                    else:
                        S.stage = 'IT_COARSE'
                else:
                    compute_end_point(S.L, do_coll_update)
                    stats['niter'].append((S.L.time, S.iter))
                    stats['residuals'].append((S.L.time, list(S.reshist)))
                    S.stage = 'DONE'
        elif stage == 'IT_FINE':
            for S in running:
                S.L.sweep = 0
            for k in range(nsweeps):
                for S in running:
                    S.L.sweep += 1
                for S in running:
                    send(S)
                    recv(S)
                for S in running:
                    one_sweep(S, 'IT_FINE')
            for S in running:
                S.stage = 'IT_CHECK'
        elif stage == 'IT_COARSE':
            for S in running:
                recv(S)
                one_sweep(S, 'IT_COARSE')
                send(S)
                S.stage = 'IT_CHECK'
        else:
            raise RuntimeError(stage)
        return all(S.done for S in running_all)

    restart_block(active_slots, time, u0)
    uend = None
    while any(active):
        MS_active = [MS[p] for p in active_slots]
        done = False
        while not done:
            done = pfasst(MS_active)
        uend = MS[active_slots[-1]].L.uend
        time[active_slots[0]] = time[active_slots[-1]] + MS[active_slots[-1]].L.dt
        for i in range(1, len(active_slots)):
            time[active_slots[i]] = time[active_slots[i] - 1] + MS[active_slots[i] - 1].L.dt
        active = [time[p] < Tend - 10 * eps for p in slots]
        active_slots = [p for p in slots if active[p]]
        restart_block(active_slots, time, uend)
    stats['niter'].sort(key=lambda x: x[0])
    stats['residuals'].sort(key=lambda x: x[0])
    return uend, stats


# ----------------------------------------------------------------------------------------------
# spectral (FFT-symbol) solve used to document the solver-equivalence budget
# ----------------------------------------------------------------------------------------------
def fd_symbol_1d(derivative, order, stencil_type, n, dx):
    """Discrete Fourier symbol lambda(k) = sum_s c_s exp(2 pi i k s / n) / dx^derivative of the
    periodic 1-D stencil that problem_helper.py:133-141 assembles."""
    coeff, steps = fd_stencil(derivative, order, stencil_type)
    k = np.arange(n)
    lam = np.zeros(n, dtype=complex)
    for c, s in zip(coeff, steps):
        lam += c * np.exp(2j * np.pi * k * s / n)
    return lam / dx**derivative


def spectral_solve(prob, rhs, factor):
    """(I - factor*A)^{-1} rhs through the FFT for a periodic FDProblem whose A = coeff * FD operator."""
    n = prob.nvars[0]
    derivative = 2 if isinstance(prob, HeatUnforced) else 1
    coeff = prob.nu if isinstance(prob, HeatUnforced) else -prob.c
    lam1 = coeff * fd_symbol_1d(derivative, prob.order, prob.stencil_type, n, prob.dx)
    lam = 0
    for ax in range(prob.ndim):
        shape = [1] * prob.ndim
        shape[ax] = n
        lam = lam + lam1.reshape(shape)
    return np.real(np.fft.ifftn(np.fft.fftn(rhs) / (1.0 - factor * lam)))


def fmt_float(x):
    return float(x) if not isinstance(x, float) else x


def isclose_rel(a, b, tol):
    """max |a-b| / max |b|."""
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(float(np.max(np.abs(b))), math.ulp(0.0))


# ----------------------------------------------------------------------------------------------
# space transfer between nested periodic grids (helpers/transfer_helper.py, transfer_classes/TransferMesh.py)
# ----------------------------------------------------------------------------------------------
def interpolation_matrix_1d_periodic(fine_grid, coarse_grid, k):
    """helpers/transfer_helper.py:153-186 (periodic, equidist_nested branch): even fine points copy their
    coarse twin, odd ones interpolate the k nearest coarse points with barycentric Lagrange polynomials."""
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
        # periodic continuation of the coarse nodes (transfer_helper.py:72-92)
        d = np.diff(nn)
        if np.all(d == 1):
            cont = coarse_grid[nn].astype(float)
        else:
            cont, shift = [coarse_grid[nn[0]]], 0.0
            for n_, d_ in zip(nn[1:], d):
                if d_ != 1:
                    shift = -1
                cont.append(coarse_grid[n_] + shift)
            cont = np.asarray(cont, dtype=float)
        if p > np.mean(fine_grid) and not (cont[0] <= p <= cont[-1]):
            cont = cont + 1
        one = np.asarray([1.0] + [0.0] * (k - 1))
        with np.errstate(divide='ignore'):
            M[i, nn] = np.asarray([BarycentricInterpolator(cont, np.roll(one, l))(p) for l in range(k)])
    return sp.csc_matrix(M)


def interpolation_matrix_1d_bounded(fine_grid, coarse_grid, k):
    """helpers/transfer_helper.py:206-231 (non-periodic, equidist_nested, pad=1): the coarse grid is mirrored by one
    point at each end (:250-273); odd fine points copy their coarse twin, even ones interpolate the k nearest
    padded coarse points with barycentric Lagrange polynomials; the padding columns are dropped (:243-244)."""
    from scipy.interpolate import BarycentricInterpolator

    nc = coarse_grid.size
    padded = np.concatenate([[2 * coarse_grid[0] - coarse_grid[1]], coarse_grid, [2 * coarse_grid[-1] - coarse_grid[-2]]])
    M = np.zeros((fine_grid.size, nc + 2))
    for i, p in enumerate(fine_grid):
        if i % 2 != 0:
            M[i, (i - 1) // 2 + 1] = 1.0
            continue
        nn = []
        for j in range(k):
            idx = i // 2 - k // 2 + 1 + j
            if idx < 0:
                idx += k
            elif idx > nc + 1:
                idx -= k
            nn.append(idx)
        nn = sorted(nn)
        unit = np.asarray([1.0] + [0.0] * (k - 1))
        with np.errstate(divide='ignore'):
            M[i, nn] = np.asarray([BarycentricInterpolator(padded[nn], np.roll(unit, l))(p) for l in range(k)])
    return sp.csc_matrix(M[:, 1:-1])


class MeshToMesh:
    """transfer_classes/TransferMesh.py:25-146 (periodic): P = kron of 1-D interpolation matrices of order
    iorder, R = kron of restr_factor * (interpolation matrix of order rorder)^T; :148-218 restrict / prolong
    (component-wise for imex arrays of shape (2, *nvars))."""

    def __init__(self, nvars_fine, nvars_coarse, iorder=2, rorder=2, periodic=True):
        nf = (nvars_fine,) if isinstance(nvars_fine, int) else tuple(nvars_fine)
        nc = (nvars_coarse,) if isinstance(nvars_coarse, int) else tuple(nvars_coarse)
        self.nf, self.nc = nf, nc
        P, R = [], []
        for a in range(len(nf)):
            if nf == nc:
                P.append(sp.eye(nf[a]))
                R.append(sp.eye(nc[a]))
                continue
            if periodic:
                fg = np.array([j * (1.0 / nf[a]) for j in range(nf[a])])
                cg = np.array([j * (1.0 / nc[a]) for j in range(nc[a])])
                build = interpolation_matrix_1d_periodic
            else:  # dirichlet grids: dx = 1/(n+1), points (j+1) dx (TransferMesh.py:62-64)
                fg = np.array([(j + 1) * (1.0 / (nf[a] + 1)) for j in range(nf[a])])
                cg = np.array([(j + 1) * (1.0 / (nc[a] + 1)) for j in range(nc[a])])
                build = interpolation_matrix_1d_bounded
            P.append(build(fg, cg, iorder))
            factor = 0.5 if rorder > 0 else 1.0
            R.append(factor * (P[-1] if iorder == rorder else build(fg, cg, rorder)).T)
        self.P, self.R = P[0], R[0]
        for a in range(1, len(nf)):
            self.P = sp.kron(self.P, P[a], format='csc')
            self.R = sp.kron(self.R, R[a], format='csc')

    def _apply(self, Mx, x, shape_in, shape_out):
        x = np.asarray(x)
        if x.shape == shape_in:
            return Mx.dot(x.flatten()).reshape(shape_out)
        return np.stack([Mx.dot(x[c].flatten()).reshape(shape_out) for c in range(x.shape[0])])

    def restrict(self, F):
        return self._apply(self.R, F, self.nf, self.nc)

    def prolong(self, G):
        return self._apply(self.P, G, self.nc, self.nf)


class MeshToMeshFFT:
    """transfer_classes/TransferMesh_FFT.py (1-D): restrict = injection F[::ratio] (:21-34); prolong = rfft of
    the coarse vector, modes [0, nc/2) and the LAST coarse entry (its Nyquist mode, stored at the fine Nyquist
    index) copied into a zero fine spectrum, irfft, times ratio (:36-57)."""

    def __init__(self, nvars_fine, nvars_coarse):
        self.nf, self.nc = int(np.prod(nvars_fine)), int(np.prod(nvars_coarse))
        self.ratio = self.nf // self.nc

    def _each(self, fn, x, n):
        x = np.asarray(x)
        return fn(x) if x.shape == (n,) else np.stack([fn(x[c]) for c in range(x.shape[0])])

    def restrict(self, F):
        return self._each(lambda v: v[:: self.ratio].copy(), F, self.nf)

    def prolong(self, G):
        def one(g):
            ch = np.fft.rfft(g)
            fh = np.zeros(self.nf // 2 + 1, dtype=np.complex128)
            h = self.nc // 2
            fh[0:h] = ch[0:h]
            fh[-1] = ch[-1]
            return np.fft.irfft(fh) * self.ratio

        return self._each(one, G, self.nc)


class MeshToMeshFFT2D:
    """transfer_classes/TransferMesh_FFT2D.py (square 2-D grids): restrict = injection F[::r, ::r] (:40-56);
    prolong = fft2, the four corner blocks of half-width nc/2 placed in the corners of a zero nf x nf spectrum,
    real part of ifft2, times 2 * ratio (:58-77; the constant is the reference's)."""

    def __init__(self, nvars_fine, nvars_coarse):
        self.nf, self.nc = int(nvars_fine[0]), int(nvars_coarse[0])
        self.ratio = self.nf // self.nc

    def _each(self, fn, x, n):
        x = np.asarray(x)
        return fn(x) if x.shape == (n, n) else np.stack([fn(x[c]) for c in range(x.shape[0])])

    def restrict(self, F):
        return self._each(lambda v: v[:: self.ratio, :: self.ratio].copy(), F, self.nf)

    def prolong(self, G):
        def one(g):
            tg = np.fft.fft2(g)
            tf = np.zeros((self.nf, self.nf), dtype=np.complex128)
            h, nf = self.nc // 2, self.nf
            tf[0:h, 0:h] = tg[0:h, 0:h]
            tf[nf - h:, 0:h] = tg[h:, 0:h]
            tf[0:h, nf - h:] = tg[0:h, h:]
            tf[nf - h:, nf - h:] = tg[h:, h:]
            return np.real(np.fft.ifft2(tf)) * self.ratio * 2

        return self._each(one, G, self.nc)


# ----------------------------------------------------------------------------------------------
# pseudo-spectral Allen-Cahn (problem_classes/AllenCahn_2D_FFT.py, AllenCahn_MPIFFT.py)
# ----------------------------------------------------------------------------------------------
class AllenCahn2D:
    """AllenCahn_2D_FFT.py:61-200: lap = -kx^2 - ky^2 on the rfft2 layout (:84-93), eval_f (:95-116),
    solve_system (:118-146), circle initial condition (:171-174)."""

    imex = True

    def __init__(self, nvars=(128, 128), nu=2, eps=0.04, radius=0.25, L=1.0, init_type='circle'):
        self.nvars, self.nu, self.eps, self.radius, self.L, self.init_type = tuple(nvars), nu, eps, radius, L, init_type
        self.ndim = 2
        n = self.nvars[0]
        self.dx = L / n
        self.xvalues = np.array([i * self.dx - L / 2.0 for i in range(n)])
        kx = np.zeros(n)
        ky = np.zeros(n // 2 + 1)
        kx[: int(n / 2) + 1] = 2 * np.pi / L * np.arange(0, int(n / 2) + 1)
        kx[int(n / 2) + 1:] = 2 * np.pi / L * np.arange(int(n / 2) + 1 - n, 0)
        ky[:] = 2 * np.pi / L * np.arange(0, n // 2 + 1)
        xv, yv = np.meshgrid(kx, ky, indexing='ij')
        self.lap = -(xv**2) - yv**2
        self.work_counters = {}

    def u_init(self):
        return np.zeros(self.nvars)

    def f_init(self):
        return np.zeros((2,) + self.nvars)

    def eval_f(self, u, t):
        f = self.f_init()
        f[0][:] = np.fft.irfft2(self.lap * np.fft.rfft2(u))
        if self.eps > 0:
            f[1][:] = 1.0 / self.eps**2 * u * (1.0 - u**self.nu)
        return f

    def solve_system(self, rhs, factor, u0, t):
        return np.fft.irfft2(np.fft.rfft2(rhs) / (1.0 - factor * self.lap))

    def u_exact(self, t):
        assert t == 0
        xv, yv = np.meshgrid(self.xvalues, self.xvalues, indexing='ij')
        return np.tanh((self.radius - np.sqrt(xv**2 + yv**2)) / (np.sqrt(2) * self.eps))


class AllenCahnND:
    """AllenCahn_MPIFFT.py:63-139 on generic_MPIFFT_Laplacian.py:113-124 (K2), :164-171 (Laplacian), :202-211
    (inversion), real-space variant; numpy's rfftn replaces the distributed FFT (not importable here, so this
    class is pinned per operation against AllenCahn2D's golden-pinned pieces and by closed-form checks only)."""

    imex = True

    def __init__(self, nvars=(64, 64, 64), eps=0.04, radius=0.25, dw=0.0, init_type='circle', L=1.0):
        self.nvars, self.eps, self.radius, self.dw, self.init_type, self.L = tuple(nvars), eps, radius, dw, init_type, L
        self.ndim = len(self.nvars)
        n = self.nvars[0]
        k = [np.fft.fftfreq(n, 1.0 / n)] * (self.ndim - 1) + [np.fft.rfftfreq(n, 1.0 / n)]
        Ks = np.meshgrid(*k, indexing='ij', sparse=True)
        self.K2 = sum((Ki * 2 * np.pi / L) ** 2 for Ki in Ks)
        self.work_counters = {'rhs': _Counter()}

    def u_init(self):
        return np.zeros(self.nvars)

    def f_init(self):
        return np.zeros((2,) + self.nvars)

    def eval_f(self, u, t):
        f = self.f_init()
        f[0][:] = np.fft.irfftn(-self.K2 * np.fft.rfftn(u), s=self.nvars, axes=tuple(range(self.ndim)))
        if self.eps > 0:
            f[1][:] = -2.0 / self.eps**2 * u * (1.0 - u) * (1.0 - 2.0 * u) - 6.0 * self.dw * u * (1.0 - u)
        self.work_counters['rhs']()
        return f

    def solve_system(self, rhs, factor, u0, t):
        return np.fft.irfftn(np.fft.rfftn(rhs) / (1.0 + factor * self.K2), s=self.nvars, axes=tuple(range(self.ndim)))

    def u_exact(self, t):
        assert t == 0
        n = self.nvars[0]
        x = np.arange(n) * self.L / n
        X = np.meshgrid(*([x] * self.ndim), indexing='ij', sparse=True)
        r2 = (X[0] - 0.5) ** 2 + (X[1] - 0.5) ** 2 if self.init_type == 'circle' else sum((Xi - 0.5) ** 2 for Xi in X)
        return np.broadcast_to(0.5 * (1.0 + np.tanh((self.radius - np.sqrt(r2)) / (np.sqrt(2) * self.eps))),
                               self.nvars).copy()


# ----------------------------------------------------------------------------------------------
# pseudo-spectral advection-diffusion in 1-D (problem_classes/AdvectionDiffusionEquation_1D_FFT.py)
# ----------------------------------------------------------------------------------------------
class AdvDiff1DIMEX:
    """AdvectionDiffusionEquation_1D_FFT.py:52-164: ddx = i kx, lap = -kx^2 on the rfft layout (:63-72), eval_f with the
    diffusion part implicit and the advection part explicit (:74-97), solve_system (:99-124), u_exact (:126-164)."""

    imex = True

    def __init__(self, nvars=256, c=1.0, freq=-1, nu=0.02, L=1.0):
        self.nvars, self.c, self.freq, self.nu, self.L = (int(nvars),), c, freq, nu, L
        self.ndim = 1
        n = int(nvars)
        self.xvalues = np.array([i * L / n - L / 2.0 for i in range(n)])
        kx = np.zeros(n // 2 + 1)
        for i in range(0, len(kx)):
            kx[i] = 2 * np.pi / L * i
        self.ddx = kx * 1j
        self.lap = -(kx**2)
        self.work_counters = {'rhs': _Counter()}

    def u_init(self):
        return np.zeros(self.nvars)

    def f_init(self):
        return np.zeros((2,) + self.nvars)

    def eval_f(self, u, t):
        f = self.f_init()
        tmp_u = np.fft.rfft(u)
        f[0][:] = np.fft.irfft(self.nu * self.lap * tmp_u)
        f[1][:] = np.fft.irfft(-self.c * self.ddx * tmp_u)
        self.work_counters['rhs']()
        return f

    def solve_system(self, rhs, factor, u0, t):
        return np.fft.irfft(np.fft.rfft(rhs) / (1.0 - self.nu * factor * self.lap))

    def u_exact(self, t):
        n = self.nvars[0]
        me = np.zeros(n)
        if self.freq > 0:
            omega = 2.0 * np.pi * self.freq
            me[:] = np.sin(omega * (self.xvalues - self.c * t)) * np.exp(-t * self.nu * omega**2)
        elif self.freq == 0:
            np.random.seed(1)
            me[:] = np.random.rand(n)
        else:
            t00 = 0.08
            nbox = int(np.ceil(np.sqrt(4.0 * self.nu * (t00 + t) * 37.0 / (self.L**2))))
            for k in range(-nbox, nbox + 1):
                for i in range(n):
                    x = self.xvalues[i] - self.c * t + k * self.L
                    me[i] += np.sqrt(t00) / np.sqrt(t00 + t) * np.exp(-(x**2) / (4.0 * self.nu * (t00 + t)))
        return me


class AdvDiff1DImplicit(AdvDiff1DIMEX):
    """AdvectionDiffusionEquation_1D_FFT.py:167-238: both parts implicit; eval_f never counts (:203 names the counter without
    calling it)."""

    imex = False

    def f_init(self):
        return np.zeros(self.nvars)

    def eval_f(self, u, t):
        tmp_u = np.fft.rfft(u)
        return np.fft.irfft(self.nu * self.lap * tmp_u - self.c * self.ddx * tmp_u)

    def solve_system(self, rhs, factor, u0, t):
        return np.fft.irfft(np.fft.rfft(rhs) / (1.0 - factor * (self.nu * self.lap - self.c * self.ddx)))
