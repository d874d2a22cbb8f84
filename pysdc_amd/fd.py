"""Finite-difference stencils and grids on the host (NumPy).

Same algorithm as the reference's helpers (pySDC/helpers/problem_helper.py:5-80 offsets and Taylor-matrix
solve, :245-269 grid) so that the weights handed to the HIP kernels carry the same ~1e-15 rounding the
reference's matrices do."""
import numpy as np
from math import factorial


def get_steps(derivative, order, stencil_type):
    if stencil_type == 'center':
        n = order + derivative - (derivative + 1) % 2 // 1
        steps = np.arange(n) - n // 2
    elif stencil_type == 'forward':
        n = order + derivative
        steps = np.arange(n)
    elif stencil_type == 'backward':
        n = order + derivative
        steps = -np.arange(n)
    elif stencil_type == 'upwind':
        n = order + derivative
        if n <= 3:
            n, steps = get_steps(derivative, order, 'backward')
        else:
            steps = np.append(-np.arange(n - 1)[::-1], [1])
    else:
        raise ValueError(
            f'Stencil must be of type "center", "forward", "backward" or "upwind", not {stencil_type}.'
        )
    return n, steps


def get_finite_difference_stencil(derivative, order=None, stencil_type=None, steps=None):
    if steps is not None:
        n = len(steps)
        steps = np.asarray(steps)
    else:
        n, steps = get_steps(derivative, order, stencil_type)
    A = np.zeros((n, n))
    idx = np.arange(n)
    inv_facs = 1.0 / np.array([float(factorial(int(i))) for i in idx])
    for i in range(n):
        A[i, :] = steps ** idx[i] * inv_facs[i]
    sol = np.zeros(n)
    sol[derivative] = 1.0
    coeff = np.linalg.solve(A, sol)
    return coeff[np.argsort(steps)], np.sort(steps)


def get_1d_grid(size, bc, left_boundary=0.0, right_boundary=1.0):
    L = right_boundary - left_boundary
    if bc == 'periodic':
        dx = L / size
        xvalues = np.array([left_boundary + dx * i for i in range(size)])
    elif 'dirichlet' in bc or 'neumann' in bc:
        dx = L / (size + 1)
        xvalues = np.array([left_boundary + dx * (i + 1) for i in range(size)])
    else:
        raise NotImplementedError(f'Boundary conditions "{bc}" not implemented.')
    return dx, xvalues


def periodic_operator_stencil(derivative, order, stencil_type, dx, coeff):
    """(offsets, weights) of coeff * d^derivative/dx^derivative; weights scaled like
    problem_helper.py:239 (A /= dx**derivative) followed by generic_ND_FD.py:149 (A *= coeff)."""
    w, steps = get_finite_difference_stencil(derivative, order, stencil_type)
    w = w / dx**derivative
    w = w * coeff
    return [int(s) for s in steps], [float(x) for x in w]
