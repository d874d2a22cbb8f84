"""Finite-difference weights and grids on the host.

Weights come from Fornberg's recursion (B. Fornberg, "Generation of finite difference formulas on arbitrarily
spaced grids", Math. Comp. 51, 1988) carried out in exact rational arithmetic, so every weight is the correctly
rounded value of the exact rational (1, -2, 1; -1/12, 4/3, -5/2, ...).  The reference obtains the same
stencils from a floating-point Taylor-matrix solve (pySDC/helpers/problem_helper.py:42-80), whose results carry
~1e-15 relative rounding; the two agree far inside the 1e-10 parity budget (tests/test_fd.py pins the literal
values the reference's own test lists, tests/test_helpers/test_problem_helper.py:6-135).

The offset conventions (which grid points a 'center' / 'forward' / 'backward' / 'upwind' stencil of a given
order touches) are the reference's (problem_helper.py:5-39); the grid convention (periodic: dx = L/n, points
i*dx; bounded: dx = L/(n+1), interior points) is problem_helper.py:245-269."""
from fractions import Fraction

import numpy as np

_KINDS = ('center', 'forward', 'backward', 'upwind')


def stencil_offsets(derivative, order, kind):
    """grid offsets (ascending) of the stencil of accuracy `order` for the `derivative`-th derivative."""
    if kind not in _KINDS:
        raise ValueError(f'Stencil must be of type "center", "forward", "backward" or "upwind", not {kind}.')
    width = derivative + order
    if kind == 'center':
        # symmetric stencils gain one order for even derivatives: one point fewer
        width -= 1 - derivative % 2
        first = -(width // 2)
    elif kind == 'forward':
        first = 0
    elif kind == 'backward' or width <= 3:  # short upwind stencils are one-sided
        first = 1 - width
    else:  # upwind: one point downstream, the rest upstream
        first = 2 - width
    return list(range(first, first + width))


def fornberg_weights(derivative, offsets, x0=0):
    """exact weights (Fractions) w_j with sum_j w_j f(x0 + offsets[j] h) = h^derivative f^(derivative)(x0) + O(h^p).

    Fornberg's recursion: c[k][j] is the weight of node j for the k-th derivative using the nodes seen so far;
    every new node updates the older columns in place."""
    nodes = [Fraction(o) for o in offsets]
    z = Fraction(x0)
    c = [[Fraction(0)] * len(nodes) for _ in range(derivative + 1)]
    c[0][0] = Fraction(1)
    scale_prev = Fraction(1)
    for i in range(1, len(nodes)):
        top = min(i, derivative)
        scale = Fraction(1)
        gap_new = nodes[i] - z
        gap_last = nodes[i - 1] - z
        for j in range(i):
            sep = nodes[i] - nodes[j]
            scale *= sep
            if j == i - 1:
                for k in range(top, 0, -1):
                    c[k][i] = scale_prev * (k * c[k - 1][j] - gap_last * c[k][j]) / scale
                c[0][i] = -scale_prev * gap_last * c[0][j] / scale
            for k in range(top, 0, -1):
                c[k][j] = (gap_new * c[k][j] - k * c[k - 1][j]) / sep
            c[0][j] = gap_new * c[0][j] / sep
        scale_prev = scale
    return c[derivative]


def finite_difference_stencil(derivative, order=None, kind=None, offsets=None):
    """(weights as float64 array, offsets as int array), offsets ascending; `offsets` overrides `kind`."""
    if offsets is None:
        offsets = stencil_offsets(derivative, order, kind)
    offsets = sorted(int(o) for o in offsets)
    if len(offsets) <= derivative:
        raise ValueError(f'{len(offsets)} points cannot carry a derivative of order {derivative}')
    w = fornberg_weights(derivative, offsets)
    return np.array([float(x) for x in w]), np.array(offsets)


def grid_1d(size, bc, left_boundary=0.0, right_boundary=1.0):
    """(dx, points) of the reference's 1-D grids."""
    length = right_boundary - left_boundary
    if bc == 'periodic':
        cells, first = size, 0
    elif 'dirichlet' in bc or 'neumann' in bc:
        cells, first = size + 1, 1
    else:
        raise NotImplementedError(f'Boundary conditions "{bc}" not implemented.')
    dx = length / cells
    return dx, np.array([left_boundary + dx * (i + first) for i in range(size)])


def periodic_operator_stencil(derivative, order, kind, dx, coeff):
    """(offsets, weights) of coeff * d^derivative/dx^derivative on a periodic grid: weights divided by
    dx**derivative, then multiplied by coeff - the two scalings the reference applies to its matrix in that
    order (problem_helper.py:239, generic_ND_FD.py:149)."""
    w, offsets = finite_difference_stencil(derivative, order, kind)
    w = w / dx**derivative
    w = w * coeff
    return [int(s) for s in offsets], [float(x) for x in w]


def dirichlet_operator_rows(derivative, order, kind, dx, coeff, size):
    """coeff * d^derivative/dx^derivative on `size` interior points between two boundary points that hold zero, as a table of
    rows: (columns[size][W] int32, -1 = unused; weights[size][W]).  Interior rows carry the stencil of the given kind; in
    the rows whose stencil would reach beyond the boundary point the reference SHIFTS a one-sided stencil of width
    order + derivative so that it starts at the boundary point (helpers/problem_helper.py:143-224, `reduce = False`): row i
    (i < half width) uses the grid offsets -(i+1) .. order + derivative - (i+2), whose first weight multiplies the boundary
    value (zero here) and is dropped; mirrored at the other end.  Weights: Fornberg's recursion in exact rationals, then the
    reference's two scalings (/ dx**derivative, * coeff)."""
    w_in, off_in = finite_difference_stencil(derivative, order, kind)
    half_left, half_right = -int(min(off_in)), int(max(off_in))
    shifted = order + derivative
    if size < max(shifted - 1, len(off_in)):
        raise ValueError(f'{size} interior points are too few for boundary stencils of width {shifted}')
    rows = []
    for i in range(size):
        if i < half_left:                      # next to the left boundary: offsets -(i+1) .. , first one is the boundary
            offs = list(range(-(i + 1), shifted - (i + 1)))
            w, offs = finite_difference_stencil(derivative, offsets=offs)
            entries = [(i + int(o), float(x)) for o, x in zip(offs[1:], w[1:])]
        elif size - 1 - i < half_right:        # next to the right boundary: mirrored
            k = size - 1 - i
            offs = list(range(-(shifted - (k + 2)), k + 2))
            w, offs = finite_difference_stencil(derivative, offsets=offs)
            entries = [(i + int(o), float(x)) for o, x in zip(offs[:-1], w[:-1])]
        else:
            entries = [(i + int(o), float(x)) for o, x in zip(off_in, w_in) if 0 <= i + int(o) < size]
        rows.append(entries)
    width = max(len(r) for r in rows)
    cols = -np.ones((size, width), dtype=np.int32)
    wts = np.zeros((size, width))
    for i, entries in enumerate(rows):
        for k, (j, x) in enumerate(entries):
            cols[i, k] = j
            wts[i, k] = x / dx**derivative * coeff
    return cols, wts
