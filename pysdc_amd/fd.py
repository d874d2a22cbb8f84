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


# ---- bounded grids with any mix of Dirichlet / Neumann ends (helpers/problem_helper.py:143-224) -------------------------------
_BC_DEFAULTS = ('val', 'neumann_bc_order', 'reduce')


def _side_params(bc_params, side, order):
    """parameters of one end: val (boundary value / boundary derivative), neumann_bc_order (accuracy of the one-sided first
    derivative that closes a Neumann end; default: the operator's order), reduce (centred stencils of growing order next to
    the boundary instead of shifted one-sided ones); unknown keys are refused like the reference does (:166)"""
    given = bc_params[side] if isinstance(bc_params, (list, tuple)) else (bc_params or {})
    extra = set(given) - set(_BC_DEFAULTS)
    if extra:
        raise AssertionError(f'unused BCs parameters : { {k: given[k] for k in sorted(extra)} }')
    return (given.get('val', 0.0), given.get('neumann_bc_order', order), bool(given.get('reduce', False)))


def bounded_operator_rows(derivative, order, kind, dx, size, bc, bc_params=None):
    """d^derivative/dx^derivative on `size` interior points of a bounded 1-D grid whose two ends carry a Dirichlet or a Neumann
    condition each - the matrix AND the boundary vector of the reference's `get_finite_difference_matrix(dim=1)`, as
    (rows, b): rows[i] = {column: weight} (already divided by dx**derivative), b[i] the constant the boundary data add to
    row i (f = A u + b).  bc: one string for both ends or a pair; an end is Neumann if its string contains 'neumann',
    Dirichlet if it contains 'dirichlet' (so 'dirichlet-zero', 'neumann-zero' are accepted like there).

    Row i within the half width of the interior stencil from an end gets a stencil that starts AT the boundary point: a
    one-sided one of order + derivative points shifted there (default), or - `reduce` - the centred stencil of order
    2 (i+1).  Its weight w_b on the boundary point multiplies data, not unknowns:
      Dirichlet, u(boundary) = val:            b[i] = w_b val
      Neumann,  u'(boundary) = val:            the boundary value is eliminated through the one-sided first-derivative stencil
                                               n of accuracy neumann_bc_order over the boundary point and its neighbours,
                                               u_b = (val dx - sum_{k>0} n_k u_k) / n_0:   row -= w_b / n_0 * n_{k>0},
                                               b[i] = w_b val dx / n_0.
    All weights are Fornberg's, exact rationals rounded once (module docstring)."""
    ends = bc if isinstance(bc, tuple) else (bc, bc)
    if len(ends) != 2 or not all(isinstance(e, str) for e in ends):
        raise AssertionError('Please pass BCs as string or tuple of strings')
    for e in ends:
        if 'neumann' not in e and 'dirichlet' not in e:
            raise AssertionError(f'unknown BC type : {e}')
    w_in, off_in = finite_difference_stencil(derivative, order, kind)
    reach = (-int(min(off_in)), int(max(off_in)))           # rows this close to the left / right end are rewritten
    rows = [{i + int(o): float(x) for o, x in zip(off_in, w_in) if 0 <= i + int(o) < size} for i in range(size)]
    b = np.zeros(size)
    for side, end in enumerate(ends):
        val, n_order, reduce = _side_params(bc_params, side, order)
        for i in range(reach[side]):
            row = i if side == 0 else size - 1 - i
            if reduce:
                w, offs = finite_difference_stencil(derivative, 2 * (i + 1), 'center')
            else:
                first = -(i + 1) if side == 0 else -(order + derivative) + (i + 2)
                w, offs = finite_difference_stencil(derivative, offsets=range(first, first + order + derivative))
            # the stencil as the reference places it: its first (left end) / last (right end) weight sits on the boundary
            # point, the others on the len(w) - 1 grid points next to that end, in order
            inner = list(w[1:]) if side == 0 else list(w[:-1])
            w_b = float(w[0] if side == 0 else w[-1])
            cols = range(len(inner)) if side == 0 else range(size - len(inner), size)
            if len(inner) > size:
                raise ValueError(f'{size} interior points are too few for a boundary stencil of {len(w)} points')
            entries = {int(c): float(x) for c, x in zip(cols, inner)}
            if 'dirichlet' in end:
                b[row] = val * w_b
            else:
                n, _ = finite_difference_stencil(1, n_order, 'forward' if side == 0 else 'backward')
                n_in = list(n[1:]) if side == 0 else list(n[:-1])
                n_b = float(n[0] if side == 0 else n[-1])
                if len(n_in) > size:
                    raise ValueError(f'{size} interior points are too few for a Neumann closure of order {n_order}')
                ncols = range(len(n_in)) if side == 0 else range(size - len(n_in), size)
                for c, x in zip(ncols, n_in):
                    entries[int(c)] = entries.get(int(c), 0.0) - w_b / n_b * float(x)
                b[row] = val * w_b / n_b * dx
            rows[row] = entries
    scale = dx**derivative
    return [{c: x / scale for c, x in r.items()} for r in rows], b / scale


def rows_to_table(rows, coeff=1.0):
    """row dictionaries -> the fixed-width table sdc_set_banded_operator takes: (columns[size][W] int32, -1 = unused;
    weights[size][W], multiplied by coeff - the factor generic_ND_FD.py:149 applies to the finished matrix)"""
    size = len(rows)
    width = max(len(r) for r in rows)
    cols = -np.ones((size, width), dtype=np.int32)
    wts = np.zeros((size, width))
    for i, r in enumerate(rows):
        for k, c in enumerate(sorted(r)):
            cols[i, k] = c
            wts[i, k] = r[c] * coeff
    return cols, wts


def rows_to_dense(rows):
    A = np.zeros((len(rows), len(rows)))
    for i, r in enumerate(rows):
        for c, x in r.items():
            A[i, c] = x
    return A


def boundary_vector_nd(b_1d, size, dim):
    """the vector get_finite_difference_matrix returns for `dim` dimensions: the 1-D boundary constants at the SAME flat
    indices of a zero vector of length size**dim (helpers/problem_helper.py:137,204,224: `b[iLine] = ...` with iLine a row of
    the 1-D matrix; its extension to the other faces is the reference's own TODO at :226)"""
    b = np.zeros(size**dim)
    for i in np.nonzero(b_1d)[0]:
        b[i if i < size - 1 - i else i - size] = b_1d[i]      # (rows of the left end count from the front, the others from the back)
    return b
