"""Product finite-difference weights (pysdc_amd/fd.py, Fornberg recursion in exact arithmetic) against the literal
stencils the reference pins in tests/test_helpers/test_problem_helper.py:6-135 and against the oracle's restatement
of the reference's Taylor-matrix solve (helpers/problem_helper.py:42-80)."""
import numpy as np
import pytest

from pysdc_amd import fd
from oracle import sdc_oracle as O

LITERAL = [
    # derivative, order, kind, offsets, numerators, denominator
    (1, 2, 'center', [-1, 0, 1], [-1, 0, 1], 2),
    (1, 4, 'center', [-2, -1, 0, 1, 2], [1, -8, 0, 8, -1], 12),
    (1, 6, 'center', [-3, -2, -1, 0, 1, 2, 3], [-1, 9, -45, 0, 45, -9, 1], 60),
    (1, 1, 'upwind', [-1, 0], [-1, 1], 1),
    (1, 2, 'upwind', [-2, -1, 0], [1, -4, 3], 2),
    (1, 3, 'upwind', [-2, -1, 0, 1], [1, -6, 3, 2], 6),
    (1, 4, 'upwind', [-3, -2, -1, 0, 1], [-5, 30, -90, 50, 15], 60),
    (1, 5, 'upwind', [-4, -3, -2, -1, 0, 1], [3, -20, 60, -120, 65, 12], 60),
    (2, 2, 'center', [-1, 0, 1], [1, -2, 1], 1),
    (2, 4, 'center', [-2, -1, 0, 1, 2], [-1, 16, -30, 16, -1], 12),
    (2, 6, 'center', [-3, -2, -1, 0, 1, 2, 3], [2, -27, 270, -490, 270, -27, 2], 180),
    (2, 8, 'center', [-4, -3, -2, -1, 0, 1, 2, 3, 4], [-9, 128, -1008, 8064, -14350, 8064, -1008, 128, -9], 5040),
    (1, 3, 'forward', [0, 1, 2, 3], [-11, 18, -9, 2], 6),
    (2, 2, 'backward', [-3, -2, -1, 0], [-1, 4, -5, 2], 1),
]


@pytest.mark.parametrize('derivative,order,kind,offsets,num,den', LITERAL)
def test_literal_stencils(derivative, order, kind, offsets, num, den):
    w, s = fd.finite_difference_stencil(derivative, order, kind)
    assert list(s) == offsets
    # exact rationals, correctly rounded: equality, not closeness
    assert list(w) == [n / den for n in num]


def test_given_offsets_override_kind():
    w, s = fd.finite_difference_stencil(2, offsets=[0, -1, -3, -2])
    assert list(s) == [-3, -2, -1, 0] and list(w) == [-1.0, 4.0, -5.0, 2.0]
    with pytest.raises(ValueError):
        fd.finite_difference_stencil(2, offsets=[0, 1])
    with pytest.raises(ValueError):
        fd.stencil_offsets(1, 2, 'sideways')


@pytest.mark.parametrize('derivative', [1, 2])
@pytest.mark.parametrize('order', [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize('kind', ['center', 'forward', 'backward', 'upwind'])
def test_agrees_with_reference_algorithm(derivative, order, kind):
    """the oracle restates the reference's Taylor-matrix solve: same offsets; weights to the rounding of THAT solve
    (1e-15 for the centred stencils the problems use, up to 1e-11 for ten-point one-sided ones, whose Taylor
    matrix is ill-conditioned - the product's weights are exact)."""
    if kind == 'center' and derivative == 2 and order == 1:
        pytest.skip('two points cannot carry a second derivative (singular in the reference as well)')
    w, s = fd.finite_difference_stencil(derivative, order, kind)
    wo, so = O.fd_stencil(derivative, order, kind)
    assert list(s) == [int(x) for x in so]
    np.testing.assert_allclose(w, wo, rtol=0, atol=(2e-14 if kind == 'center' else 1e-10) * np.max(np.abs(wo)))


def test_exactness_on_polynomials():
    for derivative, order, kind in [(1, 4, 'upwind'), (2, 6, 'center'), (1, 3, 'forward')]:
        w, s = fd.finite_difference_stencil(derivative, order, kind)
        for p in range(len(s)):
            exact = float(np.prod(np.arange(p, p - derivative, -1))) if p == derivative else 0.0
            assert abs(sum(wi * float(si) ** p for wi, si in zip(w, s)) - exact) < 1e-9


def test_grids():
    dx, x = fd.grid_1d(8, 'periodic')
    assert dx == 1 / 8 and np.array_equal(x, np.arange(8) / 8)
    dx, x = fd.grid_1d(7, 'dirichlet-zero')
    assert dx == 1 / 8 and np.allclose(x, np.arange(1, 8) / 8)
    dx, x = fd.grid_1d(3, 'neumann-zero', -1.0, 1.0)
    assert dx == 0.5 and np.allclose(x, [-0.5, 0.0, 0.5])
    with pytest.raises(NotImplementedError):
        fd.grid_1d(4, 'robin')
