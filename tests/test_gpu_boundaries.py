"""Bounded grids beyond dirichlet-zero on the device (helpers/problem_helper.py:143-224, generic_ND_FD.py:50-70):
eval_f / solve_system of Neumann and mixed ends against the dense matrices that tests/test_fd.py pins to the reference,
and the opt-in boundary data (use_bcParams: f = coeff (D u + b), the solve's right-hand side shifted by factor coeff b) -
the part of get_finite_difference_matrix's result the reference's own problem classes drop (generic_ND_FD.py:140-148).
The sweeps and runs of such levels against the reference's goldens are in tests/test_gpu_plugin.py (sweeps_neumann.npz,
runs_neumann.npz)."""
import numpy as np
import pytest

from pysdc_amd import fd

pytestmark = pytest.mark.gpu


def _dense_nd(A1, ndim):
    n = A1.shape[0]
    I = np.eye(n)
    if ndim == 1:
        return A1
    if ndim == 2:
        return np.kron(A1, I) + np.kron(I, A1)
    return np.kron(A1, np.eye(n * n)) + np.kron(np.eye(n * n), A1) + np.kron(np.kron(I, A1), I)


@pytest.mark.parametrize('nvars,bc,order,kind,derivative', [
    (33, 'neumann', 2, 'center', 2), (40, ('dirichlet', 'neumann'), 4, 'center', 2), ((12, 12), 'neumann-zero', 2, 'center', 2),
    ((9, 9, 9), ('neumann', 'dirichlet'), 4, 'center', 2), (31, 'dirichlet', 3, 'upwind', 1), (24, 'neumann', 6, 'center', 2)])
def test_operator_and_solve_equal_the_dense_matrix(nvars, bc, order, kind, derivative):
    from pysdc_amd.level import Step
    from pysdc_amd.problems import GenericNDimFinDiff
    from pysdc_amd.sweepers import generic_implicit

    coeff = 0.1 if derivative == 2 else -1.0
    S = Step(dict(problem_class=GenericNDimFinDiff,
                  problem_params=dict(nvars=nvars, coeff=coeff, derivative=derivative, freq=1, stencil_type=kind, order=order, bc=bc),
                  sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=2, quad_type='RADAU-RIGHT', QI='IE'),
                  level_params=dict(dt=1e-2), step_params=dict(maxiter=1)))
    P = S.levels[0].prob
    S.levels[0].engine   # (binds the problem to its engine)
    shape = P.nvars
    n = shape[0]
    rows, b = fd.bounded_operator_rows(derivative, order, kind, P.dx, n, bc)
    A = coeff * _dense_nd(fd.rows_to_dense(rows), len(shape))
    assert not np.any(b)
    rng = np.random.default_rng(3)
    uh = rng.standard_normal(shape)
    u = P._from_host(uh)
    f = P.eval_f(u, 0.0).get()
    want = (A @ uh.reshape(-1)).reshape(shape)
    assert np.max(np.abs(f - want)) <= 1e-12 * np.max(np.abs(want))
    factor = 0.02
    sol = P.solve_system(u, factor, u, 0.0).get()
    want = np.linalg.solve(np.eye(A.shape[0]) - factor * A, uh.reshape(-1)).reshape(shape)
    assert np.max(np.abs(sol - want)) <= 1e-10 * np.max(np.abs(want))


@pytest.mark.parametrize('bc,par', [('dirichlet', {'val': 1.5}), ('neumann', {'val': -2.0, 'neumann_bc_order': 2}),
                                    (('neumann', 'dirichlet'), ({'val': 0.5, 'reduce': True}, {'val': 3.0}))])   # (a list would mean levels)
def test_boundary_data_are_applied_when_asked_for(bc, par):
    """use_bcParams=True: boundary values / derivatives enter f and the solve; a steady state of u_t = nu u_xx with
    u(0) = a, u(1) = c is the straight line between them, and the implicit step leaves it where it is"""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import GenericNDimFinDiff
    from pysdc_amd.sweepers import generic_implicit

    n, order = 32, 4
    mk = lambda use: Step(dict(problem_class=GenericNDimFinDiff,   # noqa: E731
                               problem_params=dict(nvars=n, coeff=0.3, derivative=2, freq=1, order=order, bc=bc, bcParams=par,
                                                   use_bcParams=use),
                               sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=2, quad_type='RADAU-RIGHT', QI='IE'),
                               level_params=dict(dt=1e-2), step_params=dict(maxiter=1)))
    S = mk(True)
    P = S.levels[0].prob
    S.levels[0].engine
    rows, b = fd.bounded_operator_rows(2, order, 'center', P.dx, n, bc, par)
    A = 0.3 * fd.rows_to_dense(rows)
    assert np.any(b)
    uh = np.random.default_rng(5).standard_normal(n)
    u = P._from_host(uh)
    f = P.eval_f(u, 0.0).get()
    want = A @ uh + 0.3 * b
    assert np.max(np.abs(f - want)) <= 1e-12 * np.max(np.abs(want))
    sol = P.solve_system(u, 0.05, u, 0.0).get()
    want = np.linalg.solve(np.eye(n) - 0.05 * A, uh + 0.05 * 0.3 * b)
    assert np.max(np.abs(sol - want)) <= 1e-10 * np.max(np.abs(want))
    if bc == 'dirichlet':     # the line through the two boundary values is a steady state of the discrete operator
        line = 1.5 * np.ones(n)
        fl = P.eval_f(P._from_host(line), 0.0).get()
        assert np.max(np.abs(fl)) <= 1e-9 * np.max(np.abs(0.3 * b))
    # default: the reference's behaviour - bcParams changes nothing
    S0 = mk(False)
    P0 = S0.levels[0].prob
    S0.levels[0].engine
    rows0, _ = fd.bounded_operator_rows(2, order, 'center', P0.dx, n, bc)
    f0 = P0.eval_f(P0._from_host(uh), 0.0).get()
    want0 = 0.3 * fd.rows_to_dense(rows0) @ uh
    assert np.max(np.abs(f0 - want0)) <= 1e-12 * np.max(np.abs(want0))
    with pytest.raises(NotImplementedError):
        GenericNDimFinDiff(nvars=(8, 8), derivative=2, bc='neumann', bcParams={'val': 1.0}, use_bcParams=True)
