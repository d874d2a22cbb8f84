"""Edge sizes of the sweep path on the device against the oracle (oracle/sdc_oracle.py restating generic_implicit.py:51-103,
core/sweeper.py:125-215): the smallest grids the reference accepts (an even number of points per periodic axis: 2, 4, ...),
one node up to eight, the shortest and the longest lines of the transform kernels, and the default data flow (spectral
reuse, deferred node fields) as well as the one that stores every field.  Tolerance 1e-10 relative (BASELINE.json), residual
norms 1e-9."""
import numpy as np
import pytest

from oracle import sdc_oracle as O
from pysdc_amd import lib as L
from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS
from tests._cases import rel_err
from tests import _gpu as G

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _coll(M, qi, quad='RADAU-RIGHT'):
    c = CollBase(M, 0, 1, 'LEGENDRE', quad)
    QI = np.zeros_like(c.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS[qi](qGen=c.generator, tLeft=0).genCoeffs()
    return c, QI


@pytest.mark.parametrize('deferred', [True, False])
@pytest.mark.parametrize('nvars,M,qi', [((2,), 1, 'IE'), ((2,), 3, 'LU'), ((4,), 2, 'IE'), ((2, 2), 3, 'IE'), ((4, 4), 1, 'LU'),
                                        ((2, 2, 2), 2, 'IE'), ((4, 4, 4), 3, 'LU'), ((8, 8, 8), 8, 'IE'), ((16,), 8, 'LU'),
                                        ((4, 4), 7, 'IE'), ((2048,), 6, 'IE'), ((16, 16, 16), 1, 'IE')])
def test_small_grids_and_node_counts_vs_oracle(nvars, M, qi, deferred):
    rng = np.random.default_rng(7)
    n, nd = nvars[0], len(nvars)
    c, QI = _coll(M, qi)
    dt = 20.0 / (0.1 * 4 * nd * n * n)        # moderately stiff at every size
    P = O.HeatUnforced(nvars if nd > 1 else n, 0.1, 2, order=2)
    coll = O.Coll(c.nodes, c.weights, c.Qmat, QI)
    Lv = O.Level(P, coll, dt)
    u0 = rng.standard_normal(nvars)
    Lv.time = 0.0
    Lv.u[0] = np.array(u0)
    O.predict(Lv, 'spread')

    e = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1, order=2), M)
    e.set_coeffs(c.Qmat, QI, None, c.nodes, c.weights)
    e.set_deferred(deferred)
    e.upload(L.SLOT_U, 0, u0)
    e.predict(0.0, dt, 'spread')
    O.compute_residual(Lv)
    res, _ = e.residual(dt, 'full_abs')
    assert abs(res - Lv.status_residual) <= 1e-9 * abs(Lv.status_residual) + 1e-14
    for k in range(3):
        O.sweep(Lv)
        O.compute_residual(Lv)
        e.sweep(0.0, dt)
        res, norms = e.residual(dt, 'full_abs')
        scale = max(abs(Lv.status_residual), 1e-12 * float(np.max(np.abs(u0))))
        assert abs(res - Lv.status_residual) <= 1e-8 * scale + 1e-14, (k, res, Lv.status_residual)
        u = e.download_u()
        f = e.download_f()
        for m in range(M + 1):
            assert rel_err(u[m], Lv.u[m]) < TOL, (k, m)
            assert rel_err(f[m], Lv.f[m]) < 1e-9, (k, m)
        for dcu in (False, True):
            O.compute_end_point(Lv, dcu)
            e.end_point(dt, dcu)
            assert rel_err(e.download(L.SLOT_UEND), Lv.uend) < TOL, (k, dcu)
    e.close()


@pytest.mark.parametrize('quad', ['GAUSS', 'LOBATTO', 'RADAU-LEFT'])
def test_node_sets_whose_end_point_is_not_the_last_node(quad):
    """quadrature rules without / with the left end point (core/collocation.py:88-97): the end value needs the collocation
    update (core/sweeper.py: compute_end_point) or, for LOBATTO, is the last node again"""
    nvars, M = (32, 32), 3
    c, QI = _coll(M, 'IE', quad)
    dt = 5e-3
    P = O.HeatUnforced(nvars, 0.1, 2, order=2)
    coll = O.Coll(c.nodes, c.weights, c.Qmat, QI, right_is_node=quad == 'LOBATTO', left_is_node=quad in ('LOBATTO', 'RADAU-LEFT'))
    Lv = O.Level(P, coll, dt)
    u0 = np.random.default_rng(9).standard_normal(nvars)
    Lv.time = 0.0
    Lv.u[0] = np.array(u0)
    O.predict(Lv, 'spread')
    e = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1, order=2), M)
    e.set_coeffs(c.Qmat, QI, None, c.nodes, c.weights)
    e.upload(L.SLOT_U, 0, u0)
    e.predict(0.0, dt, 'spread')
    for k in range(2):
        O.sweep(Lv)
        e.sweep(0.0, dt)
        O.compute_residual(Lv)
        res, _ = e.residual(dt, 'full_abs')
        assert abs(res - Lv.status_residual) <= 1e-8 * abs(Lv.status_residual) + 1e-14
        O.compute_end_point(Lv, True)
        e.end_point(dt, True)
        assert rel_err(e.download(L.SLOT_UEND), Lv.uend) < TOL
    u = e.download_u()
    for m in range(M + 1):
        assert rel_err(u[m], Lv.u[m]) < TOL
    e.close()


@pytest.mark.parametrize('nvars,M', [((4,), 1), ((4,), 3), ((4, 4), 2), ((8, 8), 5), ((4, 4, 4), 3), ((8, 8, 8), 2)])
def test_imex_small_grids_vs_oracle(nvars, M):
    """imex_1st_order.py:57-137 on the smallest grids: implicit diffusion, explicit advection (complex symbol)"""
    nd, n = len(nvars), nvars[0]
    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    QI = np.zeros_like(c.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
    QE = np.zeros_like(c.Qmat)
    QE[1:, 1:] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs()
    dt = 0.5 / (n * n)
    P = O.AdvectionDiffusionIMEX(nvars if nd > 1 else n, 0.02, 1.0, 2, order=2)
    Lv = O.Level(P, O.Coll(c.nodes, c.weights, c.Qmat, QI, QE), dt)
    u0 = np.random.default_rng(13).standard_normal(nvars)
    Lv.time = 0.0
    Lv.u[0] = np.array(u0)
    O.predict(Lv, 'spread')
    e = G.engine_for('advdiff', dict(nvars=nvars, nu=0.02, c=1.0, order=2), M)
    e.set_coeffs(c.Qmat, QI, QE, c.nodes, c.weights)
    e.upload(L.SLOT_U, 0, u0)
    e.predict(0.0, dt, 'spread')
    for k in range(3):
        O.sweep(Lv)
        O.compute_residual(Lv)
        e.sweep(0.0, dt)
        res, _ = e.residual(dt, 'full_abs')
        assert abs(res - Lv.status_residual) <= 1e-8 * abs(Lv.status_residual) + 1e-14, (k, res, Lv.status_residual)
        u, f = e.download_u(), e.download_f()
        for m in range(M + 1):
            assert rel_err(u[m], Lv.u[m]) < TOL, (k, m)
            assert rel_err(f[m], Lv.f[m]) < 1e-9, (k, m)
        O.compute_end_point(Lv, False)
        e.end_point(dt, False)
        assert rel_err(e.download(L.SLOT_UEND), Lv.uend) < TOL
    e.close()


@pytest.mark.parametrize('ntraj', [1, 2, 63, 64, 65, 257])
def test_vdp_ensembles_of_ragged_size(ntraj):
    """ensembles that do not fill a wavefront / a workgroup, odd and even trajectory counts (the vectorised and the scalar
    branch of the right-hand-side launch): every trajectory against the oracle's scalar van der Pol sweeps
    (Van_der_Pol_implicit.py:76-201), Newton counts summed"""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import vanderpol_ensemble
    from pysdc_amd.sweepers import generic_implicit

    rng = np.random.default_rng(21)
    u0 = rng.uniform(-2, 2, size=(2, ntraj))
    M, dt = 3, 0.05
    c, QI = _coll(M, 'LU')
    desc = dict(problem_class=vanderpol_ensemble, problem_params=dict(ntraj=ntraj, u0=u0, mu=5.0, newton_tol=1e-9),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='LU'),
                level_params=dict(dt=dt), step_params=dict(maxiter=3))
    Lv = Step(desc).levels[0]
    Lv.status.time = 0.0
    Lv.u[0] = Lv.prob.u_exact(0.0)
    Lv.sweep.predict()
    refs = []
    for i in range(ntraj):
        P = O.VanDerPol(mu=5.0, newton_tol=1e-9, u0=tuple(u0[:, i]))
        Lo = O.Level(P, O.Coll(c.nodes, c.weights, c.Qmat, QI), dt)
        Lo.time = 0.0
        Lo.u[0] = np.array(u0[:, i])
        O.predict(Lo, 'spread')
        refs.append((P, Lo))
    for k in range(3):
        Lv.sweep.update_nodes()
        Lv.sweep.compute_residual()
        U = np.stack([np.asarray(Lv.u[m]) for m in range(M + 1)])      # (M+1, 2, ntraj)
        worst = 0.0
        for i, (P, Lo) in enumerate(refs):
            O.sweep(Lo)
            O.compute_residual(Lo)
            worst = max(worst, Lo.status_residual)
            assert rel_err(U[:, :, i], np.stack(Lo.u)) < TOL, (k, i)
        assert abs(Lv.status.residual - worst) <= 1e-9 * max(worst, 1e-6)
    assert Lv.prob.work_counters['newton'].niter == sum(P.work_counters['newton'].niter for P, _ in refs)


@pytest.mark.parametrize('n', [40, 48])
def test_3d_grids_with_an_odd_factor_fourier_solve_equals_the_krylov_solve(n):
    """40^3 (5 * 2^3) and 48^3 (3 * 2^4): the reference's direct solver needs minutes per solve there, so the exact Fourier
    solve of these 3-D grids - line transforms with a radix-5 / radix-3 stage along all three axes - is checked against the
    device's own conjugate gradients (the reference's other solver_type, generic_ND_FD.py:238-262; pinned by sweeps_cg.npz),
    which shares no transform code with it.  The 1-D / 2-D goldens of these line lengths come from the reference itself."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    ends, kernels = {}, {}
    for solver in ('direct', 'CG'):
        desc = dict(problem_class=heatNd_unforced,
                    problem_params=dict(nvars=(n, n, n), nu=0.1, freq=2, solver_type=solver, lintol=1e-13, liniter=2000),
                    sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='IE'),
                    level_params=dict(dt=2e-2, restol=1e-9), step_params=dict(maxiter=20))
        C = controller_nonMPI(1, dict(logger_level=40), desc)
        P = C.MS[0].levels[0].prob
        eng = C.MS[0].levels[0].engine
        eng.profile_enable(True)
        u0 = P.u_exact(0.0)
        u0[:] = u0.get() + 0.05 * np.random.default_rng(11).standard_normal((n, n, n))
        uend, stats = C.run(u0, 0.0, 4e-2)
        ends[solver] = uend.get()
        kernels[solver] = set(k.split('[')[0] for k in eng.profile_read())
    assert any(k.startswith('spec_') or k.startswith('fft_z') for k in kernels['direct']), kernels['direct']
    assert not any(k.startswith('cg') for k in kernels['direct']), kernels['direct']
    assert any(k.startswith('cg') for k in kernels['CG']), kernels['CG']
    assert rel_err(ends['direct'], ends['CG']) < 1e-9
