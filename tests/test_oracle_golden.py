"""CPU: the oracle (oracle/sdc_oracle.py) replayed against golden vectors produced by the reference
itself (tests/golden/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import sdc_oracle as O
from tests._cases import load_cases, make_oracle_problem, make_oracle_coll, rel_err

# the oracle follows the reference's arithmetic order; direct sparse solves are deterministic, so the
# agreement is at rounding level (Newton / CG paths included)
TOL = 1e-13

SWEEP_FILES = ['sweeps_heat.npz', 'sweeps_imex.npz', 'sweeps_adv.npz', 'sweeps_vdp.npz', 'sweeps_ac.npz', 'sweeps_cg.npz', 'sweeps_gmres.npz', 'sweeps_guess.npz',
               'sweeps_dirichlet_nd.npz', 'sweeps_dirichlet_ho.npz', 'sweeps_neumann.npz', 'sweeps_ad1d.npz']
SWEEP_CASES = ([(f, n) for f in SWEEP_FILES for n in load_cases(f)]
               + [('sweeps_pin1024.npz', 'pin_heat1d_1024_M5_IE')]   # (its 2-D companion is stored as subsamples: GPU tests)
               # grids of 3 * 2^p points (the 24^3 cases take the sparse LU minutes: they are the GPU suite's)
               + [('sweeps_radix3.npz', n) for n in load_cases('sweeps_radix3.npz') if '3d_24' not in n]
               + [('sweeps_radix5.npz', n) for n in load_cases('sweeps_radix5.npz')])


@pytest.mark.parametrize('fname,name', SWEEP_CASES)
def test_sweep_case(fname, name):
    case = load_cases(fname)[name]
    meta = case['meta']
    prob = make_oracle_problem(meta['prob'], meta['prob_params'])
    coll = make_oracle_coll(case)
    L = O.Level(prob, coll, meta['dt'])
    L.time = meta['t0']
    L.u[0] = np.array(case['u0'], dtype=float).reshape(prob.nvars)
    if meta['has_tau']:
        for m in range(coll.num_nodes):
            L.tau[m] = np.array(case['tau'][m])
    ig = meta['sweeper_params'].get('initial_guess', 'spread')
    O.predict(L, ig, rng=np.random.RandomState(meta['sweeper_params'].get('random_seed', 1984)) if ig == 'random' else None)

    def check(tag):
        got_u = np.stack(L.u)
        got_f = np.stack(L.f)
        assert rel_err(got_u, case[f'{tag}_u']) < TOL, tag
        assert rel_err(got_f, case[f'{tag}_f']) < TOL * 10, tag
        for rt in ('full_abs', 'last_abs', 'full_rel', 'last_rel'):
            L.residual_type = rt
            O.compute_residual(L)
            ref = float(case[f'{tag}_res_{rt}'])
            assert abs(L.status_residual - ref) <= 1e-11 * max(abs(ref), 1e-3) + 1e-15, (tag, rt)
        for dcu in (False, True):
            O.compute_end_point(L, dcu or not coll.right_is_node)
            assert rel_err(L.uend, case[f'{tag}_uend_{int(dcu)}']) < TOL, (tag, dcu)

    check('k0')
    for k in range(1, meta['nsweeps'] + 1):
        if f'k{k}_QI' in case:
            L.coll = make_oracle_coll(case, QI=case[f'k{k}_QI'])
        O.sweep(L)
        check(f'k{k}')
    for key, counter in prob.work_counters.items():
        assert counter.niter == int(case[f'work_{key}'][-1]), key


RUN_CASES = ([('runs.npz', n) for n in load_cases('runs.npz')] + [('runs_dirichlet.npz', n) for n in load_cases('runs_dirichlet.npz')]
             + [('runs_dirichlet_nd.npz', n) for n in load_cases('runs_dirichlet_nd.npz')] + [('runs_dirichlet_ho.npz', n) for n in load_cases('runs_dirichlet_ho.npz')]
             + [('runs_neumann.npz', n) for n in load_cases('runs_neumann.npz')]
             + [('runs_skip.npz', n) for n in load_cases('runs_skip.npz')]
             + [('runs_nsweeps2.npz', n) for n in load_cases('runs_nsweeps2.npz')] + [('sweeps_pin1024.npz', 'pin_heat1d_1024_run')]
             + [('runs_radix3.npz', n) for n in load_cases('runs_radix3.npz') if '3d_24' not in n]
             + [('runs_radix5.npz', n) for n in load_cases('runs_radix5.npz')]
             + [('runs_ad1d.npz', n) for n in load_cases('runs_ad1d.npz')]   # AdvectionDiffusionEquation_1D_FFT.py, both classes
             + [('runs_relay8.npz', n) for n in load_cases('runs_relay8.npz') if 'alltodone' not in n])  # (run_sdc has no all_to_done)   # skip_residual_computation (core/sweeper.py:176-179)


@pytest.mark.parametrize('fname,name', RUN_CASES)
def test_run_case(fname, name):
    case = load_cases(fname)[name]
    meta = case['meta']
    lp = meta['level_params']
    coll = make_oracle_coll(case)

    def make_level():
        prob = make_oracle_problem(meta['prob'], meta['prob_params'])
        return O.Level(prob, coll, lp['dt'], restol=lp.get('restol', -1.0), nsweeps=lp.get('nsweeps', 1))

    shape = make_level().prob.nvars
    uend, stats = O.run_sdc(make_level, np.array(case['u0']).reshape(shape), meta['t0'], meta['Tend'],
                            num_procs=meta['num_procs'], maxiter=meta['maxiter'],
                            mssdc_jac=meta['controller_params'].get('mssdc_jac', True),
                            skip_residual_computation=tuple(meta['sweeper_params'].get('skip_residual_computation', ())))
    assert [n for _, n in stats['niter']] == list(case['niter'])            # bit-exact iteration counts
    np.testing.assert_allclose([t for t, _ in stats['niter']], case['niter_t'], rtol=0, atol=1e-14)
    assert rel_err(uend, case['uend']) < TOL
    res = [r for _, hist in stats['residuals'] for r in hist]
    np.testing.assert_allclose(res, case['res'], rtol=1e-9, atol=1e-16)


def test_config1_known_answers():
    """SURVEY 3.5 / BASELINE.md: 5 iterations every step, err 1.14e-11."""
    case = load_cases('runs.npz')['config1']
    assert list(case['niter']) == [5] * 10
    assert abs(float(case['err']) - 1.1398e-11) < 1e-14


def test_solver_equivalence_budget():
    """FFT-symbol solve vs SuperLU vs CG(1e-12) on a 3-D 16^3 random rhs at the
    benchmark stiffness; documents the <=1e-10 budget the GPU parity tests use."""
    for order in (2, 4):
        P = O.HeatUnforced((16, 16, 16), 0.1, 2, order=order)
        rhs = np.random.default_rng(1).standard_normal(P.nvars)
        lam_max = 12.0 / P.dx**2 * P.nu
        factor = 63.0 / lam_max
        direct = P.solve_system(rhs, factor, rhs, 0.0)
        spec = O.spectral_solve(P, rhs, factor)
        assert rel_err(spec, direct) < 1e-13
        Pcg = O.HeatUnforced((16, 16, 16), 0.1, 2, order=order, solver_type='CG')
        cgsol = Pcg.solve_system(rhs, factor, np.zeros_like(rhs), 0.0)
        assert rel_err(cgsol, direct) < 1e-10
        assert 20 < Pcg.work_counters['CG'].niter < 200
    A = O.Advection((16, 16), 0.7, 2, order=3, stencil_type='upwind')
    rhs = np.random.default_rng(2).standard_normal(A.nvars)
    assert rel_err(O.spectral_solve(A, rhs, 0.01), A.solve_system(rhs, 0.01, rhs, 0.0)) < 1e-13


def test_fd_stencil_known_answers():
    """literal stencils the reference pins in tests/test_helpers/test_problem_helper.py:6-135."""
    c, s = O.fd_stencil(2, 2, 'center')
    np.testing.assert_allclose(c, [1, -2, 1], atol=1e-14)
    c, s = O.fd_stencil(2, 4, 'center')
    np.testing.assert_allclose(c, [-1 / 12, 4 / 3, -5 / 2, 4 / 3, -1 / 12], atol=1e-14)
    c, s = O.fd_stencil(2, 6, 'center')
    np.testing.assert_allclose(c, [1 / 90, -3 / 20, 3 / 2, -49 / 18, 3 / 2, -3 / 20, 1 / 90], atol=1e-14)
    c, s = O.fd_stencil(1, 2, 'center')
    np.testing.assert_allclose(c, [-0.5, 0, 0.5], atol=1e-14)
    c, s = O.fd_stencil(1, 4, 'center')
    np.testing.assert_allclose(c, [1 / 12, -2 / 3, 0, 2 / 3, -1 / 12], atol=1e-14)
    c, s = O.fd_stencil(1, 1, 'upwind')
    np.testing.assert_allclose(c, [-1, 1], atol=1e-14)
    c, s = O.fd_stencil(1, 2, 'upwind')
    np.testing.assert_allclose(c, [0.5, -2, 1.5], atol=1e-14)
    c, s = O.fd_stencil(1, 3, 'upwind')
    np.testing.assert_allclose(c, [1 / 6, -1, 1 / 2, 1 / 3], atol=1e-14)
    np.testing.assert_array_equal(s, [-2, -1, 0, 1])


def test_vdp_solve_jacobian_vs_reference():
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vdp_jacobian.npz'))
    P = O.VanDerPol(mu=float(g['mu']))
    for i in range(g['u'].shape[1]):
        got = P.solve_jacobian(g['rhs'][:, i], float(g['dt']), g['u'][:, i])
        assert np.array_equal(got, g['out'][:, i])


def _big3d_level(case, spectral):
    meta = case['meta']
    pp = dict(meta['prob_params'])
    pp['nvars'] = tuple(pp['nvars'])
    if spectral:
        pp.pop('solver_type')

        class Prob(O.HeatUnforced):  # exact solve in Fourier space (what the device does; SuperLU is minutes at 64^3)
            def solve_system(self, rhs, factor, u0, t):
                return O.spectral_solve(self, rhs, factor)

        prob = Prob(**pp)
    else:
        prob = O.HeatUnforced(**pp)
    return prob, make_oracle_coll(case)


@pytest.mark.parametrize('spectral', [False, True])
def test_big3d_sweeps_vs_reference(spectral):
    """64^3, M=5, stiffness of the headline workload: the oracle with the reference's CG follows the reference to
    rounding (CG counts identical); with the exact Fourier solve it differs by the reference's CG error only - the
    budget the GPU tests of the fused kernels (tests/test_gpu_configs.py) have to allow for."""
    case = load_cases('sweeps_big3d.npz')['cg64_heat3d_M5_IE']
    meta = case['meta']
    prob, coll = _big3d_level(case, spectral)
    L = O.Level(prob, coll, meta['dt'])
    L.time = meta['t0']
    L.u[0] = np.array(case['u0'])
    O.predict(L, 'spread')
    tol_u, tol_f = (1e-10, 1e-9) if spectral else (1e-12, 1e-11)
    sub = lambda a: np.asarray(a)[..., 1::4, 2::4, 3::4]  # noqa: E731
    for k in range(0, meta['nsweeps'] + 1):
        if k:
            O.sweep(L)
        u, f = np.stack(L.u), np.stack(L.f)
        assert rel_err(sub(u), case[f'k{k}_u_sub']) < tol_u, k
        assert rel_err(sub(f), case[f'k{k}_f_sub']) < tol_f, k
        np.testing.assert_allclose(np.sqrt(np.sum(u.reshape(6, -1) ** 2, axis=1)), case[f'k{k}_u_l2'], rtol=tol_u)
        O.compute_residual(L)
        ref = float(case[f'k{k}_res_full_abs'])
        assert abs(L.status_residual - ref) <= 1e-8 * abs(ref), k
    O.compute_end_point(L, False)
    assert rel_err(L.uend, case['k3_uend_0']) < tol_u
    if not spectral:
        assert prob.work_counters['CG'].niter == int(case['work_CG'][-1])


def test_big3d_run_vs_reference_with_exact_solve():
    """two steps to restol 1e-9 at 64^3: the exact Fourier solve reproduces the iteration counts of the reference's
    CG(1e-12) run and its end value to the CG error (what tests/test_gpu_configs.py asks of the device path)."""
    cases = load_cases('sweeps_big3d.npz')
    rc = cases['cg64_heat3d_run_LU']
    meta = rc['meta']
    prob, coll = _big3d_level(rc, True)
    uend, stats = O.run_sdc(lambda: O.Level(prob, coll, meta['level_params']['dt'], restol=meta['level_params']['restol']),
                            np.array(cases['cg64_heat3d_M5_IE']['u0']), meta['t0'], meta['Tend'], maxiter=meta['maxiter'])
    assert [n for _, n in stats['niter']] == list(rc['niter'])
    assert rel_err(uend, rc['uend']) < 1e-11
    res = [r for _, hist in stats['residuals'] for r in hist]
    np.testing.assert_allclose(res, rc['res'], rtol=1e-5, atol=5e-12)


def test_pin512_imex_case_replayed_by_the_oracle():
    """the 2-D 512^2 advection-diffusion IMEX case that pins config 3's kernel instances (sweeps_pin512.npz, generated by the
    reference with SuperLU) replayed by the oracle with its Fourier solve (equal to the sparse LU to 2e-15,
    test_solver_equivalence_budget): subsampled node values, right-hand sides, end values and residuals"""
    case = load_cases('sweeps_pin512.npz')['pin_advdiff2d_512_M5_IE']
    meta = case['meta']
    pp = {k: tuple(v) if isinstance(v, list) else v for k, v in meta['prob_params'].items()}
    prob = make_oracle_problem('advdiff', pp)
    prob.solve_system = lambda rhs, factor, u0, t: O.spectral_solve(prob.diff, rhs, factor)
    coll = make_oracle_coll(case)
    L = O.Level(prob, coll, meta['dt'])
    L.time = meta['t0']
    n = 512
    x = np.arange(n) / n
    u0 = np.sin(np.pi * 2 * x[None, :]) * np.sin(np.pi * 2 * x[:, None]) + 1e-3 * np.random.default_rng(0).standard_normal((n, n))
    thin = lambda v: np.asarray(v)[..., 1::8] if np.asarray(v).ndim <= 2 else np.asarray(v)[..., 1::8, 5::8]   # noqa: E731
    assert np.array_equal(thin(u0), case['u0_sub'])
    L.u[0] = u0
    O.predict(L, 'spread')
    for k in range(0, meta['nsweeps'] + 1):
        if k > 0:
            O.sweep(L)
        assert rel_err(thin(np.stack(L.u)), case[f'k{k}_u_sub']) < 1e-12, k
        assert rel_err(thin(np.stack(L.f)), case[f'k{k}_f_sub']) < 1e-11, k
        for rt in ('full_abs', 'last_rel'):
            L.residual_type = rt
            O.compute_residual(L)
            ref = float(case[f'k{k}_res_{rt}'])
            assert abs(L.status_residual - ref) <= 1e-9 * abs(ref) + 1e-14, (k, rt)
        O.compute_end_point(L, False)
        assert rel_err(thin(L.uend), case[f'k{k}_uend_0_sub']) < 1e-12, k


def test_pin1024_2d_case_replayed_by_the_oracle():
    """the 2-D 1024^2 heat case behind the bench's own kernel instances (sweeps_pin1024.npz: the reference with SuperLU,
    20 minutes) replayed by the oracle with its Fourier solve: subsampled node values, end values and residuals"""
    case = load_cases('sweeps_pin1024.npz')['pin_heat2d_1024_M5_IE']
    meta = case['meta']
    pp = {k: tuple(v) if isinstance(v, list) else v for k, v in meta['prob_params'].items()}
    prob = make_oracle_problem('heat_unforced', pp)
    prob.solve_system = lambda rhs, factor, u0, t: O.spectral_solve(prob, rhs, factor)
    coll = make_oracle_coll(case)
    L = O.Level(prob, coll, meta['dt'])
    L.time = meta['t0']
    n = 1024
    x = np.arange(n) / n
    u0 = np.sin(np.pi * 2 * x[None, :]) * np.sin(np.pi * 2 * x[:, None]) + 1e-3 * np.random.default_rng(0).standard_normal((n, n))
    thin = lambda v: np.asarray(v)[..., 1::16] if np.asarray(v).ndim <= 2 else np.asarray(v)[..., 1::16, 9::16]   # noqa: E731
    assert np.array_equal(thin(u0), case['u0_sub'])
    L.u[0] = u0
    O.predict(L, 'spread')
    for k in range(0, meta['nsweeps'] + 1):
        if k > 0:
            O.sweep(L)
        assert rel_err(thin(np.stack(L.u)), case[f'k{k}_u_sub']) < 1e-12, k
        L.residual_type = 'full_abs'
        O.compute_residual(L)
        ref = float(case[f'k{k}_res_full_abs'])
        assert abs(L.status_residual - ref) <= 1e-9 * abs(ref) + 1e-14, k
        O.compute_end_point(L, False)
        assert rel_err(thin(L.uend), case[f'k{k}_uend_0_sub']) < 1e-12, k
