"""GPU parity through the drop-in plug-in surface: description dict -> controller -> sweeper_class /
problem_class -> C-ABI.  Compared with golden runs of the reference (tests/golden/runs.npz): iteration counts
bit-exact, end values <= 1e-10 relative (BASELINE.json north_star), residual history to 1e-6 relative."""
import numpy as np
import pytest

from tests._cases import load_cases, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _classes():
    from pysdc_amd import problems as P, sweepers as S

    return ({'heat_unforced': P.heatNd_unforced, 'heat_forced': P.heatNd_forced, 'advection': P.advectionNd,
             'advdiff': P.advectiondiffusionNd_imex},
            {'generic_implicit': S.generic_implicit, 'imex_1st_order': S.imex_1st_order})


def description_from(meta, problem_class=None):
    probs, sweeps = _classes()
    pp = {k: tuple(v) if isinstance(v, list) else v for k, v in meta['prob_params'].items()}
    sp = {k: tuple(v) if isinstance(v, list) else v for k, v in meta['sweeper_params'].items()}  # (lists mean levels)
    return dict(problem_class=problem_class or probs[meta['prob']], problem_params=pp,
                sweeper_class=sweeps[meta['sweeper']], sweeper_params=sp,
                level_params=dict(meta['level_params']), step_params=dict(maxiter=meta['maxiter']))


RUNS = ([('runs.npz', n) for n in load_cases('runs.npz')] + [('runs_dirichlet.npz', n) for n in load_cases('runs_dirichlet.npz')]
        + [('runs_dirichlet_nd.npz', n) for n in load_cases('runs_dirichlet_nd.npz')]
        + [('runs_dirichlet_ho.npz', n) for n in load_cases('runs_dirichlet_ho.npz')]
        + [('runs_neumann.npz', n) for n in load_cases('runs_neumann.npz')]
        + [('runs_skip.npz', n) for n in load_cases('runs_skip.npz')]
        + [('runs_radix3.npz', n) for n in load_cases('runs_radix3.npz')]    # grids of 3 * 2^p points: exact Fourier solve too
        + [('runs_radix5.npz', n) for n in load_cases('runs_radix5.npz')])   # ... and of 5 * 2^p points


@pytest.mark.parametrize('fname,name', RUNS)
@pytest.mark.parametrize('fused', [True, False])
def test_run_vs_golden(fname, name, fused):
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted

    case = load_cases(fname)[name]
    meta = case['meta']
    probs, _ = _classes()
    pc = probs[meta['prob']]
    if not fused:
        if meta['num_procs'] > 2 or name == 'config1':
            pytest.skip('node-by-node path covered on the smaller cases')
        pc = type(pc.__name__ + '_nodewise', (pc,), {'fused': False})
    desc = description_from(meta, pc)
    C = controller_nonMPI(meta['num_procs'], dict(logger_level=40, **meta['controller_params']), desc)
    P = C.MS[0].levels[0].prob
    u0 = P.u_init
    u0[:] = case['u0']
    uend, stats = C.run(u0, meta['t0'], meta['Tend'])
    niter = get_sorted(stats, type='niter', sortby='time')
    assert [v for _, v in niter] == list(case['niter'])
    np.testing.assert_allclose([t for t, _ in niter], case['niter_t'], rtol=0, atol=1e-14)
    assert rel_err(uend.get(), case['uend']) < TOL
    res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]
    # residuals are differences of O(|u|) quantities: below ~1e-11 |u| they are rounding noise in both codes
    np.testing.assert_allclose(res, case['res'], rtol=1e-6, atol=1e-11 * max(1.0, float(np.max(np.abs(case['u0'])))))


def test_datatype_semantics():
    """mesh behaviour the controllers / transfer classes rely on (tests/tests_core.py:18-60)."""
    from pysdc_amd.hip_mesh import hip_mesh

    init = ((4, 8), None, np.dtype('float64'))
    a = hip_mesh(init, val=1.5)
    b = hip_mesh(a)
    b += a
    c = 2.0 * a - b * 0.5 + a
    assert isinstance(c, hip_mesh) and c.shape == (4, 8)
    assert np.all(c.get() == 2.0 * 1.5 - 3.0 * 0.5 + 1.5)
    assert abs(c) == 3.0 and isinstance(abs(c), float)
    assert np.all(a.get() == 1.5)          # copy constructor is deep (mesh.py:36-38)
    a[:] = np.arange(32.0).reshape(4, 8)
    assert a.flatten().shape == (32,) and np.array_equal(np.asarray(a), np.arange(32.0).reshape(4, 8))
    with pytest.raises(TypeError):
        import pickle

        pickle.dumps(a)


def test_level_views_and_errors():
    from pysdc_amd.level import Step
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit
    from pysdc_amd.errors import ProblemError, ParameterError

    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(16, 16), nu=0.1, freq=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT'),
                level_params=dict(dt=0.01), step_params=dict(maxiter=5))
    S = Step(desc)
    L = S.levels[0]
    assert L.u[0] is None and L.uend is None and L.tau[0] is None
    with pytest.raises(AssertionError):
        L.status.unlocked = False
        L.u[0] = L.prob.u_exact(0.0)
        for m in range(1, 4):
            L.u[m] = L.u[0]
            L.f[m] = L.prob.eval_f(L.u[0], 0.0)
        L.sweep.update_nodes()             # assert L.status.unlocked (generic_implicit.py:63)
    L.status.time = 0.0
    L.sweep.predict()
    r0 = None
    L.sweep.compute_residual()
    r0 = L.status.residual
    L.u[2] += L.u[1]                        # in-place write through a view invalidates the cached residual
    L.sweep.compute_residual()
    assert L.status.residual != r0
    L.params.residual_type = 'nonsense'
    with pytest.raises(ParameterError):
        L.sweep.compute_residual()
    with pytest.raises(ProblemError):
        heatNd_unforced(nvars=(16, 8))
    with pytest.raises(ProblemError):
        heatNd_unforced(nvars=(15, 15))
    with pytest.raises(ProblemError):
        heatNd_unforced(nvars=16, freq=3)
    with pytest.raises(ProblemError):
        heatNd_unforced(nvars=(4, 4, 4, 4))
    with pytest.raises(ParameterError):
        generic_implicit(dict(quad_type='RADAU-RIGHT'), None)


def test_device_init_field_matches_host():
    import ctypes as C
    from pysdc_amd.engine import SweepEngine
    from pysdc_amd import lib as Lb
    from pysdc_amd.synth import init_field

    for nv in ((64,), (16, 16), (8, 8, 8)):
        e = SweepEngine(nv, 2)
        freq = (C.c_int * 3)(2, 4, 2)
        Lb.check(e.lib.sdc_init_field(e.ctx, e.ptr(Lb.SLOT_U, 0), freq, 1e-3, 7), e.ctx)
        host = init_field(nv, (2, 4, 2)[: len(nv)], 1e-3, 7)
        assert np.max(np.abs(e.download(Lb.SLOT_U, 0) - host)) < 1e-14
        e.close()


def test_vdp_ensemble_vs_golden():
    """BASELINE config 4 semantics: every trajectory of the ensemble follows the reference's vanderpol
    (golden single-trajectory sweeps): node values <= 1e-10 relative, Newton iteration total bit-exact."""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import vanderpol_ensemble
    from pysdc_amd.sweepers import generic_implicit

    cases = load_cases('sweeps_vdp.npz')
    names = sorted(cases)
    meta = cases[names[0]]['meta']
    u0 = np.stack([cases[n]['u0'] for n in names], axis=1)          # (2, ntraj)
    desc = dict(problem_class=vanderpol_ensemble,
                problem_params=dict(ntraj=len(names), u0=u0, mu=5.0, newton_tol=1e-9),
                sweeper_class=generic_implicit, sweeper_params=dict(meta['sweeper_params']),
                level_params=dict(dt=meta['dt']), step_params=dict(maxiter=4))
    for fused in (True, False):
        pc = vanderpol_ensemble if fused else type('vdp_nodewise', (vanderpol_ensemble,), {'fused': False})
        desc['problem_class'] = pc
        S = Step(desc)
        L = S.levels[0]
        L.status.time = meta['t0']
        L.u[0] = L.prob.u_exact(0.0)
        L.sweep.predict()
        for k in range(1, meta['nsweeps'] + 1):
            L.sweep.update_nodes()
            L.sweep.compute_residual()
            U = np.stack([np.asarray(L.u[m]) for m in range(6)])     # (M+1, 2, ntraj)
            F = np.stack([np.asarray(L.f[m]) for m in range(6)])
            for i, n in enumerate(names):
                assert rel_err(U[:, :, i], cases[n][f'k{k}_u']) < TOL, (n, k)
                assert rel_err(F[:, :, i], cases[n][f'k{k}_f']) < 1e-9, (n, k)
            ref_res = max(float(cases[n][f'k{k}_res_full_abs']) for n in names)
            assert abs(L.status.residual - ref_res) <= 1e-9 * max(ref_res, 1e-6)
        assert L.prob.work_counters['newton'].niter == sum(int(cases[n]['work_newton'][-1]) for n in names)
        assert L.prob.work_counters['rhs'].niter == sum(int(cases[n]['work_rhs'][-1]) for n in names)


def test_vdp_spread_predictor_is_put_off_until_somebody_reads_it():
    """the node copies of a spread predictor (core/sweeper.py:129-143) are not written for the ensemble either: the first
    sweep reads u[0] alone.  Reading them before the sweep delivers u[0] and f(u[0]) at every node; sweeping afterwards
    gives the same iterate and counts as sweeping at once."""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import vanderpol_ensemble
    from pysdc_amd.sweepers import generic_implicit

    rng = np.random.default_rng(2)
    u0 = rng.uniform(-2, 2, size=(2, 777))
    desc = dict(problem_class=vanderpol_ensemble, problem_params=dict(ntraj=777, u0=u0, mu=5.0, newton_tol=1e-9),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=5, quad_type='RADAU-RIGHT', QI='LU'),
                level_params=dict(dt=0.05), step_params=dict(maxiter=4))
    out = []
    for peek in (False, True):
        L = Step(desc).levels[0]
        L.status.time = 0.0
        L.u[0] = L.prob.u_exact(0.0)
        L.sweep.predict()
        L.sweep.update_nodes()            # (the first sweep of a level creates the views of the node fields: addresses are
        L.engine.profile_enable(True)     # handed out, so everything deferred is written once - not part of what is counted)
        newton0, rhs0 = L.prob.work_counters['newton'].niter, L.prob.work_counters['rhs'].niter
        L.sweep.predict()
        if peek:
            f0 = np.stack([u0[1], 5.0 * (1 - u0[0] ** 2) * u0[1] - u0[0]])
            for m in range(1, 6):
                assert np.array_equal(np.asarray(L.u[m]), u0)
                np.testing.assert_allclose(np.asarray(L.f[m]), f0, rtol=1e-15, atol=0)
        L.sweep.compute_residual()
        res0 = L.status.residual
        L.sweep.update_nodes()
        L.sweep.compute_residual()
        prof = L.engine.profile_read()
        assert ('spread' in prof) == peek, prof.keys()
        out.append((res0, L.status.residual, np.stack([np.asarray(L.u[m]) for m in range(6)]),
                    L.prob.work_counters['newton'].niter - newton0, L.prob.work_counters['rhs'].niter - rhs0))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2]) and out[0][3:] == out[1][3:]


@pytest.mark.parametrize('ntraj', [9, 200])
def test_vdp_mfma_block_solver_vs_golden(ntraj):
    """the Newton block solves on the matrix cores (v_mfma_f64_4x4x4, two trajectories per block): the nine golden
    trajectories of the reference (tiled to fill several waves, with a ragged tail) - node values <= 1e-10, Newton and
    right-hand-side counts identical to the reference's, identical to the closed-form kernel's."""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import vanderpol_ensemble
    from pysdc_amd.sweepers import generic_implicit

    cases = load_cases('sweeps_vdp.npz')
    names = sorted(cases)
    T = len(names)
    meta = cases[names[0]]['meta']
    tile = np.stack([cases[n]['u0'] for n in names], axis=1)
    u0 = np.tile(tile, (1, -(-ntraj // T)))[:, :ntraj]
    full, rest = divmod(ntraj, T)
    out = {}
    for kind in ('mfma', 'closed_form'):
        desc = dict(problem_class=vanderpol_ensemble,
                    problem_params=dict(ntraj=ntraj, u0=u0, mu=5.0, newton_tol=1e-9, block_solver=kind),
                    sweeper_class=generic_implicit, sweeper_params=dict(meta['sweeper_params']),
                    level_params=dict(dt=meta['dt']), step_params=dict(maxiter=4))
        L = Step(desc).levels[0]
        L.status.time = meta['t0']
        L.u[0] = L.prob.u_exact(0.0)
        L.sweep.predict()
        L.engine.profile_enable(True)
        for k in range(1, meta['nsweeps'] + 1):
            L.sweep.update_nodes()
            L.sweep.compute_residual()
            U = np.stack([np.asarray(L.u[m]) for m in range(6)])
            for i in range(ntraj):
                assert rel_err(U[:, :, i], cases[names[i % T]][f'k{k}_u']) < TOL, (kind, i, k)
            per = [int(cases[n]['work_newton'][k - 1]) for n in names]
            assert L.prob.work_counters['newton'].niter == full * sum(per) + sum(per[:rest]), (kind, k)
        assert ('vdp_sweep_mfma' in L.engine.profile_read()) == (kind == 'mfma')
        out[kind] = U
    assert np.max(np.abs(out['mfma'] - out['closed_form'])) < 1e-13


def test_vdp_newton_failure_raises():
    from pysdc_amd.level import Step
    from pysdc_amd.problems import vanderpol_ensemble
    from pysdc_amd.sweepers import generic_implicit
    from pysdc_amd.errors import ProblemError

    desc = dict(problem_class=vanderpol_ensemble,
                problem_params=dict(ntraj=64, mu=5.0, newton_tol=1e-14, newton_maxiter=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU'),
                level_params=dict(dt=0.5), step_params=dict(maxiter=4))
    L = Step(desc).levels[0]
    L.status.time = 0.0
    L.u[0] = L.prob.u_exact(0.0)
    L.sweep.predict()
    with pytest.raises(ProblemError):
        L.sweep.update_nodes()


def test_slab_view_as_torch_aliases_device_memory():
    """the tensors handed to torch.distributed (RCCL send / recv / broadcast) alias the engine's slabs."""
    import torch
    from pysdc_amd.level import Step
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    S = Step(dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(16, 16), nu=0.1, freq=2),
                  sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT'),
                  level_params=dict(dt=0.01), step_params=dict(maxiter=5)))
    L = S.levels[0]
    L.u[0] = L.prob.u_exact(0.0)
    t = L.u[0].as_torch()
    assert t.is_cuda and t.dtype == torch.float64 and t.numel() == 256 and t.data_ptr() == L.u[0].ptr
    t.mul_(2.0)
    torch.cuda.synchronize()
    assert np.allclose(L.u[0].get(), 2.0 * L.prob.u_exact(0.0).get())
    own = L.prob.u_init
    assert own.as_torch().data_ptr() == own.ptr


def test_bench_single_rank_distributed_path():
    """bench.py through controller_dist + torch.distributed (nccl world of one rank, gloo side group):
    the multi-GPU code path minus the neighbour exchange."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29547')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--n', '64', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline', '--force-dist'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['finite'] and d['value'] > 0
    ref = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--n', '64', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=600)
    assert ref.returncode == 0, ref.stderr[-2000:]


DIRICHLET_SWEEPS = ([('sweeps_heat.npz', 'heat1d_dirichlet'), ('sweeps_imex.npz', 'forced1d_dirichlet')]
                    + [('sweeps_dirichlet_nd.npz', n) for n in load_cases('sweeps_dirichlet_nd.npz')]
                    + [('sweeps_dirichlet_ho.npz', n) for n in load_cases('sweeps_dirichlet_ho.npz')]
                    + [('sweeps_neumann.npz', n) for n in load_cases('sweeps_neumann.npz')])   # Neumann / mixed ends, one-sided stencils


@pytest.mark.parametrize('fname,name', DIRICHLET_SWEEPS)
@pytest.mark.parametrize('fused', [True, False])
def test_dirichlet_sweeps_vs_golden(fname, name, fused):
    """dirichlet-zero against golden sweeps of the reference: 1-D (odd extension inside the engine, fused sweeps), 2-D /
    3-D (fields packed into their odd extension around eval_f / solve_system, node-by-node sweeps), and stencils of order
    4 / 6 / 8 whose rows next to the boundary carry the reference's shifted one-sided stencils (banded row table, GMRES to
    round-off in place of the sparse LU - or the reference's own GMRES with its iteration counts)."""
    from pysdc_amd.level import Step

    case = load_cases(fname)[name]
    meta = case['meta']
    probs, sweeps = _classes()
    pc = probs[meta['prob']]
    pp = {k: tuple(v) if isinstance(v, list) else v for k, v in meta['prob_params'].items()}
    probe = pc(**pp)
    if fused and not probe.fused:
        pytest.skip('2-D / 3-D odd extensions and banded IMEX levels sweep node by node')
    if not fused:
        pc = type(pc.__name__ + '_nodewise', (pc,), {'fused': False})
    S = Step(dict(problem_class=pc, problem_params=pp, sweeper_class=sweeps[meta['sweeper']],
                  sweeper_params=dict(meta['sweeper_params']), level_params=dict(dt=meta['dt']),
                  step_params=dict(maxiter=10)))
    L = S.levels[0]
    L.status.time = meta['t0']
    u0 = L.prob.u_init
    u0[:] = case['u0']
    L.u[0] = u0
    L.sweep.predict()

    def check(tag):
        assert rel_err(np.stack([np.asarray(x) for x in L.u]), case[f'{tag}_u']) < TOL, tag
        assert rel_err(np.stack([np.asarray(x) for x in L.f]), case[f'{tag}_f']) < TOL, tag
        for rt in ('full_abs', 'last_rel'):
            L.params.residual_type = rt
            L.sweep.compute_residual()
            ref = float(case[f'{tag}_res_{rt}'])
            assert abs(L.status.residual - ref) <= 1e-8 * abs(ref) + 1e-11, (tag, rt)
        L.params.residual_type = 'full_abs'
        L.sweep.compute_end_point()
        assert rel_err(L.uend.get(), case[f'{tag}_uend_0']) < TOL, tag

    check('k0')
    for k in range(1, meta['nsweeps'] + 1):
        L.sweep.update_nodes()
        check(f'k{k}')
        for key in L.prob.work_counters:          # the reference's solver, its counts (GMRES on the banded operator)
            assert L.prob.work_counters[key].niter == int(case[f'work_{key}'][k - 1]), (key, k)


@pytest.mark.parametrize('virtual', [0, 8])
def test_views_kept_across_sweeps_see_current_values(virtual):
    """the engine defers storing F[1..M] and the predictor's node copies (sdc_set_deferred); a view taken once
    and read later must still show what the reference's L.f[m] / L.u[m] would hold at that moment.  (virtual > 0:
    iterates recomputed from the transform of u[0] - same values to round-off instead of the same bits.)"""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    n = 64
    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=(2, 2, 2)),
                sweeper_class=generic_implicit,
                sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU'),
                level_params=dict(dt=1e-2), step_params=dict(maxiter=4))
    out = []
    for deferred in (True, False):
        S = Step(desc)
        L = S.levels[0]
        L.status.time = 0.0
        L.u[0] = L.prob.u_exact(0.0)
        L.engine.set_deferred(deferred)
        L.engine.set_virtual_sweeps(virtual)
        L.sweep.predict()
        u2, f2 = L.u[2], L.f[2]                  # views taken while the spread is still pending
        seen = [u2.get(), f2.get()]
        L.sweep.update_nodes()
        L.sweep.compute_residual()
        seen += [abs(f2), u2.get(), L.status.residual]
        L.sweep.update_nodes()
        g = f2 + f2                               # arithmetic on a kept view
        seen += [g.get(), (2.0 * L.f[3]).get()]
        out.append(seen)
    for i, (x, y) in enumerate(zip(*out)):
        if i == 4:   # the residual: deferred mode reduces it from its Fourier transform (round-off level change)
            assert abs(x - y) <= 1e-9 * abs(y)
        elif virtual:
            assert rel_err(np.asarray(x), np.asarray(y)) < 1e-13, i
        else:
            assert np.array_equal(np.asarray(x), np.asarray(y)), i


def test_refresh_f0_is_evaluated_on_demand():
    """Level.refresh_f0 (f[0] = f(u[0]) after a receive, controller_MPI.py:233) defers the evaluation; reading
    L.f[0] afterwards gives f of the NEW u[0]."""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import heatNd_unforced, advectiondiffusionNd_imex
    from pysdc_amd.sweepers import generic_implicit, imex_1st_order

    for pc, sc, extra in ((heatNd_unforced, generic_implicit, {}), (advectiondiffusionNd_imex, imex_1st_order, dict(c=1.0))):
        desc = dict(problem_class=pc, problem_params=dict(nvars=(16, 16, 16), nu=0.1, freq=(2, 2, 2), **extra),
                    sweeper_class=sc, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT'),
                    level_params=dict(dt=1e-2), step_params=dict(maxiter=4))
        S = Step(desc)
        L = S.levels[0]
        L.status.time = 0.0
        L.u[0] = L.prob.u_exact(0.0)
        L.sweep.predict()
        L.sweep.update_nodes()
        new = L.prob.u_exact(0.3)
        new *= 1.5
        L.u[0] = new                                   # what a receive does
        L.refresh_f0()
        L.sweep.update_nodes()                         # sweeps do not need f[0]
        want = L.prob.eval_f(new, 0.0)
        got = L.f[0]
        assert np.array_equal(np.asarray(got), np.asarray(want))


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_plugin_random_walk_deferred_vs_eager(seed):
    """the same random sequence of sweeper calls and datatype operations on L.u / L.f views of two levels, one
    with deferred node fields and one eager: everything a user can read must agree."""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    n, M = 64, 3
    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=(2, 2, 2)),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='LU'),
                level_params=dict(dt=2e-2), step_params=dict(maxiter=4))
    levels = []
    for deferred in (True, False):
        S = Step(desc)
        Lv = S.levels[0]
        Lv.status.time = 0.0
        Lv.u[0] = Lv.prob.u_exact(0.0)
        Lv.engine.set_deferred(deferred)
        Lv.sweep.predict()
        levels.append((S, Lv))
    rng = np.random.default_rng(seed)
    ops = ['sweep'] * 5 + ['residual'] * 2 + ['end_point', 'read_u', 'read_f', 'scale_u', 'iadd_f', 'set_u0', 'integrate',
                                              'predict', 'abs_u', 'copy_u', 'kept_view']
    kept = [None, None]
    trace = []

    def same(x, y):
        x, y = np.asarray(x), np.asarray(y)
        assert np.max(np.abs(x - y)) <= 1e-12 * max(1.0, float(np.max(np.abs(y)))), ' '.join(trace)

    for _ in range(40):
        op = ops[rng.integers(len(ops))]
        m = int(rng.integers(1, M + 1))
        trace.append(f'{op}{m}')
        outs = []
        for i, (S, Lv) in enumerate(levels):
            if op == 'sweep':
                Lv.sweep.update_nodes()
            elif op == 'residual':
                Lv.sweep.compute_residual()
                outs.append(Lv.status.residual)
            elif op == 'end_point':
                Lv.sweep.compute_end_point()
                outs.append(Lv.uend.get())
            elif op == 'read_u':
                outs.append(Lv.u[m].get())
            elif op == 'read_f':
                outs.append(Lv.f[m].get())
            elif op == 'scale_u':
                Lv.u[m] = 0.5 * Lv.u[m]
            elif op == 'iadd_f':
                Lv.f[m] += Lv.f[0]
            elif op == 'set_u0':
                Lv.u[0] = Lv.u[m] + Lv.u[0]
                Lv.f[0] = Lv.prob.eval_f(Lv.u[0], 0.0)
            elif op == 'integrate':
                outs.append(np.stack([x.get() for x in Lv.sweep.integrate()]))
            elif op == 'predict':
                Lv.sweep.predict()
            elif op == 'abs_u':
                outs.append(abs(Lv.u[m] - Lv.u[0]))
            elif op == 'copy_u':
                outs.append(Lv.prob.dtype_u(Lv.u[m]).get())
            elif op == 'kept_view':
                if kept[i] is None:
                    kept[i] = (Lv.u[m], Lv.f[m])
                outs.append(np.stack([kept[i][0].get(), kept[i][1].get()]))
        if outs:
            if op == 'residual':
                assert abs(outs[0] - outs[1]) <= 1e-7 * abs(outs[1]) + 1e-13, ' '.join(trace)
            else:
                same(outs[0], outs[1])
    same(np.stack([levels[0][1].u[k].get() for k in range(M + 1)]), np.stack([levels[1][1].u[k].get() for k in range(M + 1)]))
    same(np.stack([levels[0][1].f[k].get() for k in range(M + 1)]), np.stack([levels[1][1].f[k].get() for k in range(M + 1)]))


KRYLOV = [('sweeps_cg.npz', n) for n in load_cases('sweeps_cg.npz')] + [('sweeps_gmres.npz', n) for n in load_cases('sweeps_gmres.npz')]


@pytest.mark.parametrize('fname,name', KRYLOV)
@pytest.mark.parametrize('fused', [True, False])
def test_cg_sweeps_vs_golden(fname, name, fused):
    """solver_type='CG' (generic_ND_FD.py:252-260) and 'GMRES' (:241-250, the non-symmetric advection operators among the
    cases) on the device: sweeps against the reference's sweeps with scipy's solvers, node values to 1e-10 and the
    accumulated work_counters['CG'] / ['GMRES'] after every sweep (fused: the engine's node loop; not fused:
    solve_system called from the host node loop)."""
    from pysdc_amd.level import Step

    case = load_cases(fname)[name]
    meta = case['meta']
    key = meta['prob_params']['solver_type']
    probs, sweeps = _classes()
    pc = probs[meta['prob']]
    if not fused:
        pc = type(pc.__name__ + '_nodewise', (pc,), {'fused': False})
    pp = dict(meta['prob_params'])
    if isinstance(pp.get('nvars'), list):
        pp['nvars'] = tuple(pp['nvars'])
    S = Step(dict(problem_class=pc, problem_params=pp, sweeper_class=sweeps[meta['sweeper']],
                  sweeper_params=dict(meta['sweeper_params']), level_params=dict(dt=meta['dt']),
                  step_params=dict(maxiter=10)))
    L = S.levels[0]
    L.status.time = meta['t0']
    u0 = L.prob.u_init
    u0[:] = case['u0']
    L.u[0] = u0
    L.sweep.predict()
    counts = []
    for k in range(1, meta['nsweeps'] + 1):
        L.sweep.update_nodes()
        assert rel_err(np.stack([np.asarray(x) for x in L.u]), case[f'k{k}_u']) < TOL, k
        assert rel_err(np.stack([np.asarray(x) for x in L.f]), case[f'k{k}_f']) < 1e-8, k
        L.sweep.compute_residual()
        ref = float(case[f'k{k}_res_full_abs'])
        assert abs(L.status.residual - ref) <= 1e-7 * abs(ref) + 1e-11, k
        counts.append(L.prob.work_counters[key].niter)
    assert counts == list(case[f'work_{key}']), (counts, list(case[f'work_{key}']))


def test_vdp_solve_jacobian_vs_reference():
    """vanderpol_ensemble.solve_jacobian against the reference's vanderpol.solve_jacobian, 64 random states."""
    import os
    from pysdc_amd.problems import vanderpol_ensemble

    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vdp_jacobian.npz'))
    P = vanderpol_ensemble(ntraj=g['u'].shape[1], u0=g['u'], mu=float(g['mu']))
    u, r = P.u_init, P.u_init
    u[:] = g['u']
    r[:] = g['rhs']
    before = P.work_counters['jacobian_solves'].niter
    got = P.solve_jacobian(r, float(g['dt']), u).get()
    assert np.max(np.abs(got - g['out'])) <= 1e-14 * np.max(np.abs(g['out']))
    assert P.work_counters['jacobian_solves'].niter - before == g['u'].shape[1]


@pytest.mark.parametrize('nvars', [(100,), (48, 48), (12, 12, 12)])
def test_grids_without_fft_fall_back_to_cg(nvars):
    """periodic grids whose size is not a power of two: the exact Fourier solve is not available, the engine solves
    to round-off with conjugate gradients instead (symmetric operators).  Sweeps against the oracle's direct solve."""
    from oracle import sdc_oracle as O
    from pysdc_amd.level import Step
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    M, dt = 3, 0.01
    S = Step(dict(problem_class=heatNd_unforced, problem_params=dict(nvars=nvars, nu=0.1, freq=2),
                  sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='LU'),
                  level_params=dict(dt=dt), step_params=dict(maxiter=10)))
    L = S.levels[0]
    L.status.time = 0.0
    rng = np.random.default_rng(5)
    u0h = np.asarray(O.HeatUnforced(nvars if len(nvars) > 1 else nvars[0], 0.1, 2).u_exact(0.0)) + 1e-2 * rng.standard_normal(nvars)
    u0 = L.prob.u_init
    u0[:] = u0h
    L.u[0] = u0
    L.sweep.predict()
    sw = L.sweep
    OL = O.Level(O.HeatUnforced(nvars if len(nvars) > 1 else nvars[0], 0.1, 2),
                 O.Coll(sw.coll.nodes, sw.coll.weights, sw.coll.Qmat, sw.QI), dt)
    OL.time = 0.0
    OL.u[0] = u0h.copy()
    O.predict(OL, 'spread')
    for k in range(3):
        L.sweep.update_nodes()
        O.sweep(OL)
        assert rel_err(np.stack([np.asarray(x) for x in L.u]), np.stack(OL.u)) < TOL, k
        L.sweep.compute_residual()
        O.compute_residual(OL)
        assert abs(L.status.residual - OL.status_residual) <= 1e-7 * OL.status_residual + 1e-12


@pytest.mark.parametrize('nvars,stencil', [((100,), 'center'), ((36, 36), 'upwind')])
def test_grids_without_fft_fall_back_to_gmres_for_advection(nvars, stencil):
    """... and with GMRES for operators that are not symmetric (advectionNd): sweeps against the oracle's direct solve"""
    from oracle import sdc_oracle as O
    from pysdc_amd.level import Step
    from pysdc_amd.problems import advectionNd
    from pysdc_amd.sweepers import generic_implicit

    M, dt = 3, 0.01
    order = 2 if stencil == 'center' else 1
    pp = dict(nvars=nvars if len(nvars) > 1 else nvars[0], c=1.0, freq=2, order=order, stencil_type=stencil)
    S = Step(dict(problem_class=advectionNd, problem_params=dict(pp, nvars=nvars), sweeper_class=generic_implicit,
                  sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='LU'), level_params=dict(dt=dt),
                  step_params=dict(maxiter=10)))
    L = S.levels[0]
    L.status.time = 0.0
    OP = O.Advection(**pp)
    u0h = np.asarray(OP.u_exact(0.0)) + 1e-2 * np.random.default_rng(5).standard_normal(nvars)
    u0 = L.prob.u_init
    u0[:] = u0h
    L.u[0] = u0
    L.sweep.predict()
    sw = L.sweep
    OL = O.Level(OP, O.Coll(sw.coll.nodes, sw.coll.weights, sw.coll.Qmat, sw.QI), dt)
    OL.time = 0.0
    OL.u[0] = u0h.copy()
    O.predict(OL, 'spread')
    for k in range(3):
        L.sweep.update_nodes()
        O.sweep(OL)
        assert rel_err(np.stack([np.asarray(x) for x in L.u]), np.stack(OL.u)) < TOL, k
    assert 'GMRES' not in L.prob.work_counters          # the fallback is not the user's solver: nothing is counted


def test_log_solution_hook_keeps_owning_copies():
    """hooks/log_solution.py: 'u' per step; the stored values must not change when the slab is reused."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.hooks import LogSolution
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.stats import get_sorted
    from pysdc_amd.sweepers import generic_implicit

    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(32, 32), nu=0.1, freq=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU'),
                level_params=dict(dt=0.01, restol=1e-10), step_params=dict(maxiter=20))
    C = controller_nonMPI(1, dict(logger_level=40, hook_class=[LogSolution]), desc)
    P = C.MS[0].levels[0].prob
    uend, stats = C.run(P.u_exact(0.0), 0.0, 0.03)
    us = get_sorted(stats, type='u', sortby='time')
    assert [round(t, 10) for t, _ in us] == [0.01, 0.02, 0.03]
    assert np.array_equal(us[-1][1].get(), uend.get())
    for (t, u) in us:
        assert rel_err(u.get(), P.u_exact(t).get()) < 1e-4
    assert np.max(np.abs(us[0][1].get() - us[1][1].get())) > 0


@pytest.mark.parametrize('name', list(load_cases('sweeps_guess.npz')))
@pytest.mark.parametrize('fused', [True, False])
def test_predictor_variants_vs_golden(name, fused):
    """initial_guess = spread / copy / zero / random on problems with and without explicit parts (a time-dependent
    forcing, an explicit stencil): predictor state and three sweeps against the reference."""
    from pysdc_amd.level import Step

    case = load_cases('sweeps_guess.npz')[name]
    meta = case['meta']
    probs, sweeps = _classes()
    pc = probs[meta['prob']]
    if not fused:
        pc = type(pc.__name__ + '_nodewise', (pc,), {'fused': False})
    pp = dict(meta['prob_params'])
    if isinstance(pp.get('nvars'), list):
        pp['nvars'] = tuple(pp['nvars'])
    S = Step(dict(problem_class=pc, problem_params=pp, sweeper_class=sweeps[meta['sweeper']],
                  sweeper_params=dict(meta['sweeper_params']), level_params=dict(dt=meta['dt']),
                  step_params=dict(maxiter=10)))
    L = S.levels[0]
    L.status.time = meta['t0']
    u0 = L.prob.u_init
    u0[:] = case['u0']
    L.u[0] = u0
    L.sweep.predict()

    def check(tag):
        assert rel_err(np.stack([np.asarray(x) for x in L.u]), case[f'{tag}_u']) < TOL, tag
        assert rel_err(np.stack([np.asarray(x) for x in L.f]), case[f'{tag}_f']) < TOL, tag
        L.sweep.compute_residual()
        ref = float(case[f'{tag}_res_full_abs'])
        assert abs(L.status.residual - ref) <= 1e-8 * abs(ref) + 1e-11, tag

    check('k0')
    for k in range(1, meta['nsweeps'] + 1):
        L.sweep.update_nodes()
        check(f'k{k}')


SKIP_ALL = ('IT_CHECK', 'IT_FINE', 'IT_DOWN', 'IT_UP', 'IT_COARSE')


@pytest.mark.parametrize('virtual', [0, 8])
@pytest.mark.parametrize('kind', ['heat3d', 'heat2d', 'advdiff3d', 'forced2d'])
def test_skip_residual_computation_moves_only_the_iterate(kind, virtual):
    """skip_residual_computation for every stage (core/sweeper.py:176-179): the residual stays what it was, the
    iterates and the end value are those of the run that computes it; the engine sweeps with one pointwise pass over
    the cached transforms (no inverse transform); a residual asked for afterwards is still the right one.
    virtual > 0: such a sweep launches nothing at all - the iterate is written out (spec_store) when somebody asks."""
    from pysdc_amd import problems as P, sweepers as S
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted

    pc, sc, pp, sp = {
        'heat3d': (P.heatNd_unforced, S.generic_implicit, dict(nvars=(32, 32, 32), nu=0.1, freq=(2, 2, 2)), dict(QI='LU')),
        'heat2d': (P.heatNd_unforced, S.generic_implicit, dict(nvars=(64, 64), nu=0.1, freq=(2, 4)), dict(QI='IE')),
        'advdiff3d': (P.advectiondiffusionNd_imex, S.imex_1st_order, dict(nvars=(32, 32, 32), nu=0.05, c=0.7, freq=(2, 2, 2)),
                      dict(QI='LU', QE='EE')),
        'forced2d': (P.heatNd_forced, S.imex_1st_order, dict(nvars=(64, 64), nu=0.1, freq=(2, 2)), dict(QI='LU', QE='EE')),
    }[kind]
    out = {}
    for skip in (False, True):
        desc = dict(problem_class=pc, problem_params=dict(pp), sweeper_class=sc,
                    sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT',
                                        skip_residual_computation=SKIP_ALL if skip else (), **sp),
                    level_params=dict(dt=5e-3, restol=-1.0), step_params=dict(maxiter=4))
        C = controller_nonMPI(1, dict(logger_level=40), desc)
        L = C.MS[0].levels[0]
        L.engine.profile_enable(True)
        L.engine.set_virtual_sweeps(virtual)
        uend, stats = C.run(L.prob.u_exact(0.0), 0.0, 3 * 5e-3)
        prof = L.engine.profile_read()
        niter = [v for _, v in get_sorted(stats, type='niter', sortby='time')]
        L.sweep.params.skip_residual_computation = ()
        L._res_cache = None
        L.sweep.compute_residual()   # of the last iterate, on demand
        out[skip] = (uend.get(), niter, L.status.residual, [L.u[m].get() for m in range(1, 4)], prof)
    a, b = out[False], out[True]
    assert a[1] == b[1] == [4, 4, 4]
    assert rel_err(b[0], a[0]) < 1e-13
    for x, y in zip(a[3], b[3]):
        assert rel_err(y, x) < 1e-13
    assert abs(a[2] - b[2]) <= 1e-9 * abs(a[2]) + 1e-15
    names = {k.split('[')[0] for k in b[4]}
    if virtual and kind != 'forced2d':   # (a forced iterate is not a function of u[0] alone: stored as before)
        assert 'spec_point_only' not in names and names & {'spec_store', 'spec_store_last'}, names
    else:
        assert 'spec_point_only' in names, names
    assert not names & {'spec_z_res', 'spec_point_res', 'spec_z_res_spread', 'fft_x_norm'}, names
    assert 'spec_point_only' not in {k.split('[')[0] for k in a[4]}


@pytest.mark.parametrize('M,quad,QI', [(1, 'RADAU-RIGHT', 'IE'), (2, 'RADAU-RIGHT', 'LU'), (2, 'LOBATTO', 'IE'), (4, 'GAUSS', 'LU'),
                                       (3, 'RADAU-LEFT', 'LU'), (6, 'RADAU-RIGHT', 'LU'), (7, 'LOBATTO', 'IE'),
                                       (8, 'RADAU-RIGHT', 'IE'), (8, 'GAUSS', 'LU')])
@pytest.mark.parametrize('kind', ['heat2d', 'advdiff3d'])
def test_node_counts_and_quadrature_types_vs_oracle(M, quad, QI, kind):
    """the smallest and the largest node counts the engine takes (1 .. 8: above 5 the spectral sweep is the pointwise
    kernel plus separate passes), nodes that include the left end (LOBATTO, RADAU-LEFT: first node = t0) or exclude the
    right end (GAUSS, RADAU-LEFT: the end value is the collocation update): two steps of 3 sweeps against the oracle
    driven with the same coefficients."""
    from oracle import sdc_oracle as O
    from pysdc_amd import problems as P, sweepers as S
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted
    from tests._cases import make_oracle_problem

    if kind == 'heat2d':
        pc, sc, pp, extra, oname = P.heatNd_unforced, S.generic_implicit, dict(nvars=(32, 32), nu=0.1, freq=(2, 4)), {}, 'heat_unforced'
    else:
        pc, sc, pp, extra, oname = (P.advectiondiffusionNd_imex, S.imex_1st_order, dict(nvars=(16, 16, 16), nu=0.05, c=0.7, freq=(2, 2, 2)),
                                    dict(QE='EE'), 'advdiff')
    dt = 4e-3
    desc = dict(problem_class=pc, problem_params=dict(pp), sweeper_class=sc,
                sweeper_params=dict(num_nodes=M, quad_type=quad, QI=QI, **extra),
                level_params=dict(dt=dt, restol=-1.0), step_params=dict(maxiter=3))
    C = controller_nonMPI(1, dict(logger_level=40), desc)
    L = C.MS[0].levels[0]
    sw = L.sweep
    u0 = L.prob.u_exact(0.0)
    u0h = u0.get() + 1e-3 * np.random.default_rng(M).standard_normal(u0.get().shape)
    u0[:] = u0h
    uend, stats = C.run(u0, 0.0, 2 * dt)
    res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]

    coll = O.Coll(sw.coll.nodes, sw.coll.weights, sw.coll.Qmat, sw.QI, getattr(sw, 'QE', None) if extra else None,
                  right_is_node=sw.coll.right_is_node, left_is_node=sw.coll.left_is_node)
    opp = dict(pp)
    opp['freq'] = pp['freq']

    def make_level():
        return O.Level(make_oracle_problem(oname, opp), coll, dt, restol=-1.0)

    ref, ostats = O.run_sdc(make_level, u0h, 0.0, 2 * dt, maxiter=3, do_coll_update=bool(sw.params.do_coll_update))
    assert [n for _, n in ostats['niter']] == [v for _, v in get_sorted(stats, type='niter', sortby='time')] == [3, 3]
    assert rel_err(uend.get(), ref) < TOL
    ores = [r for _, hist in ostats['residuals'] for r in hist]
    np.testing.assert_allclose(res, ores, rtol=1e-6, atol=1e-11)


FOREIGN = ['config1', 'mssdc_P2_jac', 'mssdc_P4_gs', 'fixedK_3d', 'fixedK_3d_P2', 'forced2d_run', 'forced2d_run_P2']


@pytest.mark.parametrize('name', FOREIGN)
def test_foreign_level_runs_vs_golden(name):
    """the product sweeper / problem classes inside a Level that is NOT pysdc_amd.level.Level (tests/_plain_level.py:
    plain lists replaced by reset_level, frozen attribute set - the semantics of the reference's core/level.py): the
    sweeper keeps a ForeignLevelState, adopts the lists block after block, and the fused kernels run as usual."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.hip_mesh import hip_mesh
    from pysdc_amd.level import SlabList, ForeignLevelState
    from pysdc_amd.stats import get_sorted
    from tests._plain_level import PlainStep, PlainLevel

    case = load_cases('runs.npz')[name]
    meta = case['meta']
    desc = description_from(meta)
    desc['step_class'] = PlainStep
    C = controller_nonMPI(meta['num_procs'], dict(logger_level=40, **meta['controller_params']), desc)
    Lv = C.MS[0].levels[0]
    assert type(Lv) is PlainLevel and type(Lv.u) is list
    u0 = Lv.prob.u_init
    u0[:] = case['u0']
    uend, stats = C.run(u0, meta['t0'], meta['Tend'])
    assert isinstance(Lv.sweep._dev(), ForeignLevelState) and isinstance(Lv.u, SlabList) and isinstance(uend, hip_mesh)
    niter = get_sorted(stats, type='niter', sortby='time')
    assert [v for _, v in niter] == list(case['niter'])
    assert rel_err(uend.get(), case['uend']) < TOL
    res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]
    np.testing.assert_allclose(res, case['res'], rtol=1e-6, atol=1e-11 * max(1.0, float(np.max(np.abs(case['u0'])))))


def test_foreign_level_tau_and_uend_semantics():
    """a FAS correction assigned into the host's plain list after reset_level reaches the sweep; L.uend is an owning
    object that survives later sweeps (stock hooks keep references to it)."""
    from pysdc_amd.sweepers import generic_implicit
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.level import Step
    from tests._plain_level import PlainStep

    case = load_cases('sweeps_heat.npz')['heat3d_tau']
    meta = case['meta']
    pp = dict(meta['prob_params'])
    pp['nvars'] = tuple(pp['nvars'])
    desc = dict(problem_class=heatNd_unforced, problem_params=pp, sweeper_class=generic_implicit,
                sweeper_params=dict(meta['sweeper_params']), level_params=dict(dt=meta['dt']), step_params=dict(maxiter=9))
    for step_class in (PlainStep, Step):
        S = step_class(desc)
        Lv = S.levels[0]
        Lv.status.time = meta['t0']
        S.init_step(_field(Lv.prob, case['u0']))
        for m in range(len(case['tau'])):
            Lv.tau[m] = _field(Lv.prob, case['tau'][m])
        Lv.sweep.predict()
        kept = []
        for k in range(1, meta['nsweeps'] + 1):
            Lv.sweep.update_nodes()
            Lv.sweep.compute_residual()
            assert abs(Lv.status.residual - float(case[f'k{k}_res_full_abs'])) <= 1e-8 * float(case[f'k{k}_res_full_abs']) + 1e-11
            assert rel_err(np.stack([np.asarray(x) for x in Lv.u]), case[f'k{k}_u']) < TOL
            Lv.sweep.compute_end_point()
            kept.append(Lv.uend)
            assert rel_err(Lv.uend.get(), case[f'k{k}_uend_0']) < TOL
        if step_class is PlainStep:   # earlier end values were not overwritten by later sweeps
            for k, v in enumerate(kept, 1):
                assert rel_err(v.get(), case[f'k{k}_uend_0']) < TOL
        # a tau changed alone (no U write) must invalidate the cached residual
        Lv.sweep.compute_residual()
        r0 = Lv.status.residual
        Lv.tau[0] += 1.0
        Lv.sweep.compute_residual()
        assert abs(Lv.status.residual - r0) > 0.5


def _field(prob, values):
    x = prob.u_init
    x[:] = values
    return x


def test_consecutive_runs_return_independent_objects():
    """controller.run hands out a fresh end-value object per call (controller_nonMPI.py:148,167): the result of an earlier
    run keeps its value when the controller runs again, with one step per block (advance path) and with two"""
    from pysdc_amd.controller import controller_nonMPI

    case = load_cases('runs.npz')['mssdc_P2_jac']
    meta = case['meta']
    for procs in (1, 2):
        C = controller_nonMPI(procs, dict(logger_level=40), description_from(meta))
        dt = meta['level_params']['dt']
        u0 = C.MS[0].levels[0].prob.u_init
        u0[:] = case['u0']
        u1, _ = C.run(u0, 0.0, 2 * dt)
        snap = u1.get()
        u2, _ = C.run(u1, 2 * dt, 4 * dt)
        assert u1 is not u2 and u1.ptr != u2.ptr
        assert np.array_equal(u1.get(), snap)
        assert not np.array_equal(u2.get(), snap)
        # the level's own end value is still readable after the run and equals what was returned
        assert np.array_equal(C.MS[-1 if procs == 2 else 0].levels[0].uend.get(), u2.get())


def test_consecutive_runs_continue_without_reloading_the_start_value():
    """u, _ = C.run(u, t, t + T) repeated: when u is the untouched object the previous run returned and the engine still
    holds that state, the next run starts like the next block of the old one (no copy in, no forward transform) - same
    numbers as one long run and as runs from a modified / foreign start value, which take the ordinary path."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted
    from tests.test_gpu_plugin import description_from

    case = load_cases('sweeps_big3d.npz')['cg64_heat3d_run_LU']
    meta = dict(case['meta'])
    meta['prob_params'] = {k: v for k, v in meta['prob_params'].items() if k != 'solver_type'}
    dt = meta['level_params']['dt']
    u0h = load_cases('sweeps_big3d.npz')['cg64_heat3d_M5_IE']['u0']

    def fresh():
        C = controller_nonMPI(1, dict(logger_level=40), description_from(meta))
        u = C.MS[0].levels[0].prob.u_init
        u[:] = u0h
        return C, u

    C, u = fresh()
    ref, _ = C.run(u, 0.0, 4 * dt)                                  # one long run
    C, u = fresh()
    eng = C.MS[0].levels[0].engine
    eng.profile_enable(True)
    u, _ = C.run(u, 0.0, 2 * dt)
    assert u._lineage is not None
    fwd_before = eng.profile_read().get('fft_x_fwd[1]', (0, 0))[1]
    u, st = C.run(u, 2 * dt, 4 * dt)                                # continues: no forward transform of the start value
    assert eng.profile_read().get('fft_x_fwd[1]', (0, 0))[1] == fwd_before
    assert np.array_equal(u.get(), ref.get())
    # a start value that was touched takes the ordinary path and gives the same numbers
    C, u = fresh()
    u, _ = C.run(u, 0.0, 2 * dt)
    u *= 1.0                                                        # (any write clears the lineage)
    assert u._lineage is None
    u, _ = C.run(u, 2 * dt, 4 * dt)
    assert rel_err(u.get(), ref.get()) < 1e-13


def test_gpu_timings_hook_and_host_mirror_on_device():
    """hooks/log_timings.py:328-342 GPUTimings on device levels: HIP events recorded on the engine's stream around every
    hook pair - device seconds per run / step / iteration / sweep under the reference's keys with the GPU_ prefix, beside
    the always-on wall-clock timings; and the host mirror the stock file hook asks the problem for
    (processSolutionForOutput, hooks/log_solution.py:207-282) is the end value, bit for bit."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.hooks import GPUTimings
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.stats import get_sorted
    from pysdc_amd.sweepers import generic_implicit
    from pysdc_amd.synth import init_field

    n, dt, steps = 128, 2e-3, 3
    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=5, quad_type='RADAU-RIGHT', QI='IE'),
                level_params=dict(dt=dt, restol=-1.0, nsweeps=1), step_params=dict(maxiter=4))
    C = controller_nonMPI(1, dict(logger_level=40, hook_class=[GPUTimings]), desc)
    P = C.MS[0].levels[0].prob
    u0 = P.u_init
    u0[:] = init_field((n, n, n), 2, 1e-3, 0)
    uend, stats = C.run(u0, 0.0, steps * dt)

    def series(kind):
        return [v for _, v in get_sorted(stats, type=kind, sortby='time')]

    g_run, g_step, g_it, g_sw = (series(f'GPU_timing_{k}') for k in ('run', 'step', 'iteration', 'sweep'))
    w_run, w_step = series('timing_run'), series('timing_step')
    assert len(g_run) == 1 and len(g_step) == steps and len(g_it) == 4 * steps and len(g_sw) == 4 * steps
    assert all(v > 0 for v in g_run + g_step + g_it + g_sw)
    assert sum(g_sw) <= sum(g_it) * 1.001 <= sum(g_step) * 1.002 <= g_run[0] * 1.003      # nested intervals on one stream
    # device time of a step cannot exceed the wall time around it by more than the event resolution
    assert all(g <= w * 1.05 + 2e-4 for g, w in zip(g_step, w_step)) and g_run[0] <= w_run[0] * 1.05 + 2e-4
    assert sum(g_sw) > 0.3 * g_run[0]                                                       # the sweeps are most of a run
    mirror = P.processSolutionForOutput(uend)
    assert mirror.shape == (1, n, n, n) and mirror.dtype == np.float64 and np.array_equal(mirror[0], uend.get())


@pytest.mark.parametrize('bc,order', [('periodic', 2), ('periodic', 4), ('periodic', 8), ('dirichlet-zero', 2),
                                      ('dirichlet-zero', 4), ('dirichlet-zero', 6)])
def test_spatial_order_of_eval_f(bc, order):
    """the reference's accuracy test (tests/test_2d_fd_accuracy.py:9-36): eval_f of heatNd_unforced against the analytic
    Laplacian of its sine mode on 2-D grids of growing size - the error falls with the stencil's order; for dirichlet-zero
    that also holds the rows next to the boundary to it (the reference's shifted one-sided stencils keep the order)."""
    from pysdc_amd.problems import heatNd_unforced

    freq = (2, 2)
    errs, hs = [], []
    for p in (4, 5, 6, 7):
        n = 2**p - (0 if bc == 'periodic' else 1)
        P = heatNd_unforced(nvars=(n, n), nu=1.0, freq=freq, order=order, bc=bc)
        x, y = P.grids
        u = np.sin(np.pi * freq[0] * x) * np.sin(np.pi * freq[1] * y)
        exact = -(np.pi**2) * (freq[0] ** 2 + freq[1] ** 2) * u
        ud = P.u_init
        ud[:] = np.broadcast_to(u, (n, n))
        f = P.eval_f(ud, 0.0).get()
        errs.append(np.max(np.abs(f - exact)) / np.max(np.abs(exact)))
        hs.append(P.dx)
    rates = [np.log(errs[i] / errs[i + 1]) / np.log(hs[i] / hs[i + 1]) for i in range(len(errs) - 1)]
    floor = errs[-1] < 1e-11          # high orders on the finest grids reach round-off: only the earlier rates count
    use = rates[:-1] if floor else rates
    assert all(r > order - 0.35 for r in use), (order, bc, errs, rates)
