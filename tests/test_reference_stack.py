"""Build-container test of the drop-in boundary: the REFERENCE's own ``Step``, ``Level``, ``controller_nonMPI``, hooks and
convergence controllers (imported from /root/reference) run around the PRODUCT's sweeper / problem / datatype classes,
named in the description dict exactly as INTEGRATION.md says.  There is no GPU here, so the device is replaced by
tests/_host_engine.py (host memory + oracle numerics); what is under test is the host-side plumbing: the sweeper
accepting the reference's frozen Level, ForeignLevelState adopting its plain lists, ``L.uend`` / ``L.status`` /
``work_counters`` as the stock controller, hooks and convergence controllers read them.  Results are compared with the
golden runs of the pure reference (tests/golden/runs.npz).  Skipped where /root/reference does not exist (GPU box)."""
import os
import sys

import numpy as np
import pytest

from tests._cases import load_cases, rel_err
from tests._host_engine import host_device

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'pySDC')), reason='reference not present')


@pytest.fixture(scope='module')
def ref():
    sys.dont_write_bytecode = True
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'qmat_shim')
    for p in (REF, shim):
        if p not in sys.path:
            sys.path.insert(0, p)
    import pySDC.core.level as core_level
    from pySDC.helpers.stats_helper import get_sorted
    from pySDC.implementations.controller_classes.controller_nonMPI import controller_nonMPI

    return dict(controller=controller_nonMPI, get_sorted=get_sorted, Level=core_level.Level)


def _description(meta):
    from pysdc_amd import problems as P, sweepers as S

    probs = {'heat_unforced': P.heatNd_unforced, 'heat_forced': P.heatNd_forced, 'advdiff': P.advectiondiffusionNd_imex}
    sweeps = {'generic_implicit': S.generic_implicit, 'imex_1st_order': S.imex_1st_order}
    pp = dict(meta['prob_params'])
    if isinstance(pp.get('nvars'), list):
        pp['nvars'] = tuple(pp['nvars'])
    return dict(problem_class=probs[meta['prob']], problem_params=pp, sweeper_class=sweeps[meta['sweeper']],
                sweeper_params=dict(meta['sweeper_params']), level_params=dict(meta['level_params']),
                step_params=dict(maxiter=meta['maxiter']))


@pytest.mark.parametrize('name', ['config1', 'mssdc_P2_jac', 'mssdc_P4_gs', 'fixedK_3d', 'forced2d_run', 'forced2d_run_P2'])
def test_reference_controller_drives_product_classes(ref, name):
    case = load_cases('runs.npz')[name]
    meta = case['meta']
    with host_device():
        from pysdc_amd.hip_mesh import hip_mesh
        from pysdc_amd.level import SlabList

        C = ref['controller'](meta['num_procs'], dict(logger_level=40, **meta['controller_params']), _description(meta))
        L = C.MS[0].levels[0]
        assert type(L) is ref['Level']                      # the reference's frozen Level, not pysdc_amd.level.Level
        assert type(L.u) is list                            # ... with its plain lists, until the sweeper adopts them
        u0 = L.prob.u_init
        u0[:] = case['u0']
        uend, stats = C.run(u0, meta['t0'], meta['Tend'])
        assert isinstance(uend, hip_mesh) and isinstance(L.u, SlabList)
        eng = L.sweep._dev().engine
        assert {'predict', 'sweep', 'residual', 'end_point'} <= set(eng.calls)   # the fused entry points were used
        niter = ref['get_sorted'](stats, type='niter', sortby='time')
        assert [v for _, v in niter] == list(case['niter'])
        assert rel_err(uend.get(), case['uend']) < 1e-10
        res = [v for _, v in ref['get_sorted'](stats, type='residual_post_iteration', sortby='time')]
        np.testing.assert_allclose(res, case['res'], rtol=1e-6, atol=1e-11)
        # stock hooks did run on the device datatype: timings and the restart log are in the statistics
        types = {k.type for k in stats}
        assert {'timing_run', 'timing_step', 'timing_sweep', 'restart'} <= types, types


def test_reference_hooks_and_convergence_controllers_on_device_datatype(ref):
    """stock LogSolution / LogWork hooks and a stock optional convergence controller (fixed-iteration override
    through CheckConvergence parameters is the always-on one; here: EstimateEmbeddedError-free, the cheap
    'CheckIterationEstimator'-free set) see the product's datatype through the reference's own interfaces."""
    from pySDC.implementations.hooks.log_solution import LogSolution
    from pySDC.implementations.hooks.log_work import LogWork

    case = load_cases('runs.npz')['mssdc_P2_jac']
    meta = case['meta']
    with host_device():
        desc = _description(meta)
        C = ref['controller'](2, dict(logger_level=40, hook_class=[LogSolution, LogWork], mssdc_jac=True), desc)
        u0 = C.MS[0].levels[0].prob.u_init
        u0[:] = case['u0']
        uend, stats = C.run(u0, meta['t0'], meta['Tend'])
        sol = ref['get_sorted'](stats, type='u', sortby='time')
        assert len(sol) == len(case['niter'])
        # every logged solution is an OWNING object that kept its value (the last one equals the run's end value,
        # earlier ones differ from it: they were not overwritten by later steps)
        assert rel_err(sol[-1][1].get(), case['uend']) < 1e-10
        assert all(rel_err(s.get(), case['uend']) > 1e-6 for _, s in sol[:-1])
        assert [v for _, v in ref['get_sorted'](stats, type='niter', sortby='time')] == list(case['niter'])


def test_reference_error_hooks_and_a_hook_that_indexes_the_solution(ref):
    """the stock error-logging hooks (hooks/log_errors.py: abs(u_exact - L.uend), L.u[0] * 1.0) and a user hook that treats the
    datatype as the ndarray the reference's `mesh` is (datatype_classes/mesh.py:12-60): integer and slice indexing of L.uend and
    L.u[m], writing through an index - against the golden run"""
    from pySDC.core.hooks import Hooks
    from pySDC.implementations.hooks.log_errors import LogGlobalErrorPostStep, LogLocalErrorPostStep

    case = load_cases('runs.npz')['mssdc_P2_jac']
    meta = case['meta']
    seen = []

    class Probe(Hooks):
        def post_step(self, step, level_number):
            super().post_step(step, level_number)
            L = step.levels[level_number]
            L.sweep.compute_end_point()
            u = L.uend
            host = u.get()
            mid = u.shape[0] // 2
            pt = (3,) * host.ndim
            assert u[pt] == host[pt] and isinstance(u[pt], float)          # integers on every axis: a number
            assert np.array_equal(u[-1].get() if host.ndim > 1 else u[-1], host[-1])
            assert np.array_equal(u[1:-1].get(), host[1:-1]) and u[1:-1].shape == host[1:-1].shape
            assert np.array_equal(u[::-2].get(), host[::-2]) and np.array_equal(u[..., :mid].get(), host[..., :mid])
            assert len(u) == len(host) and np.array_equal(np.asarray(L.u[1])[:4], L.u[1][:4].get())
            w = u * 1.0
            w[2:5] = 7.0
            w[0] = -1.0
            ref_w = host.copy()
            ref_w[2:5], ref_w[0] = 7.0, -1.0
            assert np.array_equal(w.get(), ref_w) and np.array_equal(u.get(), host)      # (the copy changed, the level did not)
            seen.append(float(u[(mid,) * host.ndim]))

    with host_device():
        C = ref['controller'](2, dict(logger_level=40, hook_class=[LogGlobalErrorPostStep, LogLocalErrorPostStep, Probe],
                                      mssdc_jac=True), _description(meta))
        P = C.MS[0].levels[0].prob
        u0 = P.u_init
        u0[:] = case['u0']
        uend, stats = C.run(u0, meta['t0'], meta['Tend'])
        assert rel_err(uend.get(), case['uend']) < 1e-10
        eg = ref['get_sorted'](stats, type='e_global_post_step', sortby='time')
        el = ref['get_sorted'](stats, type='e_local_post_step', sortby='time')
        assert len(eg) == len(case['niter']) == len(el) == len(seen)
        t_end, e_end = eg[-1]
        want = float(np.max(np.abs(P.u_exact(t_end).get() - case['uend'])))
        assert abs(e_end - want) <= 1e-10 * max(want, 1e-30) + 1e-14
        assert all(np.isfinite(v) for _, v in el)


@pytest.mark.parametrize('name', ['mlsdc_heat2d', 'pfasst_heat2d_P2', 'pfasst_heat2d_P4', 'mlsdc_forced2d', 'pfasst_forced2d_P2',
                                  'mlsdc_heat2d_M53'])
def test_reference_multilevel_stack_drives_product_classes(ref, name):
    """two levels inside the REFERENCE's Step: its BaseTransfer (FAS restriction / prolongation on datatype operations,
    core/base_transfer.py:93-251) and its MLSDC / PFASST stages around the product's sweepers, problems, datatype and
    space transfer class (mesh_to_mesh), against the golden runs of the pure reference."""
    from pysdc_amd.transfer import mesh_to_mesh

    case = load_cases('runs_ml.npz')[name]
    meta = case['meta']
    with host_device():
        desc = _description(meta)
        desc['problem_params'] = dict(desc['problem_params'], nvars=[tuple(v) for v in meta['prob_params']['nvars']])
        desc['space_transfer_class'] = mesh_to_mesh
        desc['space_transfer_params'] = dict(iorder=meta['iorder'], rorder=meta['rorder'], periodic=True)
        C = ref['controller'](meta['num_procs'], dict(logger_level=40, **meta['controller_params']), desc)
        assert len(C.MS[0].levels) == 2 and type(C.MS[0].levels[1]) is ref['Level']
        assert type(C.MS[0].base_transfer).__module__.startswith('pySDC.')      # the reference's FAS code, not ours
        u0 = C.MS[0].levels[0].prob.u_init
        u0[:] = case['u0']
        uend, stats = C.run(u0, meta['t0'], meta['Tend'])
        niter = ref['get_sorted'](stats, type='niter', sortby='time')
        assert [v for _, v in niter] == list(case['niter'])
        assert rel_err(uend.get(), case['uend']) < 1e-10
        res = [v for _, v in ref['get_sorted'](stats, type='residual_post_iteration', sortby='time')]
        np.testing.assert_allclose(res, case['res'], rtol=1e-6, atol=1e-11)


def test_reference_file_hook_mirrors_device_solutions_to_disk(ref, tmp_path):
    """the stock LogToFile hook (hooks/log_solution.py:207-282) around the product classes: it asks the problem for the
    output file (pySDC's own FieldsIO format) and for the host form of L.u[0] / L.uend (processSolutionForOutput: the
    device-to-host mirror) - the file then holds the initial value and every step's end value of the golden run"""
    from pySDC.helpers.fieldsIO import FieldsIO
    from pySDC.implementations.hooks.log_solution import LogSolution, LogToFile

    case = load_cases('runs.npz')['mssdc_P2_jac']
    meta = case['meta']

    class ToFile(LogToFile):
        filename = str(tmp_path / 'run.pySDC')
        time_increment = 0

    with host_device():
        C = ref['controller'](1, dict(logger_level=40, hook_class=[ToFile, LogSolution]), _description(meta))
        u0 = C.MS[0].levels[0].prob.u_init
        u0[:] = case['u0']
        uend, stats = C.run(u0, meta['t0'], meta['Tend'])
        logged = ref['get_sorted'](stats, type='u', sortby='time')
        f = FieldsIO.fromFile(ToFile.filename)
        assert len(f.times) == len(logged) + 1                      # initial value + one record per step
        t0, first = f.readField(0)
        assert t0 == meta['t0'] and np.array_equal(first[0], case['u0'])
        for idx, (t, sol) in enumerate(logged, start=1):
            tf, field = f.readField(idx)
            assert abs(tf - t) < 1e-12 and np.array_equal(field[0], sol.get())
        assert np.array_equal(f.readField(len(f.times) - 1)[1][0], uend.get())
