"""CPU: the product's OWN controller, Level / Step containers, sweepers, hooks and the queued-residual plumbing
(pysdc_amd.engine.ResidualFuture held by LevelStatus, the controller's convergence test reading its `converged` flag, hooks
that log what is on its way) around the host stand-in for the HIP library (tests/_host_engine.py: NumPy slabs, oracle
numerics) against golden runs of the reference - BASELINE config 1 (heat 1-D N = 1024, M = 3, controller_nonMPI) among them.
What runs on the GPU box through the real library (tests/test_gpu_plugin.py::test_run_vs_golden) runs here through the same
host code."""
import numpy as np
import pytest

from tests._cases import load_cases, rel_err
from tests._host_engine import host_device

RUNS = ['config1', 'mssdc_P2_jac', 'mssdc_P4_gs', 'forced2d_run_P2']


def _description(meta):
    from pysdc_amd.problems import heatNd_unforced, heatNd_forced, advectionNd
    from pysdc_amd.sweepers import generic_implicit, imex_1st_order

    probs = {'heat_unforced': heatNd_unforced, 'heat_forced': heatNd_forced, 'advection': advectionNd}
    sweeps = {'generic_implicit': generic_implicit, 'imex_1st_order': imex_1st_order}
    pp = {k: tuple(v) if isinstance(v, list) else v for k, v in meta['prob_params'].items()}
    return dict(problem_class=probs[meta['prob']], problem_params=pp, sweeper_class=sweeps[meta['sweeper']],
                sweeper_params=dict(meta['sweeper_params']), level_params=dict(meta['level_params']),
                step_params=dict(maxiter=meta['maxiter']))


@pytest.mark.parametrize('name', RUNS)
@pytest.mark.parametrize('queued', [True, False])
def test_own_controller_reproduces_golden_runs_on_the_host_engine(name, queued):
    import pysdc_amd.sweepers as SW

    case = load_cases('runs.npz')[name]
    meta = case['meta']
    saved = SW.QUEUED_RESIDUALS
    SW.QUEUED_RESIDUALS = queued
    try:
        with host_device():
            from pysdc_amd.controller import controller_nonMPI
            from pysdc_amd.engine import ResidualFuture
            from pysdc_amd.stats import get_sorted

            C = controller_nonMPI(meta['num_procs'], dict(logger_level=40, **meta['controller_params']), _description(meta))
            L0 = C.MS[0].levels[0]
            u0 = L0.prob.u_init
            u0[:] = case['u0']
            uend, stats = C.run(u0, meta['t0'], meta['Tend'])
            niter = get_sorted(stats, type='niter', sortby='time')
            assert [v for _, v in niter] == list(case['niter'])
            assert rel_err(uend.get(), case['uend']) < 1e-11
            res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]
            assert all(isinstance(v, float) for v in res)          # (what was logged while on its way is a number now)
            np.testing.assert_allclose(res, case['res'], rtol=1e-6, atol=1e-11 * max(1.0, float(np.max(np.abs(case['u0'])))))
            held = L0.status._residual
            assert isinstance(held, ResidualFuture) == queued or isinstance(held, float)
            assert isinstance(L0.status.residual, float) and all(isinstance(v, float) for v in L0.residual)
    finally:
        SW.QUEUED_RESIDUALS = saved


def test_level_status_with_a_residual_that_is_on_its_way():
    from pysdc_amd.engine import ResidualFuture
    from pysdc_amd.level import LevelStatus

    calls = []

    def fetch(block):
        calls.append(block)
        return (0.25, np.array([0.1, 0.25]), True) if (block or len(calls) > 2) else None

    st = LevelStatus()
    fut = ResidualFuture(fetch)
    st.residual = fut
    assert not st.residual_is_deferred()          # queued, not put off: nothing has to happen before the state changes
    assert st.peek_residual() is fut and not fut.done() and calls == [False]
    st.drop_deferred_residual()
    assert st.peek_residual() is fut              # (only thunks are dropped)
    assert st.residual == 0.25 and calls[-1] is True
    assert st.peek_residual() == 0.25 and fut.converged is True and list(fut.norms) == [0.1, 0.25] and float(fut) == 0.25
    ready = ResidualFuture.ready(3.0, [1.0, 3.0], restol=2.0)
    assert ready.done() and ready.result() == 3.0 and ready.converged is False
    assert ResidualFuture.ready(1.0, [1.0], restol=2.0).converged is True and ResidualFuture.ready(1.0, [1.0]).converged is False
    # a thunk (work not queued yet) is deferred, evaluated on the first read and replaced by its number
    st.residual = lambda: 7.0
    assert st.residual_is_deferred() and st.residual == 7.0 and not st.residual_is_deferred()


def test_a_step_that_asks_for_a_restart_is_refused_loudly():
    """S.status.restart set by a hook (what a convergence controller of the reference or a subclassed sweeper would do,
    controller_nonMPI.py:150-163): the package's own controllers have no restart logic and say so"""
    from pysdc_amd.errors import ControllerError
    from pysdc_amd.hooks import Hooks

    class AskForRestart(Hooks):
        def post_step(self, step, level_number):
            step.status.restart = True

    case = load_cases('runs.npz')['config1']
    meta = case['meta']
    with host_device():
        from pysdc_amd.controller import controller_nonMPI

        C = controller_nonMPI(1, dict(logger_level=40, hook_class=[AskForRestart]), _description(meta))
        u0 = C.MS[0].levels[0].prob.u_init
        u0[:] = case['u0']
        with pytest.raises(ControllerError, match='restart'):
            C.run(u0, meta['t0'], meta['Tend'])
