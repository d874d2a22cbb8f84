"""Residuals the host does not wait for (include/sdcmi.h: sdc_residual_post / sdc_residual_wait; SURVEY 8f rank 1 - the
residual AND the convergence test of check_convergence.py:72-75 on the device): the record the device leaves in pinned host
memory holds the number sdc_residual returns, bit for bit, for every residual type and every state the engine answers for;
runs to a tolerance stop at the reference's iteration; runs with a fixed number of sweeps never wait and still log numbers."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _desc(n, M, restol, maxiter, ndim=3, dt=1e-3):
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    return dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n,) * ndim, nu=0.1, freq=2, order=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='IE'),
                level_params=dict(dt=dt, restol=restol, nsweeps=1), step_params=dict(maxiter=maxiter))


@pytest.mark.parametrize('rt', ['full_abs', 'last_abs', 'full_rel', 'last_rel'])
def test_posted_record_equals_the_blocking_call(rt):
    from pysdc_amd.controller import controller_nonMPI

    Cn = controller_nonMPI(1, dict(logger_level=40), _desc(32, 3, -1.0, 2))
    L = Cn.MS[0].levels[0]
    e = L.engine
    u0 = L.prob.u_exact(0.0)
    L.u[0] = u0
    L.status.time = 0.0
    L.sweep.predict()
    for state in ('predicted', 'swept', 'swept again', 'u0 replaced'):
        if state.startswith('swept'):
            L.sweep.update_nodes()
        if state == 'u0 replaced':
            L.u[0] = 0.5 * L.prob.u_exact(0.0)
        tol = 1e-4
        fut = e.residual_post(1e-3, rt, restol=tol)
        ref, norms = e.residual(1e-3, rt)
        assert fut.result() == ref and np.array_equal(fut.norms, norms), (state, fut.result(), ref)
        assert fut.converged == (ref <= tol)
        assert float(fut) == ref and fut() == ref and fut.done()


def test_tickets_older_than_the_ring_are_refused_and_python_collects_in_time():
    from pysdc_amd import lib as Lb
    from tests._gpu import engine_for

    e = engine_for('heat_unforced', dict(nvars=(16, 16, 16), nu=0.1), 3)
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    c = CollBase(3, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    QI = np.zeros_like(c.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
    e.set_coeffs(c.Qmat, QI, None, c.nodes, c.weights)
    e.upload(Lb.SLOT_U, 0, np.random.default_rng(0).standard_normal(e.nvars))
    e.predict(0.0, 1e-3, 'spread', 0.0, 0.0)
    e.set_unlocked(True)
    futs = []
    for k in range(600):                      # far more than the 256 records the library keeps
        e.sweep(0.0, 1e-3)
        futs.append(e.residual_post(1e-3, 'full_abs'))
    vals = [f.result() for f in futs]         # the early ones were collected by residual_post itself before their slots went
    assert all(np.isfinite(v) and v >= 0 for v in vals) and vals[0] > vals[-1]
    # the raw entry point refuses a ticket whose record is gone
    norms = np.zeros(3)
    res, conv, ready = C.c_double(), C.c_int(), C.c_int()
    rc = e.lib.sdc_residual_wait(e.ctx, 1, 1, norms.ctypes.data_as(C.POINTER(C.c_double)), C.byref(res), C.byref(conv), C.byref(ready))
    assert rc == -3   # SDC_ERR_STATE
    e.close()


def test_fixed_number_of_sweeps_never_waits_and_logs_numbers():
    """restol < 0: the controller's test cannot depend on a residual, the hooks keep what is on its way; return_stats hands
    out floats equal to the ones a run that reads every residual at once records"""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.engine import ResidualFuture
    from pysdc_amd.stats import get_sorted
    import pysdc_amd.sweepers as SW

    collected_during_run = []
    orig = ResidualFuture.result
    running = [False]

    def counting(self):
        if self._value is None and running[0]:
            collected_during_run.append(1)
        return orig(self)

    runs = {}
    saved = SW.QUEUED_RESIDUALS
    try:
        for queued in (True, False):
            SW.QUEUED_RESIDUALS = queued
            Cn = controller_nonMPI(1, dict(logger_level=40), _desc(32, 3, -1.0, 4))
            L = Cn.MS[0].levels[0]
            u0 = L.prob.u_exact(0.0)
            ResidualFuture.result = counting
            try:
                collected_during_run.clear()
                hooks_stats = [h.return_stats for h in Cn.hooks]
                running[0] = True
                for h in Cn.hooks:       # (return_stats at the end of run() is where the numbers are collected)
                    h.return_stats = (lambda f: (lambda: (running.__setitem__(0, False), f())[1]))(h.return_stats)
                uend, stats = Cn.run(u0, 0.0, 5e-3)
            finally:
                ResidualFuture.result = orig
                running[0] = False
            res = get_sorted(stats, type='residual_post_sweep', sortby='time')
            assert len(res) == 20 and all(isinstance(v, float) for _, v in res), res[:3]
            assert all(isinstance(v, float) for v in L.residual)
            runs[queued] = ([v for _, v in res], uend.get(), list(L.residual))
            if queued:
                assert not collected_during_run   # nobody waited for a residual while the run was going
    finally:
        SW.QUEUED_RESIDUALS = saved
    assert runs[False][0] == runs[True][0] and runs[False][2] == runs[True][2]
    assert np.array_equal(runs[False][1], runs[True][1])


def test_runs_to_a_tolerance_stop_where_the_blocking_path_stops():
    """the convergence decision comes from the device's flag (ResidualFuture.converged): it equals the host's comparison for
    every residual of the run, and the run takes the iterations of the run that reads every residual through sdc_residual"""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted
    import pysdc_amd.controller as CT
    import pysdc_amd.sweepers as SW

    seen = []
    orig = CT.check_convergence

    def spy(S):
        out = orig(S)
        r = S.levels[0].status.peek_residual()
        if getattr(r, 'queued', False) and S.status.iter > 0:
            seen.append((r.converged, r.result() <= S.levels[0].params.restol))
        return out

    out = {}
    saved = SW.QUEUED_RESIDUALS
    try:
        for queued in (True, False):
            SW.QUEUED_RESIDUALS = queued
            Cn = controller_nonMPI(1, dict(logger_level=40), _desc(32, 3, 1e-8, 30, dt=1e-2))
            L = Cn.MS[0].levels[0]
            CT.check_convergence = spy
            try:
                uend, stats = Cn.run(L.prob.u_exact(0.0), 0.0, 3e-2)
            finally:
                CT.check_convergence = orig
            out[queued] = ([v for _, v in get_sorted(stats, type='niter', sortby='time')], uend.get(),
                           [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')])
    finally:
        SW.QUEUED_RESIDUALS = saved
    assert seen and all(a == b for a, b in seen), seen
    assert seen[-1][0] is True and any(a is False for a, _ in seen)
    assert out[True][0] == out[False][0] and all(1 < k < 30 for k in out[True][0]), out[True][0]
    assert out[True][2] == out[False][2] and np.array_equal(out[True][1], out[False][1])


def test_blocking_calls_between_posts_do_not_lose_pending_tickets():
    """sdc_residual (blocking) draws tickets from the same 256-record ring as sdc_residual_post: a future that stays pending
    while hundreds of blocking calls go by is collected before its record is overwritten"""
    from pysdc_amd import lib as Lb
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS
    from tests._gpu import engine_for

    e = engine_for('heat_unforced', dict(nvars=(16, 16, 16), nu=0.1), 3)
    c = CollBase(3, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    QI = np.zeros_like(c.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
    e.set_coeffs(c.Qmat, QI, None, c.nodes, c.weights)
    e.upload(Lb.SLOT_U, 0, np.random.default_rng(1).standard_normal(e.nvars))
    e.predict(0.0, 1e-3, 'spread', 0.0, 0.0)
    e.set_unlocked(True)
    e.sweep(0.0, 1e-3)
    first = e.residual_post(1e-3, 'full_abs')
    want, _ = e.residual(1e-3, 'full_abs')
    for _ in range(400):
        e.residual(1e-3, 'last_abs')
    assert first.result() == want
    e.close()
