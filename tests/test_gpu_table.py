"""Long runs of sweeps in Fourier space (a step iterated to a residual tolerance): from the 8th sweep of a step on the
real node multipliers of a mode pair come from a table that every launch advances by one sweep (sdc_set_multiplier_table,
k_spec_z<N,M,5,0>) instead of being recomputed by replaying all earlier sweeps.  Same arithmetic in the same order: the
residuals of every sweep, the end value and the node values must agree with the replaying launches and with sweeps that
store every iterate (generic_implicit.py:51-103 applied again and again), which the goldens pin to the reference
(tests/test_gpu_pin1024.py runs the table against the reference's 1024^2 sweeps directly)."""
import numpy as np
import pytest

from pysdc_amd import lib as L
from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS
from tests import _gpu as G

pytestmark = pytest.mark.gpu


def _engine(nvars, M, QI='IE'):
    e = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1, order=2), M)
    coll = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    qi = np.zeros_like(coll.Qmat)
    qi[1:, 1:] = QDELTA_GENERATORS[QI](qGen=coll.generator, tLeft=0).genCoeffs()
    e.set_coeffs(coll.Qmat, qi, None, coll.nodes, coll.weights)
    return e


def _run(nvars, M, nsweeps, steps, configure, dt, deferred=True, QI='IE'):
    e = _engine(nvars, M, QI)
    configure(e)
    if not deferred:
        e.set_deferred(False)
    rng = np.random.default_rng(5)
    u0 = G.profile_for(nvars, 2) + 1e-3 * rng.standard_normal(nvars)
    e.upload(L.SLOT_U, 0, u0)
    e.profile_enable(True)
    res, ends = [], []
    t = 0.0
    for _ in range(steps):
        e.predict(t, dt, 'spread')
        for _k in range(nsweeps):
            e.sweep(t, dt)
            res.append(e.residual(dt, 'full_abs')[0])
        e.end_point(dt, False)
        ends.append(e.download(L.SLOT_UEND))
        nodes = e.download_u()
        e.advance()
        t += dt
    prof = e.profile_read()
    e.close()
    return np.array(res), ends, nodes, {k.split('[')[0] for k in prof}


@pytest.mark.parametrize('nvars,M', [((1024, 1024), 5), ((512, 512), 3), ((512, 512, 512), 2)])
def test_table_matches_replay_and_stored_iterates(nvars, M):
    n = nvars[0]
    dt = 315.0 / (0.1 * 4 * len(nvars) * n * n)   # the stiffness of the headline workload
    K = 24
    tab = _run(nvars, M, K, 2, lambda e: None, dt)                                   # default: table from sweep 8
    rep = _run(nvars, M, K, 2, lambda e: (e.set_multiplier_table(0), e.set_virtual_sweeps(64)), dt)
    sto = _run(nvars, M, K, 2, lambda e: e.set_virtual_sweeps(0), dt)
    assert 'spec_z_res_tab' in tab[3] and 'spec_z_res' not in tab[3], tab[3]
    assert 'spec_z_res_tab' not in rep[3] and 'spec_z_res' not in rep[3], rep[3]
    assert 'spec_z_res' in sto[3] and 'spec_z_res_v0' not in sto[3], sto[3]
    for other, tol in ((rep, 1e-13), (sto, 1e-9)):
        # (residuals fall to ~1e-13 of the solution: against stored iterates they agree to rounding of the iterate)
        np.testing.assert_allclose(tab[0], other[0], rtol=tol, atol=1e-13 * np.max(np.abs(tab[1][0])))
        for a, b in zip(tab[1], other[1]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-13 * np.max(np.abs(b)))
        np.testing.assert_allclose(tab[2], other[2], rtol=0, atol=1e-13 * np.max(np.abs(other[2])))


def test_table_feeds_sweeps_that_store_node_values():
    """eager node fields: the launch hands the iterate itself to the transform (spec_z_tab)"""
    nvars, M, K = (512, 512), 3, 14
    dt = 315.0 / (0.1 * 8 * 512 * 512)
    tab = _run(nvars, M, K, 1, lambda e: None, dt, deferred=False)
    sto = _run(nvars, M, K, 1, lambda e: e.set_virtual_sweeps(0), dt, deferred=False)
    assert 'spec_z_tab' in tab[3], tab[3]
    np.testing.assert_allclose(tab[0], sto[0], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(tab[2], sto[2], rtol=0, atol=1e-13 * np.max(np.abs(sto[2])))


def test_table_is_dropped_when_the_step_size_changes():
    """the multipliers belong to dt * QDelta: another dt fills the table again (gmode 1 at the first sweep that uses the table)"""
    nvars, M = (512, 512), 3
    e = _engine(nvars, M)
    f = _engine(nvars, M)
    f.set_virtual_sweeps(0)
    u0 = G.profile_for(nvars, 2) + 1e-3 * np.random.default_rng(1).standard_normal(nvars)
    for eng in (e, f):
        eng.upload(L.SLOT_U, 0, u0)
    t = 0.0
    for dt in (2e-4, 3e-4, 3e-4):
        for eng in (e, f):
            eng.predict(t, dt, 'spread')
            for _ in range(13):
                eng.sweep(t, dt)
        ra, rb = e.residual(dt, 'full_abs')[0], f.residual(dt, 'full_abs')[0]
        assert abs(ra - rb) <= 1e-9 * abs(rb) + 1e-15
        for eng in (e, f):
            eng.end_point(dt, False)
        ua, ub = e.download(L.SLOT_UEND), f.download(L.SLOT_UEND)
        np.testing.assert_allclose(ua, ub, rtol=0, atol=1e-13 * np.max(np.abs(ub)))
        for eng in (e, f):
            eng.advance()
        t += dt
    e.close()
    f.close()


def test_last_sweep_of_a_step_writes_the_end_spectrum_itself():
    """steps of equal length: from the second step on the launch of the sweep that the step before ended with also writes
    the last node's spectrum (end value / next start value) - no spec_store_last launch any more, same values as the engine
    that stores every iterate"""
    nvars, M, K, steps = (512, 512), 3, 4, 4
    dt = 315.0 / (0.1 * 8 * 512 * 512)
    e = _engine(nvars, M)
    f = _engine(nvars, M)
    f.set_virtual_sweeps(0)
    u0 = G.profile_for(nvars, 2) + 1e-3 * np.random.default_rng(3).standard_normal(nvars)
    for eng in (e, f):
        eng.upload(L.SLOT_U, 0, u0)
    e.profile_enable(True)
    t = 0.0
    for step in range(steps):
        for eng in (e, f):
            eng.predict(t, dt, 'spread')
            for _ in range(K if step != 2 else K + 1):    # (one step takes a sweep more: the guess is wrong there)
                eng.sweep(t, dt)
        ra, rb = e.residual(dt, 'full_abs')[0], f.residual(dt, 'full_abs')[0]
        assert abs(ra - rb) <= 1e-9 * abs(rb) + 1e-15
        for eng in (e, f):
            eng.end_point(dt, False)
        if step % 2:                                           # (the end value is looked at, or only handed over)
            ua, ub = e.download(L.SLOT_UEND), f.download(L.SLOT_UEND)
            np.testing.assert_allclose(ua, ub, rtol=0, atol=1e-13 * np.max(np.abs(ub)))
        for eng in (e, f):
            eng.advance()
        t += dt
    np.testing.assert_allclose(e.download(L.SLOT_U, 0), f.download(L.SLOT_U, 0), rtol=0, atol=1e-13 * np.max(np.abs(u0)))
    prof = {k.split('[')[0]: v for k, v in e.profile_read().items()}
    # step 0: no guess (spec_store_last); step 1: guessed 4, right; step 2: guessed 4, took 5 (the fourth launch wrote a
    # spectrum that the fifth made stale: spec_store_last again); step 3: guessed 5, took 4 (spec_store_last)
    assert prof['spec_z_res_last'][1] == 2 and prof['spec_store_last'][1] == 3, prof
    e.close()
    f.close()


def test_residual_of_the_predictors_state_is_evaluated_when_it_is_read():
    """steps that hand their end value over in Fourier space: max|f(u0)| behind the residual of the spread state would cost a
    norm-only transform of one field per step.  With restol < 0 no convergence test can depend on it: compute_residual
    leaves a thunk in L.status.residual (pysdc_amd.level.LevelStatus), the transform runs if somebody reads the attribute
    - and then delivers the value the engine computes at once without the switch."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit
    import pysdc_amd.level as LV

    n, dt = 64, 1e-3
    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=2, order=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='IE'),
                level_params=dict(dt=dt, restol=-1.0, nsweeps=1), step_params=dict(maxiter=3))
    seen = {}
    saved = LV.LAZY_PREDICTOR_RESIDUAL
    try:
        for lazy in (True, False):
            LV.LAZY_PREDICTOR_RESIDUAL = lazy
            C = controller_nonMPI(1, dict(logger_level=40), desc)
            Lv = C.MS[0].levels[0]
            u0 = Lv.prob.u_exact(0.0)
            C.run(u0, 0.0, 2 * dt)                      # two steps: the third starts from a spectrum
            eng = Lv.engine
            Lv.status.time = 2 * dt
            eng.profile_enable(True)
            Lv.sweep.predict()                          # (without the switch the engine computes the norm here)
            Lv.sweep.compute_residual(stage='IT_CHECK')
            assert Lv.status.residual_is_deferred() == lazy
            names = {k.split('[')[0] for k in eng.profile_read()}
            assert ('fft_z_sym' in names) == (not lazy), names
            first = Lv.status.residual                   # somebody reads it
            assert isinstance(first, float) and first > 0 and not Lv.status.residual_is_deferred()
            assert Lv.status.residual == first
            names = {k.split('[')[0] for k in eng.profile_read()}
            assert 'fft_z_sym' in names, names
            Lv.sweep.update_nodes()
            Lv.sweep.compute_residual(stage='IT_FINE')
            seen[lazy] = (first, Lv.status.residual)
            # a put-off residual nobody asked for is dropped by the sweep that replaces its state
            Lv.sweep.predict()
            Lv.sweep.compute_residual(stage='IT_CHECK')
            Lv.sweep.update_nodes()
            assert not Lv.status.residual_is_deferred()
    finally:
        LV.LAZY_PREDICTOR_RESIDUAL = saved
    assert seen[True] == seen[False], seen
