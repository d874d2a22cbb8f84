"""GPU parity at BASELINE sizes (512^3, 1024^3), where the oracle cannot run, through size-independent
properties:
  * eigenmode sweep: for u0 = an eigenvector of the periodic Laplacian every node value is a scalar multiple
    of u0, the scalars being those of the reference's sweep on the scalar test equation (the property
    tests/test_sweepers/test_imexsweeper.py:110-128 checks in matrix form);
  * solve round trip: (I - factor*A) applied to the result of solve_system returns the right-hand side;
  * the two sweep data flows (spectral reuse / gather on F) agree on a noisy field.
All comparisons run on the device (axpby + max norm), no multi-GB host copies."""
import ctypes as C

import numpy as np
import pytest

from pysdc_amd import lib as L
from tests import _gpu as G
from tests._cases import rel_err

pytestmark = pytest.mark.gpu


def _free_gb():
    import torch

    return torch.cuda.mem_get_info()[0] / 1e9


def _coeffs(M, QI='IE'):
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    qi = np.zeros_like(c.Qmat)
    qi[1:, 1:] = QDELTA_GENERATORS[QI](qGen=c.generator, tLeft=0).genCoeffs()
    return c, qi


def _scalar_sweeps(lam, dt, c, qi, nsweeps):
    """the reference's generic_implicit sweep on u' = lam*u, u0 = 1, spread predictor."""
    M = c.num_nodes
    u = np.ones(M + 1)
    f = lam * u
    out = []
    for _ in range(nsweeps):
        integral = np.array([sum(dt * c.Qmat[m + 1, j] * f[j] for j in range(1, M + 1)) for m in range(M)])
        for m in range(M):
            for j in range(1, M + 1):
                integral[m] -= dt * qi[m + 1, j] * f[j]
            integral[m] += u[0]
        for m in range(M):
            rhs = integral[m]
            for j in range(1, m + 1):
                rhs += dt * qi[m + 1, j] * f[j]
            u[m + 1] = rhs / (1.0 - dt * qi[m + 1, m + 1] * lam)
            f[m + 1] = lam * u[m + 1]
        out.append(u.copy())
    return out


@pytest.mark.parametrize('n,QI,reuse', [(512, 'IE', True), (512, 'LU', False), (1024, 'IE', True)])
def test_eigenmode_sweep_fullsize(n, QI, reuse):
    need = 8e-9 * n**3 * (24 if reuse else 19)
    if _free_gb() < need + 4:
        pytest.skip(f'needs {need:.0f} GB of HBM')
    M, nu = 5, 0.1
    dt = 1e-3 * (512.0 / n) ** 2
    e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=nu), M)
    c, qi = _coeffs(M, QI)
    e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
    e.set_spectral_reuse(reuse)
    freq = (C.c_int * 3)(2, 2, 2)
    L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.0, 0), e.ctx)
    e.invalidate_spectra(1)
    dx = 1.0 / n
    lam = 3.0 * nu * (2.0 * np.cos(2.0 * np.pi / n) - 2.0) / dx**2      # discrete eigenvalue of sin(2 pi x) per axis
    e.predict(0.0, dt)
    expected = _scalar_sweeps(lam, dt, c, qi, 3)
    u0n = e.vec_amax(e.N, e.ptr(L.SLOT_U, 0))
    assert abs(u0n - 1.0) < 1e-12
    tmp = e.ptr(L.SLOT_UEND)
    for k in range(3):
        e.sweep(0.0, dt)
        for m in range(1, M + 1):
            # U[m] - c_m * U[0]  and  F[m] - lam * c_m * U[0]
            e.vec_axpby(e.N, 1.0, e.ptr(L.SLOT_U, m), -expected[k][m], e.ptr(L.SLOT_U, 0), tmp)
            assert e.vec_amax(e.N, tmp) < 1e-12, (k, m)
            e.vec_axpby(e.N, 1.0, e.ptr(L.SLOT_F, m), -lam * expected[k][m], e.ptr(L.SLOT_U, 0), tmp)
            assert e.vec_amax(e.N, tmp) < 1e-10 * abs(lam), (k, m)
    # collocation residual of the scalar problem
    res, _ = e.residual(dt)
    u = expected[-1]
    ref = max(abs(1.0 + dt * sum(c.Qmat[m, j] * lam * u[j] for j in range(1, M + 1)) - u[m]) for m in range(1, M + 1))
    assert abs(res - ref) < 1e-11
    e.close()


@pytest.mark.parametrize('n', [512, 1024])
def test_solve_roundtrip_fullsize(n):
    if _free_gb() < 8e-9 * n**3 * 10 + 4:
        pytest.skip('not enough HBM')
    nu = 0.1
    e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=nu), 2)
    freq = (C.c_int * 3)(2, 4, 6)
    rhs, sol, back = e.ptr(L.SLOT_U, 0), e.ptr(L.SLOT_U, 1), e.ptr(L.SLOT_U, 2)
    L.check(e.lib.sdc_init_field(e.ctx, rhs, freq, 0.5, 3), e.ctx)          # O(1) noise: all modes excited
    factor = 63.0 / (12.0 * nu * n * n)
    e.solve(rhs, factor, sol)
    e.eval_f(sol, 0.0, e.ptr(L.SLOT_F, 0))                                  # A sol
    e.vec_axpby(e.N, 1.0, sol, -factor, e.ptr(L.SLOT_F, 0), back)           # (I - factor A) sol
    e.vec_axpby(e.N, 1.0, back, -1.0, rhs, back)
    assert e.vec_amax(e.N, back) < 1e-12 * e.vec_amax(e.N, rhs)
    e.close()


def test_reuse_and_gather_paths_agree_fullsize():
    n, M = 512, 5
    if _free_gb() < 8e-9 * n**3 * 45 + 4:
        pytest.skip('not enough HBM')
    dt = 1e-3
    c, qi = _coeffs(M, 'IE')
    engines = []
    for reuse in (True, False):
        e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        e.set_spectral_reuse(reuse)
        freq = (C.c_int * 3)(2, 2, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 1e-3, 0), e.ctx)
        e.invalidate_spectra(1)
        e.predict(0.0, dt)
        for _ in range(3):
            e.sweep(0.0, dt)
        engines.append(e)
    a, b = engines
    tmp = a.ptr(L.SLOT_UEND)
    for m in range(1, M + 1):
        a.vec_axpby(a.N, 1.0, a.ptr(L.SLOT_U, m), -1.0, b.ptr(L.SLOT_U, m), tmp)
        assert a.vec_amax(a.N, tmp) < 1e-13
    ra, rb = a.residual(dt)[0], b.residual(dt)[0]
    assert abs(ra - rb) < 1e-11
    for e in engines:
        e.close()


@pytest.mark.parametrize('n,prob', [(64, 'heat_unforced'), (256, 'heat_unforced'), (64, 'advdiff'), (128, 'advdiff')])
def test_fused_residual_matches_separate_kernels(n, prob):
    """the sweep's fused eval_f + residual kernel against the separate stencil and residual kernels on the
    same noisy state (same U by construction; F and the node norms must agree); implicit and IMEX operators."""
    M, dt = 5, 1e-3 * (512.0 / n) ** 2
    c, qi = _coeffs(M, 'LU')
    qe = None
    if prob == 'advdiff':
        from pysdc_amd.coeffs import QDELTA_GENERATORS

        qe = np.zeros_like(c.Qmat)
        qe[1:, 1:], qe[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    out = []
    for fused in (True, False):
        e = G.engine_for(prob, dict(nvars=(n, n, n), nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, qe, c.nodes, c.weights)
        e.set_fused_residual(fused)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.3, 11), e.ctx)
        e.invalidate_spectra(1)
        e.predict(0.0, dt)
        for _ in range(2):
            e.sweep(0.0, dt)
        res, norms = e.residual(dt)
        out.append((e, res, norms))
    (a, ra, na), (b, rb, nb) = out
    assert np.allclose(na, nb, rtol=1e-9, atol=1e-13) and abs(ra - rb) <= 1e-9 * abs(rb)
    tmp = a.ptr(L.SLOT_UEND)
    for m in range(1, M + 1):
        for comp in range(a.ncomp):
            a.vec_axpby(a.N, 1.0, a.ptr(L.SLOT_F, m, comp), -1.0, b.ptr(L.SLOT_F, m, comp), tmp)
            assert a.vec_amax(a.N, tmp) <= 1e-12 * b.vec_amax(b.N, b.ptr(L.SLOT_F, m, comp))
    # a write through the API invalidates the cached norms
    a.upload(L.SLOT_U, 2, np.zeros((n, n, n)))
    assert abs(a.residual(dt)[1][1] - na[1]) > 1e-6      # node 2's norm is recomputed from the new state
    for e, _, _ in out:
        e.close()


@pytest.mark.parametrize('virtual', [0, 8, 2])
@pytest.mark.parametrize('n,prob', [(64, 'heat_unforced'), (128, 'heat_unforced'), (64, 'advdiff')])
def test_deferred_node_fields_match_eager(n, prob, virtual):
    """sdc_set_deferred: the sweep that does not store F[1..M] / the predictor that does not store the node
    copies must hand out the same bits as the eager engine once the fields are asked for, at every stage.
    virtual > 0 (sdc_set_virtual_sweeps, the default): sweeps that follow a spread predictor do not even store the
    transforms of their iterate - it is recomputed from the transform of u[0] through node multipliers, which rounds
    differently: same values to 1e-13 of the field's size instead of the same bits."""
    M, dt = 5, 1e-3 * (512.0 / n) ** 2
    c, qi = _coeffs(M, 'IE')
    qe = None
    if prob == 'advdiff':
        from pysdc_amd.coeffs import QDELTA_GENERATORS

        qe = np.zeros_like(c.Qmat)
        qe[1:, 1:], qe[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    engines = []
    for deferred in (True, False):
        e = G.engine_for(prob, dict(nvars=(n, n, n), nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, qe, c.nodes, c.weights)
        e.set_deferred(deferred)
        e.set_virtual_sweeps(virtual)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.3, 5), e.ctx)
        e.invalidate_spectra(1)
        engines.append(e)
    a, b = engines

    def eq(x, y):
        return np.array_equal(x, y) if virtual == 0 else rel_err(x, y) < 1e-13

    def same(stage, slots=('u', 'f')):
        if 'u' in slots:
            assert eq(a.download_u(), b.download_u()), stage
        if 'f' in slots:
            assert eq(a.download_f(), b.download_f()), stage

    for e in engines:
        e.predict(0.0, dt)
    ra, rb = a.residual(dt), b.residual(dt)
    assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1])
    for e in engines:
        e.end_point(dt, False)                      # pending spread: uend = u0
    assert np.array_equal(a.download(L.SLOT_UEND), b.download(L.SLOT_UEND))
    for e in engines:
        e.sweep(0.0, dt)                            # first sweep straight from the pending spread
    ra, rb = a.residual(dt), b.residual(dt)         # deferred: norms of the inverse transform of the residual's
    assert np.allclose(ra[1], rb[1], rtol=1e-9, atol=1e-14) and abs(ra[0] - rb[0]) <= 1e-9 * rb[0]  # spectrum
    same('sweep 1', ('f',))                          # F asked for first ...
    same('sweep 1')                                  # ... and again together with U
    for e in engines:
        e.sweep(0.0, dt)
        e.end_point(dt, True)                       # the collocation update integrates F
    assert eq(a.download(L.SLOT_UEND), b.download(L.SLOT_UEND))
    for e in engines:
        e.sweep(0.0, dt)
    for e in engines:
        e.integrate(dt, [e.ptr(L.SLOT_TAU, m) for m in range(M)])
    for m in range(M):
        assert eq(a.download(L.SLOT_TAU, m), b.download(L.SLOT_TAU, m))
    same('sweep 3')
    # u[0] replaced after a sweep (what a receive from the previous time slice does): the residual of the
    # cached iterate against the NEW u[0] is reduced from its transform; the node fields stay what they were
    new_u0 = a.download(L.SLOT_U, 0) * 1.001 + 1e-4
    for e in engines:
        e.sweep(0.0, dt)
        e.upload(L.SLOT_U, 0, new_u0)
    ra, rb = a.residual(dt), b.residual(dt)
    assert np.allclose(ra[1], rb[1], rtol=1e-9, atol=1e-14) and abs(ra[0] - rb[0]) <= 1e-9 * rb[0]
    same('u0 replaced after a sweep')
    for e in engines:
        e.sweep(0.0, dt)                            # ... and the next sweep starts from the new u[0]
    same('sweep after the replaced u0')
    # a predictor whose copies are still pending when U[0] is replaced keeps the OLD u0 at the nodes
    for e in engines:
        e.predict(0.0, dt)
        e.upload(L.SLOT_U, 0, np.full((n, n, n), 0.25))
    same('u0 replaced after predict')
    # switching the mode off stores whatever is pending
    for e in engines:
        e.predict(0.0, dt)
        e.sweep(0.0, dt)
    a.set_deferred(False)
    same('mode switched off')
    for e in engines:
        e.close()


@pytest.mark.parametrize('n,ndim', [(64, 3), (256, 2), (512, 1)])
def test_advance_hands_the_spectrum_over(n, ndim):
    """sdc_advance (u[0] <- uend on the same level, spectrum of the last node reused as the transform of the new
    u[0]) against the plain hand-over through host memory: the next step must agree to round-off."""
    M, dt = 3, 2e-4
    c, qi = _coeffs(M, 'LU')
    engines = []
    for adv in (True, False):
        e = G.engine_for('heat_unforced', dict(nvars=(n,) * ndim, nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.1, 3), e.ctx)
        e.invalidate_spectra(1)
        for step in range(3):
            e.predict(0.0, dt)
            for _ in range(3):
                e.sweep(0.0, dt)
            e.end_point(dt, False)
            if adv:
                e.advance()
            else:
                e.upload(L.SLOT_U, 0, e.download(L.SLOT_UEND))
        e.predict(0.0, dt)
        e.sweep(0.0, dt)
        engines.append(e)
    a, b = engines
    ua, ub = a.download_u(), b.download_u()
    assert np.max(np.abs(ua - ub)) <= 1e-13 * np.max(np.abs(ub))
    ra, rb = a.residual(dt), b.residual(dt)
    assert np.allclose(ra[1], rb[1], rtol=1e-7, atol=1e-14)
    for e in engines:
        e.close()


@pytest.mark.parametrize('n,prob', [(64, 'heat_unforced'), (64, 'advdiff')])
def test_replace_u0_updates_the_residual_from_kept_fields(n, prob):
    """time-parallel flow on one engine: sweep, u[0] replaced (a receive), residual against the new u[0], next
    sweep.  With kept residual fields the norms come out of the replacing pass itself; they must agree with the
    eager engine that recomputes the residual from U and F, and the following sweeps must not notice."""
    M, dt = 5, 1e-3 * (512.0 / n) ** 2
    c, qi = _coeffs(M, 'IE')
    qe = None
    if prob == 'advdiff':
        from pysdc_amd.coeffs import QDELTA_GENERATORS

        qe = np.zeros_like(c.Qmat)
        qe[1:, 1:], qe[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    engines = []
    for keep in (True, False):
        e = G.engine_for(prob, dict(nvars=(n, n, n), nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, qe, c.nodes, c.weights)
        e.set_deferred(keep)
        e.set_keep_residual_fields(keep)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.3, 5), e.ctx)
        e.invalidate_spectra(1)
        e.predict(0.0, dt)
        engines.append(e)
    a, b = engines
    import torch

    for it in range(3):
        for e in engines:
            e.sweep(0.0, dt)
        ra, rb = a.residual(dt), b.residual(dt)
        assert np.allclose(ra[1], rb[1], rtol=1e-9, atol=1e-14)
        new_u0 = b.download(L.SLOT_U, 0) * (1.0 + 1e-3 * (it + 1)) + 1e-5
        t = torch.from_numpy(new_u0.reshape(-1)).cuda()
        for e in engines:
            e.replace_u0(t.data_ptr())
        ra, rb = a.residual(dt), b.residual(dt)                 # a: norms from the replacing pass
        assert np.allclose(ra[1], rb[1], rtol=1e-9, atol=1e-14) and abs(ra[0] - rb[0]) <= 1e-9 * rb[0]
        assert np.array_equal(a.download(L.SLOT_U, 0), new_u0)
    ua, ub = a.download_u(), b.download_u()
    assert np.max(np.abs(ua - ub)) <= 1e-13 * np.max(np.abs(ub))
    for e in engines:
        e.close()


def test_early_end_point_and_stream_wait():
    """sdc_set_early_end_point: the sweep leaves the end value in UEND (same bits as sdc_end_point afterwards) and
    another stream that waits through sdc_stream_wait_uend sees it complete while the engine's stream goes on."""
    import torch
    from pysdc_amd.hip_mesh import _CAI

    n, M, dt = 128, 5, 4e-3
    c, qi = _coeffs(M, 'IE')
    engines = []
    for early in (True, False):
        e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        e.set_early_end_point(early)
        e.set_virtual_sweeps(0)       # (same bits wanted below: both engines store the transforms of every iterate)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.3, 5), e.ctx)
        e.invalidate_spectra(1)
        e.predict(0.0, dt)
        engines.append(e)
    a, b = engines
    side = torch.cuda.Stream()
    got = torch.empty(a.N, dtype=torch.float64, device='cuda')
    uend_a = torch.as_tensor(_CAI(a.lib.sdc_slot_ptr(a.ctx, L.SLOT_UEND, 0, 0), a.N, a), device='cuda')
    for it in range(3):
        for e in engines:
            e.sweep(0.0, dt)
        a.end_point(dt, False)                       # free for a
        a.stream_wait_uend(side.cuda_stream)
        with torch.cuda.stream(side):
            got.copy_(uend_a)
        ra = a.residual(dt)
        b.end_point(dt, False)
        rb = b.residual(dt)
        side.synchronize()
        want = b.download(L.SLOT_UEND)
        assert np.array_equal(got.cpu().numpy().reshape(want.shape), want), it
        assert np.array_equal(ra[1], rb[1])
    for e in engines:
        e.close()


@pytest.mark.parametrize('seed', list(range(12)))
@pytest.mark.parametrize('prob', ['heat_unforced', 'advdiff', 'heat_forced'])
def test_deferred_state_machine_random_walk(prob, seed):
    """random sequences of C-ABI calls on two engines - every deferral switched on (deferred node fields, kept
    residual fields, early end value) against everything eager: whatever is read back must agree."""
    import os
    import torch

    n = int(os.environ.get('PYSDC_FUZZ_N', '64'))             # (scripts/fuzz_more.py also walks other sizes and node counts)
    M = int(os.environ.get('PYSDC_FUZZ_M', '6' if seed >= 10 else '3'))   # above 5 nodes: pointwise spectral kernel + passes
    dt = 0.05 * (64.0 / n) ** 2
    c, qi = _coeffs(M, 'LU')
    qe = None
    if prob in ('advdiff', 'heat_forced'):
        from pysdc_amd.coeffs import QDELTA_GENERATORS

        qe = np.zeros_like(c.Qmat)
        qe[1:, 1:], qe[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    engines = []
    for lazy in (True, False):
        e = G.engine_for(prob, dict(nvars=(n, n, n), nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, qe, c.nodes, c.weights)
        if prob == 'heat_forced':       # time-dependent factor of the forcing at t0 and the node times
            e.set_forcing_values([np.cos(0.3 * k) for k in range(M + 1)])
            e.set_spectral_reuse(lazy)  # eager engine: the general data flow (gather on the F slab)
        e.set_deferred(lazy)
        # a third of the walks each: residual fields kept / end value produced early (time-parallel levels) / neither -
        # only then are iterates recomputed from the transform of u[0] (sdc_set_virtual_sweeps; PYSDC_FUZZ_VIRTUAL: its limit)
        e.set_keep_residual_fields(lazy and seed % 3 == 0)
        e.set_early_end_point(lazy and seed % 3 == 1)
        if os.environ.get('PYSDC_FUZZ_VIRTUAL') is not None:
            e.set_virtual_sweeps(int(os.environ['PYSDC_FUZZ_VIRTUAL']))
        freq = (C.c_int * 3)(2, 2, 4)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.2, 17), e.ctx)
        e.invalidate_spectra(1)
        e.predict(0.0, dt)
        engines.append(e)
    a, b = engines
    rng = np.random.default_rng(100 + seed)

    def close(x, y, what):
        x, y = np.asarray(x), np.asarray(y)
        assert np.max(np.abs(x - y)) <= 1e-12 * max(1.0, np.max(np.abs(y))), ' '.join(what)

    ops = ['sweep'] * 6 + ['residual'] * 3 + ['end_point', 'end_point_coll', 'get_u', 'get_f', 'put_u0', 'put_um',
                                               'replace_u0', 'advance', 'advance', 'get_u0', 'integrate', 'predict', 'predict_copy', 'get_f0',
                                               'toggle_reuse', 'toggle_fused', 'toggle_skip', 'tau_on', 'tau_off']
    state = dict(reuse=True, fused=True, skip=False)
    trace = []
    for step in range(60):
        op = ops[rng.integers(len(ops))]
        trace.append(op)
        m = int(rng.integers(1, M + 1))
        if op == 'sweep':
            for e in engines:
                e.sweep(0.0, dt)
        elif op == 'residual':
            rt = ['full_abs', 'last_abs', 'full_rel', 'last_rel'][rng.integers(4)]
            ra, rb = a.residual(dt, rt), b.residual(dt, rt)
            assert np.allclose(ra[1], rb[1], rtol=1e-7, atol=1e-13), trace
            assert abs(ra[0] - rb[0]) <= 1e-7 * abs(rb[0]) + 1e-13, trace
        elif op in ('end_point', 'end_point_coll'):
            for e in engines:
                e.end_point(dt, op == 'end_point_coll')
            close(a.download(L.SLOT_UEND), b.download(L.SLOT_UEND), trace)
        elif op == 'get_u':
            close(a.download(L.SLOT_U, m), b.download(L.SLOT_U, m), trace)
        elif op == 'get_f':
            for comp in range(a.ncomp):
                close(a.download(L.SLOT_F, m, comp), b.download(L.SLOT_F, m, comp), trace)
        elif op == 'get_f0':
            close(a.download(L.SLOT_F, 0), b.download(L.SLOT_F, 0), trace)
        elif op == 'get_u0':   # (after an advance the start value may still lie in the old end-value buffer)
            close(a.download(L.SLOT_U, 0), b.download(L.SLOT_U, 0), trace)
        elif op in ('put_u0', 'put_um'):
            x = rng.standard_normal((n, n, n)) * 0.1
            for e in engines:
                e.upload(L.SLOT_U, 0 if op == 'put_u0' else m, x)
            if op == 'put_u0':                       # the hosts' f[0] = f(u[0]) after a receive
                for e in engines:
                    L.check(e.lib.sdc_defer_f0(e.ctx), e.ctx)
        elif op == 'replace_u0':
            x = torch.from_numpy(rng.standard_normal(n**3) * 0.1).cuda()
            for e in engines:
                e.replace_u0(x.data_ptr())
                L.check(e.lib.sdc_defer_f0(e.ctx), e.ctx)
        elif op == 'advance':
            for e in engines:
                e.end_point(dt, False)
                e.advance()
                e.predict(0.0, dt)
        elif op == 'integrate':
            for e in engines:
                e.integrate(dt, [e.ptr(L.SLOT_TAU, k) for k in range(M)])
            for k in range(M):
                close(a.download(L.SLOT_TAU, k), b.download(L.SLOT_TAU, k), trace)
        elif op == 'predict':
            for e in engines:
                e.predict(0.0, dt)
        elif op == 'predict_copy':
            for e in engines:
                e.predict(0.0, dt, 'copy')
        elif op == 'toggle_reuse':
            state['reuse'] = not state['reuse']
            for e in engines:
                e.set_spectral_reuse(state['reuse'])
        elif op == 'toggle_fused':
            state['fused'] = not state['fused']
            for e in engines:
                e.set_fused_residual(state['fused'])
        elif op == 'toggle_skip':      # sweeps of the deferring engine stop producing the residual (it may still be asked)
            state['skip'] = not state['skip']
            a.set_skip_residual(state['skip'])
        elif op == 'tau_on':
            taus = [rng.standard_normal((n, n, n)) * 1e-3 for _ in range(M)]
            for e in engines:
                e.set_tau_active(True)
                for k in range(M):
                    e.upload(L.SLOT_TAU, k, taus[k])
        elif op == 'tau_off':
            for e in engines:
                e.set_tau_active(False)
    close(a.download_u(), b.download_u(), trace)
    close(a.download_f(), b.download_f(), trace)
    for e in engines:
        e.close()


@pytest.mark.parametrize('n', [512, 1024])
def test_fourier_space_steps_fullsize(n):
    """BASELINE-size run of the bench's data flow - sweeps that never leave Fourier space, residual norms from the
    residual's spectrum, end value from the last node only, hand-over with sdc_advance - on an eigenmode of the
    discrete operator, where every node value, residual and end value is a known multiple of u0 (scalar SDC)."""
    if _free_gb() < 8e-9 * n**3 * 25 + 4:
        pytest.skip('not enough HBM')
    M, nu = 5, 0.1
    dt = 1e-3 * (512.0 / n) ** 2
    e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=nu), M)
    c, qi = _coeffs(M, 'IE')
    e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
    freq = (C.c_int * 3)(2, 2, 2)
    L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.0, 0), e.ctx)
    e.invalidate_spectra(1)
    dx = 1.0 / n
    lam = 3.0 * nu * (2.0 * np.cos(2.0 * np.pi / n) - 2.0) / dx**2
    K = 4
    scal = _scalar_sweeps(lam, dt, c, qi, K)

    def scalar_residual(u):
        return max(abs(1.0 + dt * sum(c.Qmat[m, j] * lam * u[j] for j in range(1, M + 1)) - u[m]) for m in range(1, M + 1))

    amp = 1.0
    for step in range(2):
        e.predict(0.0, dt)
        res, _ = e.residual(dt)                              # spread predictor: from max|f(u0)|
        assert abs(res - amp * scalar_residual(np.ones(M + 1))) < 1e-11
        for k in range(K):
            e.sweep(0.0, dt)
            res, norms = e.residual(dt)
            assert abs(res - amp * scalar_residual(scal[k])) < 1e-11, (step, k)
        e.end_point(dt, False)
        amp *= scal[-1][M]
        assert abs(e.vec_amax(e.N, e.lib.sdc_slot_ptr(e.ctx, L.SLOT_UEND, 0, 0)) - abs(amp)) < 1e-12
        if step == 0:
            e.advance()
    # the node fields of the last step, brought back to real space on demand: c_m * u0 of that step
    tmp = e.ptr(L.SLOT_UEND)
    for m in (1, M):
        e.vec_axpby(e.N, 1.0, e.ptr(L.SLOT_U, m), -scal[-1][m], e.ptr(L.SLOT_U, 0), tmp)
        assert e.vec_amax(e.N, tmp) < 1e-12
        e.vec_axpby(e.N, 1.0, e.ptr(L.SLOT_F, m), -lam * scal[-1][m], e.ptr(L.SLOT_U, 0), tmp)
        assert e.vec_amax(e.N, tmp) < 1e-10 * abs(lam)
    e.close()


def test_2048_squared_fourier_sweeps_agree_with_general_path():
    """n = 2048 in 2-D (the longest line the transforms take): sweeps that stay in Fourier space against the general
    data flow (gather on F, transform, solve, transform back) on noisy data."""
    n, M, dt = 2048, 3, 1e-5
    c, qi = _coeffs(M, 'LU')
    engines = []
    for reuse in (True, False):
        e = G.engine_for('heat_unforced', dict(nvars=(n, n), nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        e.set_spectral_reuse(reuse)
        freq = (C.c_int * 3)(2, 4, 0)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.1, 9), e.ctx)
        e.invalidate_spectra(1)
        e.predict(0.0, dt)
        for _ in range(3):
            e.sweep(0.0, dt)
        engines.append(e)
    a, b = engines
    ra, rb = a.residual(dt), b.residual(dt)
    assert np.allclose(ra[1], rb[1], rtol=1e-8, atol=1e-13)
    ua, ub = a.download_u(), b.download_u()
    assert np.max(np.abs(ua - ub)) <= 1e-12 * np.max(np.abs(ub))
    for e in engines:
        e.close()


@pytest.mark.parametrize('prob', ['heat_unforced', 'advdiff'])
@pytest.mark.parametrize('nvars', [(512,), (1024,), (512, 512), (1024, 1024), (128, 128, 128)])
def test_fourier_space_sweeps_agree_with_general_path_in_every_dimension(nvars, prob):
    """the line-transform kernels of the sizes the 3-D benchmarks use (512, 1024), driven as 1-D and 2-D problems: an
    engine that defers everything and sweeps in Fourier space against one that stores every field and gathers on the F
    slab - node values, right-hand sides, residuals and end values after three sweeps."""
    M, dt = 5, 2e-4
    c, qi = _coeffs(M, 'LU')
    qe = None
    if prob == 'advdiff':
        from pysdc_amd.coeffs import QDELTA_GENERATORS

        qe = np.zeros_like(c.Qmat)
        qe[1:, 1:], qe[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    u0 = np.random.default_rng(len(nvars)).standard_normal(nvars)
    got = []
    for lazy in (True, False):
        e = G.engine_for(prob, dict(nvars=nvars, nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, qe, c.nodes, c.weights)
        e.set_deferred(lazy)
        e.set_spectral_reuse(lazy)
        e.upload(L.SLOT_U, 0, u0)
        e.predict(0.0, dt)
        res = []
        for _ in range(3):
            e.sweep(0.0, dt)
            res.append(e.residual(dt, 'full_abs')[0])
        e.end_point(dt, False)
        got.append((e.download_u(), e.download_f(), np.array(res), e.download(L.SLOT_UEND)))
        e.close()
    a, b = got
    for x, y in zip(a, b):
        x, y = np.asarray(x), np.asarray(y)
        assert np.max(np.abs(x - y)) <= 1e-11 * max(1.0, float(np.max(np.abs(y)))), nvars


@pytest.mark.parametrize('vmax', [8, 2, 1])
@pytest.mark.parametrize('n,ndim,prob,qd', [(64, 3, 'heat_unforced', 'IE'), (128, 3, 'heat_unforced', 'LU'),
                                           (64, 3, 'advdiff', 'LU'), (256, 2, 'heat_unforced', 'LU'),
                                           (64, 3, 'heat_unforced', 'MIN-SR-S'),
                                           # lines of 512 and 1024: one multiplier evaluation per mode pair (MODE 4)
                                           (512, 3, 'heat_unforced', 'IE'), (1024, 2, 'heat_unforced', 'LU'),
                                           (512, 2, 'heat_unforced', 'MIN-SR-S')])
def test_iterates_recomputed_from_the_start_value(n, ndim, prob, qd, vmax):
    """sdc_set_virtual_sweeps: sweeps after a spread predictor read the transform of u[0] only, repeat the earlier sweeps of
    the step in registers (node multipliers) and store no iterate - against the engine that stores every iterate's
    transforms: residual norms after every sweep, the end value, the node values on demand, three steps handed over through
    sdc_advance, more sweeps per step than vmax (the iterate is then stored and the sweeps go on as before), and a new
    u[0] in the middle of a step (what a receive does)."""
    from pysdc_amd.coeffs import QDELTA_GENERATORS

    M, dt = 5, 1e-3 * (512.0 / n) ** 2
    c, qi = _coeffs(M, qd)
    qe = None
    if prob == 'advdiff':
        qe = np.zeros_like(c.Qmat)
        qe[1:, 1:], qe[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    engines = []
    for v in (vmax, 0):
        e = G.engine_for(prob, dict(nvars=(n,) * ndim, nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, qe, c.nodes, c.weights)
        e.set_virtual_sweeps(v)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.3, 5), e.ctx)
        e.invalidate_spectra(1)
        e.profile_enable(True)
        engines.append(e)
    a, b = engines
    scale = float(np.max(np.abs(b.download(L.SLOT_U, 0))))

    def residuals_agree(tag):
        ra, rb = a.residual(dt), b.residual(dt)
        assert np.all(np.abs(ra[1] - rb[1]) <= 1e-9 * rb[1] + 1e-13 * scale), (tag, ra, rb)

    for step in range(3):
        for e in engines:
            e.predict(0.0, dt)
        residuals_agree((step, 'predict'))
        for k in range(4):
            for e in engines:
                e.sweep(0.0, dt)
            residuals_agree((step, k))
        if step == 1:      # somebody looks at a node value in the middle of the step; the sweeps go on afterwards
            assert rel_err(a.download(L.SLOT_U, 2), b.download(L.SLOT_U, 2)) < 1e-12
            # (f = A u amplifies the round-off of the grid-scale components of u by up to 12 nu n^2)
            assert rel_err(a.download(L.SLOT_F, M), b.download(L.SLOT_F, M)) < 1e-10
            for e in engines:
                e.sweep(0.0, dt)
            residuals_agree((step, 'after a look'))
        if step == 2:      # a new u[0] arrives: the residual of the OLD iterate against it, then sweeps from the new one
            new_u0 = b.download(L.SLOT_U, 0) * 1.001 + 1e-4
            for e in engines:
                e.upload(L.SLOT_U, 0, new_u0)
            residuals_agree((step, 'new u0'))
            for e in engines:
                e.sweep(0.0, dt)
            residuals_agree((step, 'sweep from the new u0'))
        for e in engines:
            e.end_point(dt, False)
        if step == 0:      # the end value is read before the step is handed over ...
            assert rel_err(a.download(L.SLOT_UEND), b.download(L.SLOT_UEND)) < 1e-12
        for e in engines:  # ... or it is not (steps 1 and 2): sdc_advance takes the last node's transform
            L.check(e.lib.sdc_advance(e.ctx), e.ctx)
    assert rel_err(a.download(L.SLOT_U, 0), b.download(L.SLOT_U, 0)) < 1e-12
    names = {k.split('[')[0] for k in a.profile_read()}
    if prob == 'heat_unforced':   # (real symbol; with complex multipliers reading the stored iterate is cheaper)
        assert 'spec_z_res_v0' in names and ('spec_z_res_v1' in names) == (vmax > 1), names
        assert names & {'spec_store', 'spec_store_last'}, names
    else:
        assert 'spec_z_res_v0' not in names and 'spec_z_res' in names, names
    assert not {k.split('[')[0] for k in b.profile_read()} & {'spec_z_res_v0', 'spec_store', 'spec_store_last'}
    for e in engines:
        e.close()


@pytest.mark.parametrize('n,ndim', [(1024, 2), (512, 2), (512, 3)])
def test_eager_sweeps_recompute_the_iterate_too(n, ndim):
    """every sweep stores U and F in real space (sdc_set_deferred off, the reference's update_nodes): the launch that
    transforms the new iterate back reads the transform of u[0] only as well (node multipliers, mode pairs: lines of 512
    and 1024) - against the engine that stores and reads the transforms of every iterate."""
    M, dt = 5, 1e-3 * (512.0 / n) ** 2
    c, qi = _coeffs(M, 'IE')
    engines = []
    for v in (16, 0):
        e = G.engine_for('heat_unforced', dict(nvars=(n,) * ndim, nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        e.set_deferred(False)
        e.set_virtual_sweeps(v)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.3, 5), e.ctx)
        e.invalidate_spectra(1)
        e.profile_enable(True)
        engines.append(e)
    a, b = engines
    for step in range(2):
        for e in engines:
            e.predict(0.0, dt)
        for k in range(3):
            for e in engines:
                e.sweep(0.0, dt)
            ra, rb = a.residual(dt), b.residual(dt)
            assert np.allclose(ra[1], rb[1], rtol=1e-9, atol=1e-13), (step, k)
            for m in (1, M):
                assert rel_err(a.download(L.SLOT_U, m), b.download(L.SLOT_U, m)) < 1e-12, (step, k, m)
            # (f = A u amplifies the round-off of the grid-scale components of u by up to 12 nu n^2)
            assert rel_err(a.download(L.SLOT_F, M), b.download(L.SLOT_F, M)) < 1e-10, (step, k)
        for e in engines:
            e.end_point(dt, False)
        assert rel_err(a.download(L.SLOT_UEND), b.download(L.SLOT_UEND)) < 1e-12
        for e in engines:
            e.advance()
    names = {k.split('[')[0] for k in a.profile_read()}
    assert {'spec_z_v0', 'spec_z_v1', 'spec_z_v2'} <= names and 'spec_z' not in names, names
    assert 'spec_z_v0' not in {k.split('[')[0] for k in b.profile_read()}
    for e in engines:
        e.close()


