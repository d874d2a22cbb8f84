"""GPU parity: the HIP engine (through the C-ABI) against golden vectors of the reference and against the
oracle on seeded inputs.  Tolerance: <= 1e-10 relative on f64 state (BASELINE.json north_star); residual
values to 1e-8 relative (they are differences of O(1) quantities)."""
import numpy as np
import pytest

from pysdc_amd import lib as L
from tests._cases import load_cases, make_oracle_problem, make_oracle_coll, rel_err
from tests import _gpu as G

pytestmark = pytest.mark.gpu
TOL = 1e-10


def supported(case):
    meta = case['meta']
    pp = meta['prob_params']
    if 'k0_u' not in case:
        return False              # run cases / cases stored as subsamples have their own tests
    if meta['sweeper_params'].get('initial_guess') == 'random':
        return False              # drawn node by node on the host: covered through the plug-in classes
    if meta['prob'] == 'vanderpol':
        return False
    if pp.get('bc', 'periodic') != 'periodic':
        return False
    n = G.norm_nvars(pp['nvars'])[0]
    return n & (n - 1) == 0 or (n % 3 == 0 and 24 <= n <= 768 and (n // 3) & (n // 3 - 1) == 0)   # 2^p, 3 * 2^p


# (sweeps_radix3.npz: the even grids the reference accepts that are not powers of two - 96, 192, 48^2, 24^3 with its direct
# solver - on the engine's line transforms of length 3 * 2^p)
SWEEP_FILES = ['sweeps_heat.npz', 'sweeps_imex.npz', 'sweeps_adv.npz', 'sweeps_guess.npz', 'sweeps_pin1024.npz', 'sweeps_radix3.npz', 'sweeps_radix5.npz']
SWEEP_CASES = [(f, n) for f in SWEEP_FILES for n, c in load_cases(f).items() if supported(c)]


@pytest.mark.parametrize('reuse', [True, False])
@pytest.mark.parametrize('fname,name', SWEEP_CASES)
def test_sweep_vs_golden(fname, name, reuse):
    """reuse=True: sweeps gather on the cached Fourier transforms (spectral reuse); False: every sweep gathers
    on the F slab and transforms M fields (the reference's data flow).  Both must match the reference."""
    case = load_cases(fname)[name]
    meta = case['meta']
    M = len(case['coll_nodes'])
    dt, t0 = meta['dt'], meta['t0']
    e = G.engine_for(meta['prob'], meta['prob_params'], M)
    e.set_spectral_reuse(reuse)
    G.set_case_coeffs(e, case)
    G.set_forcing_times(e, meta, t0, dt, case['coll_nodes'])
    e.upload(L.SLOT_U, 0, case['u0'])
    if meta['has_tau']:
        e.set_tau_active(True)
        for m in range(M):
            e.upload(L.SLOT_TAU, m, case['tau'][m])
    e.predict(t0, dt, meta['sweeper_params'].get('initial_guess', 'spread'))
    right = bool(case['coll_right_is_node'])

    def check(tag):
        assert rel_err(e.download_u(), case[f'{tag}_u']) < TOL, tag
        assert rel_err(e.download_f(), case[f'{tag}_f']) < TOL, tag
        for rt in ('full_abs', 'last_abs', 'full_rel', 'last_rel'):
            res, _ = e.residual(dt, rt)
            ref = float(case[f'{tag}_res_{rt}'])
            scale = max(1.0, float(np.max(np.abs(case[f'{tag}_u'])))) if 'abs' in rt else 1.0
            assert abs(res - ref) <= 1e-8 * abs(ref) + 1e-11 * scale, (tag, rt, res, ref)
        for dcu in (False, True):
            e.end_point(dt, dcu or not right)
            assert rel_err(e.download(L.SLOT_UEND), case[f'{tag}_uend_{int(dcu)}']) < TOL, (tag, dcu)

    check('k0')
    for k in range(1, meta['nsweeps'] + 1):
        if f'k{k}_QI' in case:
            G.set_case_coeffs(e, case, QI=case[f'k{k}_QI'])
        e.sweep(t0, dt)
        check(f'k{k}')
    e.close()


@pytest.mark.parametrize('nvars,order', [((32,), 2), ((64, 64), 4), ((16, 16, 16), 2), ((32, 32, 32), 6),
                                         ((128,), 8), ((256, 256), 2), ((2048,), 2), ((2048, 2048), 2)])
def test_eval_f_and_solve_vs_oracle(nvars, order):
    from oracle import sdc_oracle as O

    P = O.HeatUnforced(nvars if len(nvars) > 1 else nvars[0], 0.1, 2, order=order)
    e = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1, order=order), 3)
    rng = np.random.default_rng(11)
    u = rng.standard_normal(nvars)
    e.upload(L.SLOT_U, 0, u)
    e.eval_f(e.ptr(L.SLOT_U, 0), 0.0, e.ptr(L.SLOT_F, 0))
    assert rel_err(e.download(L.SLOT_F, 0), P.eval_f(u, 0.0)) < 1e-13
    factor = 63.0 / (4.0 * len(nvars) * 0.1 / P.dx**2)
    e.solve(e.ptr(L.SLOT_U, 0), factor, e.ptr(L.SLOT_U, 1))
    got = e.download(L.SLOT_U, 1)
    ref = O.spectral_solve(P, u, factor) if np.prod(nvars) > 5000 else P.solve_system(u, factor, u, 0.0)
    assert rel_err(got, ref) < 1e-12
    # in place is allowed
    e.solve(e.ptr(L.SLOT_U, 0), factor, e.ptr(L.SLOT_U, 0))
    assert rel_err(e.download(L.SLOT_U, 0), ref) < 1e-12
    e.close()


def test_vector_ops():
    e = G.engine_for('heat_unforced', dict(nvars=(16, 16), nu=0.1), 2)
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal((16, 16)), rng.standard_normal((16, 16))
    e.upload(L.SLOT_U, 0, x)
    e.upload(L.SLOT_U, 1, y)
    e.vec_axpby(e.N, 2.5, e.ptr(L.SLOT_U, 0), -0.5, e.ptr(L.SLOT_U, 1), e.ptr(L.SLOT_U, 2))
    assert np.array_equal(e.download(L.SLOT_U, 2), 2.5 * x + (-0.5) * y)
    assert e.vec_amax(e.N, e.ptr(L.SLOT_U, 0)) == float(np.max(np.abs(x)))
    e.vec_fill(e.N, 3.25, e.ptr(L.SLOT_U, 2))
    assert np.all(e.download(L.SLOT_U, 2) == 3.25)
    e.vec_copy(e.N, e.ptr(L.SLOT_U, 1), e.ptr(L.SLOT_U, 2))
    assert np.array_equal(e.download(L.SLOT_U, 2), y)
    xn = x.copy()
    xn[3, 4] = np.nan
    e.upload(L.SLOT_U, 0, xn)
    assert np.isnan(e.vec_amax(e.N, e.ptr(L.SLOT_U, 0)))   # np.max propagates NaN; so does abs()
    e.close()


def test_error_paths():
    from pysdc_amd.engine import SweepEngine
    from pysdc_amd.errors import ParameterError, UnlockError

    with pytest.raises(ParameterError):
        SweepEngine((7, 7), 3)
    with pytest.raises(ParameterError):
        SweepEngine((8, 8, 8, 8), 3)
    e = SweepEngine((8, 8), 3)
    with pytest.raises(UnlockError):
        e.sweep(0.0, 0.1)          # coefficients not set
    bad = np.zeros((4, 4))
    bad[1, 2] = 1.0
    with pytest.raises(ParameterError):
        e.set_coeffs(np.zeros((4, 4)), bad, None, np.ones(3), np.ones(3))
    e.close()
    e = SweepEngine((12, 12), 3)   # even but not 2^p: no FFT; symmetric operators fall back to CG, the others to GMRES
    for weights in ([-0.5, 0.0, 0.5], [1.0, -2.0, 1.0]):
        e.set_stencil(0, [-1, 0, 1], weights)
        e.upload(L.SLOT_U, 0, np.random.default_rng(0).standard_normal((12, 12)))
        e.solve(e.ptr(L.SLOT_U, 0), 0.1, e.ptr(L.SLOT_U, 1))
        e.eval_f(e.ptr(L.SLOT_U, 1), 0.0, e.ptr(L.SLOT_F, 1))
        back = e.download(L.SLOT_U, 1) - 0.1 * e.download(L.SLOT_F, 1)          # (I - 0.1 A) x
        assert np.max(np.abs(back - e.download(L.SLOT_U, 0))) < 1e-12
    assert e.work_counters()['CG'] == 0 and e.work_counters()['GMRES'] == 0   # not the user's solvers: nothing counted
    e.close()


@pytest.mark.parametrize('where', ['u0', 'u0_after_sweep'])
def test_residual_propagates_nan_like_np_max(where):
    """mesh.__abs__ is np.max(np.abs(.)): a NaN in a residual field makes the residual NaN (the controllers' convergence
    test then fails loudly).  The Fourier-space sweeps reduce the norm in the last inverse pass with fmax (which drops
    NaNs - a plain fmax chain would report 0 here) plus a test on one output per thread, which is exact because every
    output of a transform is a sum over all inputs of its column."""
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    M, n, dt = 5, 64, 1e-2
    e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=0.1), M)
    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    qi = np.zeros_like(c.Qmat)
    qi[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
    e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
    u0 = np.random.default_rng(0).standard_normal((n, n, n))
    bad = u0.copy()
    bad[5, 6, 7] = np.nan
    e.upload(L.SLOT_U, 0, bad if where == 'u0' else u0)
    e.predict(0.0, dt)
    e.profile_enable(True)
    e.sweep(0.0, dt)
    res, norms = e.residual(dt)
    if where == 'u0':
        assert np.isnan(res) and np.all(np.isnan(norms))
    else:
        assert np.isfinite(res)
        e.upload(L.SLOT_U, 0, bad)     # a poisoned value arrives as the new u[0] (a receive): residual of the cached iterate
        res, norms = e.residual(dt)
        assert np.isnan(res) and np.all(np.isnan(norms))
    assert any(k.startswith('fft_x_norm') for k in e.profile_read())
    e.close()


@pytest.mark.parametrize('fname,name', [('sweeps_heat.npz', 'heat3d_o2_M5_IE_dt0.1'), ('sweeps_heat.npz', 'heat2d_o4_M3_IE_dt0.001'),
                                        ('sweeps_imex.npz', 'advdiff3d_M5'), ('sweeps_imex.npz', 'forced2d_M3')])
def test_integrate_vs_golden_node_values(fname, name):
    """Sweeper.integrate (generic_implicit.py:29-49 / imex_1st_order.py:37-55): dt * sum_j Qmat[m+1, j] * f[j], accumulated
    left to right - from the reference's own f values at the nodes (golden), after the predictor and after every sweep."""
    import torch

    case = load_cases(fname)[name]
    meta = case['meta']
    M = len(case['coll_nodes'])
    dt, t0 = meta['dt'], meta['t0']
    e = G.engine_for(meta['prob'], meta['prob_params'], M)
    G.set_case_coeffs(e, case)
    G.set_forcing_times(e, meta, t0, dt, case['coll_nodes'])
    e.upload(L.SLOT_U, 0, case['u0'])
    e.predict(t0, dt, 'spread')
    out = torch.empty(M * e.N, dtype=torch.float64, device='cuda')
    for k in range(0, meta['nsweeps'] + 1):
        if k:
            e.sweep(t0, dt)
        e.integrate(dt, [out.data_ptr() + 8 * m * e.N for m in range(M)])
        got = out.cpu().numpy().reshape((M,) + e.nvars)
        f = case[f'k{k}_f']
        fsum = f if f.ndim == got.ndim else f[:, 0] + f[:, 1]          # IMEX: impl + expl
        for m in range(M):
            ref = np.zeros(e.nvars)
            for j in range(1, M + 1):
                ref += dt * case['coll_Qmat'][m + 1, j] * fsum[j]
            assert rel_err(got[m], ref) < TOL, (k, m)
    e.close()


@pytest.mark.parametrize('virtual', [8, 0])
def test_put_off_end_value_survives_a_predictor(virtual):
    """sweep, compute_end_point (put off while sweeps stay in Fourier space), predict, THEN read the end value: the
    reference's predict leaves L.uend alone (core/sweeper.py:125-162), so the value of the sweep must come out - the
    engine has to transform it back before the predictor drops the iterate it belongs to"""
    n, M, dt = 64, 3, 1e-2
    u0 = np.random.default_rng(4).standard_normal((n, n, n))
    ends = []
    for deferred in (True, False):
        e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=0.1), M)
        from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

        c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
        qi = np.zeros_like(c.Qmat)
        qi[1:, 1:] = QDELTA_GENERATORS['LU'](qGen=c.generator, tLeft=0).genCoeffs()
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        e.set_deferred(deferred)
        e.set_virtual_sweeps(virtual)
        e.upload(L.SLOT_U, 0, u0)
        e.predict(0.0, dt)
        e.sweep(0.0, dt)
        e.sweep(0.0, dt)
        e.end_point(dt, False)
        e.predict(0.0, dt)                     # drops the iterate; UEND has not been read yet
        ends.append(e.download(L.SLOT_UEND))
        e.close()
    assert rel_err(ends[0], ends[1]) < 1e-12 and np.max(np.abs(ends[1] - u0)) > 1e-3


@pytest.mark.parametrize('nvars', [(64, 64, 64), (128, 128)])
def test_start_value_replaced_as_a_spectrum_equals_replaced_as_a_field(nvars):
    """sdc_replace_u0_spectrum (time-parallel hand-over that carries spectra): node norms of the residual against the new
    start value - from one more field through the inverse passes - and the sweeps that follow equal those of an engine
    that was handed the same value as a field (sdc_replace_u0 after kept residual fields)"""
    import torch

    M, dt = 5, 2e-3
    rng = np.random.default_rng(9)
    u0, v = rng.standard_normal(nvars), rng.standard_normal(nvars)

    def engine(spectral):
        from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

        e = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1), M)
        c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
        qi = np.zeros_like(c.Qmat)
        qi[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        e.set_virtual_sweeps(0)
        e.set_early_end_point(True)
        if spectral:
            L.check(e.lib.sdc_set_wire_spectral(e.ctx, 1), e.ctx)
        else:
            e.set_keep_residual_fields(True)
        e.upload(L.SLOT_U, 0, u0)
        e.predict(0.0, dt)
        e.sweep(0.0, dt)
        e.residual(dt)
        return e

    a, b = engine(True), engine(False)
    # the new start value: as a field for b; its spectrum for a (made by a third engine's forward transform = end spectrum)
    b.upload(L.SLOT_UEND, 0, v)
    b.replace_u0(b.ptr(L.SLOT_UEND))
    src = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1), M)
    src.set_coeffs(*[np.zeros((M + 1, M + 1))] * 2, None, np.ones(M), np.ones(M))
    src.upload(L.SLOT_UEND, 0, v)
    nspec = 2 * (nvars[0] // 2 + 1) * int(np.prod(nvars[1:]))
    from pysdc_amd.hip_mesh import _CAI

    spec = torch.as_tensor(_CAI(src.end_spectrum(), nspec, src), device='cuda')
    torch.as_tensor(_CAI(a.spectrum_inbox(), nspec, a), device='cuda').copy_(spec)
    torch.cuda.synchronize()
    a.replace_u0_spectrum()
    ra, na = a.residual(dt)
    rb, nb = b.residual(dt)
    np.testing.assert_allclose(na, nb, rtol=1e-9)
    for k in range(2):
        a.sweep(0.0, dt)
        b.sweep(0.0, dt)
        np.testing.assert_allclose(a.residual(dt)[1], b.residual(dt)[1], rtol=1e-8)
    a.end_point(dt, False)
    b.end_point(dt, False)
    assert rel_err(a.download(L.SLOT_UEND), b.download(L.SLOT_UEND)) < 1e-11
    assert rel_err(a.download(L.SLOT_U, 0), v) < 1e-12          # U[0] itself, produced from the spectrum on demand
    for e in (a, b, src):
        e.close()


def test_error_paths_of_the_round_3_entry_points():
    """new C-ABI calls fail loudly with the reference's exception types: no communicator, a level that does not sweep in
    Fourier space asked for the spectral hand-over, a banded operator that does not fit, an unknown solver"""
    from pysdc_amd.comm import DeviceComm, shm_unique_id
    from pysdc_amd.engine import SweepEngine
    from pysdc_amd.errors import CommunicationError, ParameterError, UnlockError

    e = SweepEngine((16, 16), 3)
    for call in (lambda: L.check(e.lib.sdc_comm_handover_post(e.ctx, 2), e.ctx),
                 lambda: L.check(e.lib.sdc_comm_exchange(e.ctx, 1, -1), e.ctx),
                 lambda: L.check(e.lib.sdc_comm_bcast_end_spectrum(e.ctx, 0), e.ctx)):
        with pytest.raises(UnlockError):           # SDC_ERR_STATE: no communicator (sdc_comm_init)
            call()
    with pytest.raises(UnlockError):               # no stencil set: not a level that sweeps in Fourier space
        L.check(e.lib.sdc_set_wire_spectral(e.ctx, 1), e.ctx)
    with pytest.raises(UnlockError):
        e.replace_u0_spectrum()                    # no spectrum inbox
    with pytest.raises(ParameterError):
        e.set_banded_operator(np.zeros((17, 3), dtype=np.int32), np.zeros((17, 3)))   # 17 interior points > 16 per axis
    with pytest.raises(ParameterError):
        e.set_banded_operator(np.full((15, 3), 15, dtype=np.int32), np.zeros((15, 3)))  # column index out of range
    with pytest.raises(KeyError):
        e.set_solver('BiCGStab')
    comm = DeviceComm(e, 1, 0, uid=shm_unique_id())
    with pytest.raises(ParameterError):
        comm.handover_post(2)                      # more active ranks than the communicator has
    with pytest.raises(ParameterError):
        comm.exchange(send_to=3)
    assert comm.info() == dict(rank=0, size=1, two_hop_handovers=0, mesh_broadcasts=0, wire='shm')
    comm.close()
    with pytest.raises(CommunicationError):        # a peer that never shows up: the mailbox wire times out instead of hanging
        import os

        os.environ['SDC_COMM_TIMEOUT'] = '0.3'
        try:
            lonely = DeviceComm(e, 2, 1, uid=shm_unique_id())
            lonely.exchange(recv_from=0)
        finally:
            del os.environ['SDC_COMM_TIMEOUT']
    lonely.close()
    e.close()
