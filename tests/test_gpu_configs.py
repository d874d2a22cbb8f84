"""GPU parity of the BASELINE configurations AT THEIR STATED SIZES, each against something that is not this product:

  * the fused 3-D Fourier-space sweep (k_spec_z -> fft_y_inv -> fft_x_norm, the bench's dominant kernels) against a
    64^3 sweep / run of the REFERENCE itself (tests/golden/sweeps_big3d.npz: heatNd_unforced, solver_type='CG');
  * config 3 (advection-diffusion IMEX 512^3): a complex eigenmode of both periodic operators, on which every node
    value is a known multiple of the mode - the multiples being the reference's imex_1st_order sweep on the scalar
    equation u' = (lam + mu) u (imex_1st_order.py:57-108), evaluated on the host in complex arithmetic;
  * config 4 (van der Pol, 1e7 trajectories): the nine golden trajectories of the reference (sweeps_vdp.npz) tiled
    to 1e7 - every tile must equal the first one bit for bit, the first one the reference to 1e-10, and the work
    counters are the tiled sums of the reference's;
  * config 5 (Allen-Cahn 256^3 / 128^3, two levels, MLSDC and 8-step PFASST): a z-invariant initial value with
    allencahn2d_imex's reaction term must reproduce, plane by plane, the REFERENCE's 2-D run at 256^2 / 128^2
    (tests/golden/runs_cfg5.npz) - which pins the 3-D transforms, the tensor-product transfer and the FAS path.
Tolerance 1e-10 relative on f64 state (BASELINE.json north_star); iteration and work counts bit-exact."""
import ctypes as C

import numpy as np
import pytest

from pysdc_amd import lib as L
from tests._cases import load_cases, rel_err
from tests import _gpu as G

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _free_gb():
    import torch

    return torch.cuda.mem_get_info()[0] / 1e9


def _sub(a):
    return np.asarray(a)[..., 1::4, 2::4, 3::4]


# ---------------------------------------------------------------------------------------------------------------
# the bench's data flow at 64^3 against the reference
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('deferred,virtual', [(True, 8), (True, 2), (True, 0), (False, 0)])
def test_fused_3d_sweep_vs_reference_64(deferred, virtual):
    """engine level, default data flow of the bench (spectral reuse, fused k_spec_z, norm-only last pass; node fields
    deferred or stored; iterates recomputed from the transform of u[0] - virtual sweeps, the default - or their transforms
    stored by every sweep).  The reference solved with CG(rtol 1e-12): its node values carry ~1e-12 of solver error."""
    case = load_cases('sweeps_big3d.npz')['cg64_heat3d_M5_IE']
    meta = case['meta']
    M, dt, t0 = len(case['coll_nodes']), meta['dt'], meta['t0']
    pp = dict(meta['prob_params'])
    pp.pop('solver_type')
    e = G.engine_for('heat_unforced', pp, M)
    e.set_deferred(deferred)
    e.set_virtual_sweeps(virtual)
    G.set_case_coeffs(e, case)
    e.upload(L.SLOT_U, 0, case['u0'])
    e.profile_enable(True)
    e.predict(t0, dt, 'spread')

    def check_fields(tag):
        u, f = e.download_u(), e.download_f()
        assert rel_err(_sub(u), case[f'{tag}_u_sub']) < TOL, tag
        assert rel_err(_sub(f), case[f'{tag}_f_sub']) < TOL, tag
        flat_u, flat_f = u.reshape(M + 1, -1), f.reshape(M + 1, -1)
        np.testing.assert_allclose(np.max(np.abs(flat_u), axis=1), case[f'{tag}_u_max'], rtol=TOL)
        np.testing.assert_allclose(np.sqrt(np.sum(flat_u**2, axis=1)), case[f'{tag}_u_l2'], rtol=TOL)
        np.testing.assert_allclose(np.max(np.abs(flat_f), axis=1), case[f'{tag}_f_max'], rtol=1e-9)
        np.testing.assert_allclose(np.sqrt(np.sum(flat_f**2, axis=1)), case[f'{tag}_f_l2'], rtol=TOL)

    def check_scalars(tag):
        for rt in ('full_abs', 'last_abs', 'full_rel', 'last_rel'):
            res, _ = e.residual(dt, rt)
            ref = float(case[f'{tag}_res_{rt}'])
            assert abs(res - ref) <= 1e-8 * abs(ref) + 1e-11, (tag, rt, res, ref)

    check_scalars('k0')
    for k in range(1, meta['nsweeps'] + 1):
        e.sweep(t0, dt)
        check_scalars(f'k{k}')          # BEFORE any field is read: the residual comes from the sweep's own epilogue
        if k == meta['nsweeps']:
            e.end_point(dt, False)
            assert rel_err(e.download(L.SLOT_UEND), case[f'k{k}_uend_0']) < TOL
        else:
            e.end_point(dt, False)
            assert rel_err(_sub(e.download(L.SLOT_UEND)), case[f'k{k}_uend_0_sub']) < TOL
    names = {k.split('[')[0] for k in e.profile_read()}
    if deferred:   # the fused Fourier-space kernels did run
        assert {'spec_z_res_v0' if virtual else 'spec_z_res', 'fft_y_inv', 'fft_x_norm'} <= names, names
        assert ('spec_z_res' in names) == (virtual < meta['nsweeps']) and ('spec_z_res_v1' in names) == (virtual > 1), names
    else:
        assert {'fft_y_inv', 'fft_x_inv'} <= names and 'fft_x_fwd[5]' not in e.profile_read(), names  # spectral reuse
    for k in (meta['nsweeps'],):
        check_fields(f'k{k}')           # node fields brought back to real space on demand (or stored, eager)
        e.end_point(dt, True)           # collocation update
        assert rel_err(_sub(e.download(L.SLOT_UEND)), case[f'k{k}_uend_1_sub']) < TOL
    e.close()


def test_fused_3d_sweep_fields_after_every_sweep_vs_reference_64():
    """same case, reading all node fields after every sweep (materialisation interleaved with Fourier-space sweeps)"""
    case = load_cases('sweeps_big3d.npz')['cg64_heat3d_M5_IE']
    meta = case['meta']
    M, dt, t0 = len(case['coll_nodes']), meta['dt'], meta['t0']
    pp = dict(meta['prob_params'])
    pp.pop('solver_type')
    e = G.engine_for('heat_unforced', pp, M)
    G.set_case_coeffs(e, case)
    e.upload(L.SLOT_U, 0, case['u0'])
    e.predict(t0, dt, 'spread')
    for k in range(0, meta['nsweeps'] + 1):
        if k:
            e.sweep(t0, dt)
        assert rel_err(_sub(e.download_u()), case[f'k{k}_u_sub']) < TOL, k
        assert rel_err(_sub(e.download_f()), case[f'k{k}_f_sub']) < TOL, k
        res, _ = e.residual(dt, 'full_abs')
        ref = float(case[f'k{k}_res_full_abs'])
        assert abs(res - ref) <= 1e-8 * abs(ref) + 1e-11, (k, res, ref)
    e.close()


def test_fused_3d_run_vs_reference_64():
    """plug-in path (controller -> sweeper_class / problem_class -> C-ABI) with the default exact Fourier solve against
    the reference's two-step run to restol 1e-9: iteration counts identical, end value <= 1e-10."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted
    from tests.test_gpu_plugin import description_from

    cases = load_cases('sweeps_big3d.npz')
    case = cases['cg64_heat3d_run_LU']
    meta = case['meta']
    meta['prob_params'].pop('solver_type')
    desc = description_from(meta)
    Ctl = controller_nonMPI(1, dict(logger_level=40), desc)
    Lv = Ctl.MS[0].levels[0]
    u0 = Lv.prob.u_init
    u0[:] = cases['cg64_heat3d_M5_IE']['u0']
    Lv.engine.profile_enable(True)
    Lv.engine.set_virtual_sweeps(8)   # (12 and 10 sweeps: the first 8 of a step recompute the iterate, the rest store it)
    uend, stats = Ctl.run(u0, meta['t0'], meta['Tend'])
    niter = [v for _, v in get_sorted(stats, type='niter', sortby='time')]
    assert niter == list(case['niter']), (niter, case['niter'])
    assert rel_err(uend.get(), case['uend']) < TOL
    res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]
    np.testing.assert_allclose(res, case['res'], rtol=1e-5, atol=5e-12)   # (the reference's CG noise: 2e-3 of 3e-10)
    names = {k.split('[')[0] for k in Lv.engine.profile_read()}
    assert {'spec_z_res_v0', 'spec_z_res_v7+', 'spec_z_res', 'fft_y_inv', 'fft_x_norm'} <= names, names


# ---------------------------------------------------------------------------------------------------------------
# config 3 at 512^3
# ---------------------------------------------------------------------------------------------------------------
def _imex_scalar_sweeps(lam, mu, dt, Q, QI, QE, nsweeps):
    """imex_1st_order.update_nodes (imex_1st_order.py:57-108) on u' = lam u (implicit) + mu u (explicit), u0 = 1,
    spread predictor, in complex arithmetic; returns the node values after every sweep."""
    M = Q.shape[0] - 1
    u = np.ones(M + 1, dtype=complex)
    fi, fe = lam * u, mu * u
    out = []
    for _ in range(nsweeps):
        integral = np.zeros(M, dtype=complex)
        for m in range(M):
            for j in range(1, M + 1):
                integral[m] += dt * Q[m + 1, j] * (fi[j] + fe[j])
            for j in range(1, M + 1):
                integral[m] -= dt * (QI[m + 1, j] * fi[j] + QE[m + 1, j] * fe[j])
            integral[m] += u[0]
        for m in range(M):
            rhs = integral[m]
            for j in range(1, m + 1):
                rhs += dt * (QI[m + 1, j] * fi[j] + QE[m + 1, j] * fe[j])
            u[m + 1] = rhs / (1.0 - dt * QI[m + 1, m + 1] * lam)
            fi[m + 1], fe[m + 1] = lam * u[m + 1], mu * u[m + 1]
        out.append(u.copy())
    return out


def _grid_max_of_real_part(z, n):
    """max over the grid of |Re(z e^{i theta})| when theta runs through all multiples of 2 pi / n"""
    t = 2.0 * np.pi * np.arange(n) / n
    return float(np.max(np.abs(z.real * np.cos(t) - z.imag * np.sin(t))))


@pytest.mark.parametrize('n', [128, 512])
def test_config3_advdiff_imex_eigenmode(n):
    """BASELINE config 3 (imex_1st_order, 3-D advection-diffusion, M=5) at 512^3 through the bench's data flow
    (k_spec_z<512,5,1,1> with the explicit symbol riding along): u0 = Re e^{i k.x}, k = 2 pi (3, 5, 2)."""
    import torch

    M, nu, c = 5, 0.02, 1.0
    if _free_gb() < 8e-9 * n**3 * 40 + 4:
        pytest.skip('not enough HBM')
    dt = 1e-3 * (512.0 / n) ** 2
    e = G.engine_for('advdiff', dict(nvars=(n, n, n), nu=nu, c=c), M)
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    coll = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    QI, QE = np.zeros_like(coll.Qmat), np.zeros_like(coll.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=coll.generator, tLeft=0).genCoeffs()
    QE[1:, 1:] = QDELTA_GENERATORS['EE'](qGen=coll.generator, tLeft=0).genCoeffs()
    e.set_coeffs(coll.Qmat, QI, QE, coll.nodes, coll.weights)
    kvec = (3, 5, 2)
    idx = np.arange(n)
    phase = (2.0 * np.pi / n) * ((kvec[0] * idx)[:, None, None] + (kvec[1] * idx)[None, :, None]
                                 + (kvec[2] * idx)[None, None, :])
    cosf = torch.from_numpy(np.cos(phase)).cuda()
    sinf = torch.from_numpy(np.sin(phase)).cuda()
    del phase
    tmp = torch.empty_like(cosf)
    dx = 1.0 / n
    lam = nu * sum(2.0 * np.cos(2.0 * np.pi * k / n) - 2.0 for k in kvec) / dx**2          # [1, -2, 1] / dx^2
    mu = -c * sum(1j * np.sin(2.0 * np.pi * k / n) for k in kvec) / dx                      # [-1/2, 0, 1/2] / dx
    K = 3
    scal = _imex_scalar_sweeps(lam, mu, dt, coll.Qmat, QI, QE, K)

    def scalar_residuals(u):
        return [1.0 + dt * sum(coll.Qmat[m, j] * (lam + mu) * u[j] for j in range(1, M + 1)) - u[m]
                for m in range(1, M + 1)]

    def deviation(ptr, z):
        """max | field - Re(z e^{i k.x}) |"""
        e.vec_axpby(e.N, 1.0, ptr, -z.real, cosf.data_ptr(), tmp.data_ptr())
        e.vec_axpby(e.N, 1.0, tmp.data_ptr(), z.imag, sinf.data_ptr(), tmp.data_ptr())
        return e.vec_amax(e.N, tmp.data_ptr())

    e.vec_copy(e.N, cosf.data_ptr(), e.ptr(L.SLOT_U, 0))
    e.invalidate_spectra(1)
    e.profile_enable(True)
    e.predict(0.0, dt)
    res, _ = e.residual(dt)
    assert abs(res - max(_grid_max_of_real_part(r, n) for r in scalar_residuals(np.ones(M + 1)))) < 1e-10
    for k in range(K):
        e.sweep(0.0, dt)
        res, norms = e.residual(dt)
        ref = [_grid_max_of_real_part(r, n) for r in scalar_residuals(scal[k])]
        np.testing.assert_allclose(norms, ref, rtol=1e-8, atol=1e-11)
        assert abs(res - max(ref)) < 1e-8 * max(ref) + 1e-11, (k, res, max(ref))
    names = {k_.split('[')[0] for k_ in e.profile_read()}
    if n >= 64:
        assert 'spec_z_res' in names and 'fft_x_norm' in names, names   # (complex symbols: iterates are stored)
    e.end_point(dt, False)
    assert deviation(e.ptr(L.SLOT_UEND), scal[-1][M]) < 1e-12
    for m in range(1, M + 1):
        assert deviation(e.ptr(L.SLOT_U, m), scal[-1][m]) < 1e-12, m
        assert deviation(e.ptr(L.SLOT_F, m, 0), lam * scal[-1][m]) < 1e-10 * abs(lam), m
        assert deviation(e.ptr(L.SLOT_F, m, 1), mu * scal[-1][m]) < 1e-10 * abs(mu), m
    # one more sweep on the general data flow (gather on the stored F) continues the same scalar recursion
    e.set_spectral_reuse(False)
    e.sweep(0.0, dt)
    nxt = _imex_scalar_sweeps(lam, mu, dt, coll.Qmat, QI, QE, K + 1)[-1]
    for m in (1, M):
        assert deviation(e.ptr(L.SLOT_U, m), nxt[m]) < 1e-12, m
    e.close()


# ---------------------------------------------------------------------------------------------------------------
# config 4 at 1e7 trajectories
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('ntraj', [10_000_000])
def test_config4_vdp_tiled_golden_trajectories(ntraj):
    import torch

    from pysdc_amd.level import Step
    from pysdc_amd.problems import vanderpol_ensemble
    from pysdc_amd.sweepers import generic_implicit

    cases = load_cases('sweeps_vdp.npz')
    names = sorted(cases)
    T = len(names)
    meta = cases[names[0]]['meta']
    M = meta['sweeper_params']['num_nodes']
    u0_tile = np.stack([cases[n]['u0'] for n in names], axis=1)            # (2, T)
    reps = -(-ntraj // T)
    u0 = np.tile(u0_tile, (1, reps))[:, :ntraj]
    desc = dict(problem_class=vanderpol_ensemble, problem_params=dict(ntraj=ntraj, u0=u0, mu=5.0, newton_tol=1e-9),
                sweeper_class=generic_implicit, sweeper_params=dict(meta['sweeper_params']),
                level_params=dict(dt=meta['dt']), step_params=dict(maxiter=4))
    Lv = Step(desc).levels[0]
    Lv.status.time = meta['t0']
    Lv.u[0] = Lv.prob.u_exact(0.0)
    Lv.sweep.predict()
    full, rest = divmod(ntraj, T)

    def counter_total(key, k):
        per = [int(cases[n][f'work_{key}'][k - 1]) for n in names]
        return full * sum(per) + sum(per[:rest])

    for k in range(1, meta['nsweeps'] + 1):
        Lv.sweep.update_nodes()
        Lv.sweep.compute_residual()
        ref_res = max(float(cases[n][f'k{k}_res_full_abs']) for n in names)
        assert abs(Lv.status.residual - ref_res) <= 1e-9 * max(ref_res, 1e-6), k
        assert Lv.prob.work_counters['newton'].niter == counter_total('newton', k), k
        assert Lv.prob.work_counters['rhs'].niter == counter_total('rhs', k), k
        for slab in (Lv.u, Lv.f):
            for m in range(1, M + 1):
                t = slab[m].as_torch().reshape(2, ntraj)
                first = t[:, :T]
                # every complete tile equals the first one bit for bit (compared on the device)
                tiles = t[:, :full * T].reshape(2, full, T)
                assert bool(torch.all(tiles == first[:, None, :])), (k, m)
                if rest:
                    assert bool(torch.all(t[:, full * T:] == first[:, :rest])), (k, m)
        U = np.stack([Lv.u[m].as_torch().reshape(2, ntraj)[:, :T].cpu().numpy() for m in range(M + 1)])
        F = np.stack([Lv.f[m].as_torch().reshape(2, ntraj)[:, :T].cpu().numpy() for m in range(M + 1)])
        for i, n in enumerate(names):
            assert rel_err(U[:, :, i], cases[n][f'k{k}_u']) < TOL, (n, k)
            assert rel_err(F[:, :, i], cases[n][f'k{k}_f']) < 1e-9, (n, k)


# ---------------------------------------------------------------------------------------------------------------
# config 5 at 256^3 / 128^3
# ---------------------------------------------------------------------------------------------------------------
def _ac3d_class():
    from pysdc_amd.problems import allencahn_imex

    class allencahn3d_with_2d_reaction(allencahn_imex):
        """the product's 3-D pseudo-spectral level (FFT pipeline, symbol -(2 pi k)^2 per axis) carrying the reaction
        term of the reference's allencahn2d_imex (AllenCahn_2D_FFT.py: 1/eps^2 u (1 - u^nu)), so that a z-invariant
        state evolves exactly like the reference's 2-D problem"""

        def __init__(self, nvars=None, nu=2, eps=0.04, radius=0.25):
            super().__init__(nvars=nvars, eps=eps, radius=radius)
            self._nu2d = nu

        def configure_engine(self, engine):
            engine.set_symbol(0, self._symbol())
            engine.set_reaction(1, 1.0 / self.eps**2, 0.0, int(self._nu2d))

    return allencahn3d_with_2d_reaction


@pytest.mark.parametrize('name', ['cfg5_ac2d_mlsdc', 'cfg5_ac2d_pfasst_P8'])
def test_config5_allencahn_256_two_level_vs_2d_reference(name):
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted
    from pysdc_amd.sweepers import imex_1st_order
    from pysdc_amd.transfer import mesh_to_mesh

    case = load_cases('runs_cfg5.npz')[name]
    meta = case['meta']
    nf, nc = meta['prob_params']['nvars'][0][0], meta['prob_params']['nvars'][1][0]
    assert (nf, nc) == (256, 128)
    P = meta['num_procs']
    if _free_gb() < P * 5.5 + 4:
        pytest.skip('not enough HBM')
    pp = dict(nvars=[(nf,) * 3, (nc,) * 3], nu=meta['prob_params']['nu'], eps=meta['prob_params']['eps'],
              radius=meta['prob_params']['radius'])
    desc = dict(problem_class=_ac3d_class(), problem_params=pp, sweeper_class=imex_1st_order,
                sweeper_params=dict(meta['sweeper_params']), level_params=dict(meta['level_params']),
                step_params=dict(maxiter=meta['maxiter']), space_transfer_class=mesh_to_mesh,
                space_transfer_params=dict(iorder=meta['iorder'], rorder=meta['rorder'], periodic=True))
    Ctl = controller_nonMPI(P, dict(logger_level=40, **meta['controller_params']), desc)
    prob = Ctl.MS[0].levels[0].prob
    u0 = prob.u_init
    u0[:] = np.ascontiguousarray(np.broadcast_to(case['u0'][:, :, None], (nf, nf, nf)))
    uend, stats = Ctl.run(u0, meta['t0'], meta['Tend'])
    niter = [v for _, v in get_sorted(stats, type='niter', sortby='time')]
    assert niter == list(case['niter']), (niter, list(case['niter']))
    got = uend.get()
    scale = float(np.max(np.abs(case['uend'])))
    assert float(np.max(np.abs(got - case['uend'][:, :, None]))) < TOL * scale          # every z-plane
    res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]
    np.testing.assert_allclose(res, case['res'], rtol=1e-5, atol=1e-11)


@pytest.mark.parametrize('name', list(load_cases('runs_ac_fft.npz')))
def test_allencahn_3d_runs_with_fourier_transfer_vs_2d_reference(name):
    """two-level MLSDC / PFASST with the 3-D Fourier space transfer (mesh_to_mesh_fft3d; the reference's own needs mpi4py_fft,
    TransferMesh_MPIFFT.py:51-136): a z-invariant 32^3 / 16^3 Allen-Cahn run reproduces the reference's 2-D run with
    mesh_to_mesh_fft2d (tests/golden/runs_ac_fft.npz) in every plane - iteration counts, end value, residuals"""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted
    from pysdc_amd.sweepers import imex_1st_order
    from pysdc_amd.transfer import mesh_to_mesh_fft3d

    case = load_cases('runs_ac_fft.npz')[name]
    meta = case['meta']
    assert meta['transfer'] == 'mesh_to_mesh_fft2d'
    nf, nc = meta['prob_params']['nvars'][0][0], meta['prob_params']['nvars'][1][0]
    pp = dict(nvars=[(nf,) * 3, (nc,) * 3], nu=meta['prob_params']['nu'], eps=meta['prob_params']['eps'],
              radius=meta['prob_params']['radius'])
    desc = dict(problem_class=_ac3d_class(), problem_params=pp, sweeper_class=imex_1st_order,
                sweeper_params=dict(meta['sweeper_params']), level_params=dict(meta['level_params']),
                step_params=dict(maxiter=meta['maxiter']), space_transfer_class=mesh_to_mesh_fft3d, space_transfer_params={})
    Ctl = controller_nonMPI(meta['num_procs'], dict(logger_level=40, **meta['controller_params']), desc)
    prob = Ctl.MS[0].levels[0].prob
    u0 = prob.u_init
    u0[:] = np.ascontiguousarray(np.broadcast_to(case['u0'][:, :, None], (nf, nf, nf)))
    uend, stats = Ctl.run(u0, meta['t0'], meta['Tend'])
    niter = [v for _, v in get_sorted(stats, type='niter', sortby='time')]
    assert niter == list(case['niter']), (niter, list(case['niter']))
    scale = float(np.max(np.abs(case['uend'])))
    assert float(np.max(np.abs(uend.get() - case['uend'][:, :, None]))) < TOL * scale          # every z-plane
    res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]
    np.testing.assert_allclose(res, case['res'], rtol=1e-5, atol=1e-11)
