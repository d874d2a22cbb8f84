"""GPU: space transfer kernels, FAS restriction / prolongation and MLSDC / PFASST runs on device levels
against golden vectors of the reference (tests/golden/{transfer,fas,runs_ml}.npz)."""
import numpy as np
import pytest

from tests._cases import load_cases, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _tup(v):
    return tuple(v) if isinstance(v, list) else v


def _classes():
    from pysdc_amd import problems as P, sweepers as S

    return ({'heat_unforced': P.heatNd_unforced, 'heat_forced': P.heatNd_forced, 'allencahn2d': P.allencahn2d_imex},
            {'generic_implicit': S.generic_implicit, 'imex_1st_order': S.imex_1st_order})


def _description(meta, lp, io=6, ro=2):
    from pysdc_amd.transfer import mesh_to_mesh

    probs, sweeps = _classes()
    pp = {k: ([_tup(x) if isinstance(x, list) else x for x in v] if isinstance(v, list) and k == 'nvars' else v)
          for k, v in meta['prob_params'].items()}
    if isinstance(pp['nvars'], list) and not isinstance(pp['nvars'][0], (tuple, int)):
        pp['nvars'] = [tuple(x) for x in pp['nvars']]
    return dict(problem_class=probs[meta['prob']], problem_params=pp, sweeper_class=sweeps[meta['sweeper']],
                sweeper_params=dict(meta['sweeper_params']), level_params=dict(lp),
                step_params=dict(maxiter=meta.get('maxiter', 10)), space_transfer_class=mesh_to_mesh,
                space_transfer_params=dict(iorder=io, rorder=ro, periodic=True))


@pytest.mark.parametrize('name', list(load_cases('transfer.npz')))
def test_space_transfer_kernels(name):
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.transfer import mesh_to_mesh

    c = load_cases('transfer.npz')[name]
    m = c['meta']
    nf, nc = _tup(m['nf']), _tup(m['nc'])
    pf = heatNd_unforced(nvars=nf, nu=0.1, freq=2)
    pc = heatNd_unforced(nvars=nc, nu=0.1, freq=2)
    T = mesh_to_mesh(pf, pc, dict(iorder=m['iorder'], rorder=m['rorder'], periodic=True))
    F, G = pf.u_init, pc.u_init
    F[:] = c['fine']
    G[:] = c['coarse']
    assert rel_err(T.restrict(F).get(), c['restricted']) < 1e-14
    assert rel_err(T.prolong(G).get(), c['prolonged']) < 1e-14


@pytest.mark.parametrize('name', list(load_cases('transfer_fft.npz')))
def test_fourier_transfer_kernels(name):
    """mesh_to_mesh_fft (1-D) / mesh_to_mesh_fft2d on the device against vectors of the reference classes."""
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.transfer import mesh_to_mesh_fft, mesh_to_mesh_fft2d

    c = load_cases('transfer_fft.npz')[name]
    m = c['meta']
    d = 1 if m['kind'] == 'fft1d' else 2
    pf = heatNd_unforced(nvars=(m['nf'],) * d, nu=0.1, freq=2)
    pc = heatNd_unforced(nvars=(m['nc'],) * d, nu=0.1, freq=2)
    T = (mesh_to_mesh_fft if d == 1 else mesh_to_mesh_fft2d)(pf, pc, {})
    F, G = pf.u_init, pc.u_init
    F[:] = c['fine']
    G[:] = c['coarse']
    assert np.array_equal(T.restrict(F).get(), c['restricted'])
    assert rel_err(T.prolong(G).get(), c['prolonged']) < 1e-13


@pytest.mark.parametrize('name', [n for n, c in load_cases('transfer_fft.npz').items() if c['meta']['kind'] != 'fft1d'])
@pytest.mark.parametrize('axis', [0, 1, 2])
def test_fourier_transfer_in_3d_is_the_2d_rule_plane_by_plane(name, axis):
    """mesh_to_mesh_fft3d (the reference's 3-D Fourier transfer needs mpi4py_fft: TransferMesh_MPIFFT.py:51-136) pinned
    to the reference's 2-D class: a field that does not depend on one axis is restricted / prolonged in every plane across
    that axis exactly as the golden vectors of mesh_to_mesh_fft2d say - for each of the three axes"""
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.transfer import mesh_to_mesh_fft3d

    c = load_cases('transfer_fft.npz')[name]
    m = c['meta']
    pf = heatNd_unforced(nvars=(m['nf'],) * 3, nu=0.1, freq=2)
    pc = heatNd_unforced(nvars=(m['nc'],) * 3, nu=0.1, freq=2)
    T = mesh_to_mesh_fft3d(pf, pc, {})

    def lift(plane, n):   # the 2-D field repeated along `axis`
        return np.ascontiguousarray(np.moveaxis(np.broadcast_to(plane, (n,) + plane.shape), 0, axis))

    F, G = pf.u_init, pc.u_init
    F[:] = lift(c['fine'], m['nf'])
    G[:] = lift(c['coarse'], m['nc'])
    assert np.array_equal(T.restrict(F).get(), lift(c['restricted'], m['nc']))
    got = T.prolong(G).get()
    assert rel_err(got, lift(c['prolonged'], m['nf'])) < 1e-13
    # a field that depends on all three axes: against NumPy's fftn with the same corner rule
    rng = np.random.default_rng(5)
    g = rng.standard_normal((m['nc'],) * 3)
    G[:] = g
    gh = np.fft.fftn(g)
    h, nf = m['nc'] // 2, m['nf']
    fh = np.zeros((nf,) * 3, dtype=complex)
    lo, hi_c, hi_f = slice(0, h), slice(h, None), slice(nf - h, None)
    for sa in ((lo, lo), (hi_c, hi_f)):
        for sb in ((lo, lo), (hi_c, hi_f)):
            for sc in ((lo, lo), (hi_c, hi_f)):
                fh[sa[1], sb[1], sc[1]] = gh[sa[0], sb[0], sc[0]]
    want = np.real(np.fft.ifftn(fh)) * 2 * (nf // m['nc']) ** 2
    assert rel_err(T.prolong(G).get(), want) < 1e-13


@pytest.mark.parametrize('name', list(load_cases('transfer_dirichlet.npz')))
def test_dirichlet_transfer_kernels(name):
    """non-periodic mesh_to_mesh between 1-D dirichlet-zero grids on the device against the reference class."""
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.transfer import mesh_to_mesh

    c = load_cases('transfer_dirichlet.npz')[name]
    m = c['meta']
    pf = heatNd_unforced(nvars=m['nf'], nu=0.1, freq=2, bc='dirichlet-zero')
    pc = heatNd_unforced(nvars=m['nc'], nu=0.1, freq=2, bc='dirichlet-zero')
    T = mesh_to_mesh(pf, pc, dict(iorder=m['iorder'], rorder=m['rorder'], periodic=False))
    F, G = pf.u_init, pc.u_init
    F[:] = c['fine']
    G[:] = c['coarse']
    assert rel_err(T.restrict(F).get(), c['restricted']) < 1e-14
    assert rel_err(T.prolong(G).get(), c['prolonged']) < 1e-14


@pytest.mark.parametrize('name', list(load_cases('fas.npz')))
def test_fas_on_device(name):
    from pysdc_amd.level import Step

    case = load_cases('fas.npz')[name]
    meta = case['meta']
    S = Step(_description(meta, dict(dt=meta['dt'])))
    F, G = S.levels
    for L in S.levels:
        L.status.time = meta['t0']
    u0 = F.prob.u_init
    u0[:] = case['u0']
    F.u[0] = u0
    F.sweep.predict()
    F.sweep.update_nodes()

    def stack(lst):
        return np.stack([np.asarray(x) for x in lst])

    def check(tag, coarse=True):
        assert rel_err(stack(F.u), case[f'{tag}_fu']) < TOL, tag
        assert rel_err(stack(F.f), case[f'{tag}_ff']) < TOL, tag
        if coarse:
            assert rel_err(stack(G.u), case[f'{tag}_gu']) < TOL, tag
            assert rel_err(stack(G.f), case[f'{tag}_gf']) < TOL, tag
            tau_scale = max(float(np.max(np.abs(case[f'{tag}_gu']))), 1.0)
            assert np.max(np.abs(stack(G.tau) - case[f'{tag}_gtau'])) < 1e-9 * tau_scale, tag

    check('a', coarse=False)
    S.transfer(F, G)
    check('b')
    G.sweep.update_nodes()
    G.sweep.compute_residual()
    assert abs(G.status.residual - float(case['c_gres'])) < 1e-9
    check('c')
    S.transfer(G, F)
    check('d')


ML_RUNS = ([('runs_ml.npz', n) for n in load_cases('runs_ml.npz')] + [('runs_ml8.npz', n) for n in load_cases('runs_ml8.npz')] + [('runs_ac.npz', n) for n in load_cases('runs_ac.npz')]
           + [('runs_ac_fft.npz', n) for n in load_cases('runs_ac_fft.npz')]
           + [('runs_ml_dirichlet.npz', n) for n in load_cases('runs_ml_dirichlet.npz')])


@pytest.mark.parametrize('fname,name', ML_RUNS)
def test_mlsdc_pfasst_on_device(fname, name):
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.stats import get_sorted

    case = load_cases(fname)[name]
    meta = case['meta']
    desc = _description(meta, meta['level_params'], meta.get('iorder', 6), meta.get('rorder', 2))
    if not meta.get('periodic', True):
        desc['space_transfer_params']['periodic'] = False
    if meta.get('transfer') == 'mesh_to_mesh_fft2d':
        from pysdc_amd.transfer import mesh_to_mesh_fft2d

        desc['space_transfer_class'], desc['space_transfer_params'] = mesh_to_mesh_fft2d, {}
    C = controller_nonMPI(meta['num_procs'], dict(logger_level=40, **meta['controller_params']), desc)
    P = C.MS[0].levels[0].prob
    u0 = P.u_init
    u0[:] = case['u0']
    uend, stats = C.run(u0, meta['t0'], meta['Tend'])
    niter = get_sorted(stats, type='niter', sortby='time')
    assert [v for _, v in niter] == list(case['niter'])          # bit-exact iteration counts
    assert rel_err(uend.get(), case['uend']) < TOL
    res = [v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')]
    np.testing.assert_allclose(res, case['res'], rtol=1e-5, atol=1e-11)


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('name', list(load_cases('sweeps_ac.npz')))
def test_allencahn_sweeps_on_device(name, fused):
    """pseudo-spectral Allen-Cahn (nonlinear explicit part): node-by-node IMEX sweeps on the device against
    golden sweeps of the reference's allencahn2d_imex."""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import allencahn2d_imex
    from pysdc_amd.sweepers import imex_1st_order

    case = load_cases('sweeps_ac.npz')[name]
    meta = case['meta']
    pp = dict(meta['prob_params'])
    pp['nvars'] = tuple(pp['nvars'])
    pc = allencahn2d_imex if fused else type('ac2d_python_nodes', (allencahn2d_imex,), {'fused': False})
    S = Step(dict(problem_class=pc, problem_params=pp, sweeper_class=imex_1st_order,
                  sweeper_params=dict(meta['sweeper_params']), level_params=dict(dt=meta['dt']),
                  step_params=dict(maxiter=10)))
    L = S.levels[0]
    L.status.time = meta['t0']
    from oracle import sdc_oracle as O

    assert rel_err(L.prob.u_exact(0.0).get(), O.AllenCahn2D(**pp).u_exact(0.0)) < 1e-14
    u0 = L.prob.u_init
    u0[:] = case['u0']          # circle + seeded noise, as stored with the golden case
    L.u[0] = u0
    L.sweep.predict()

    def check(tag):
        assert rel_err(np.stack([np.asarray(x) for x in L.u]), case[f'{tag}_u']) < TOL, tag
        assert rel_err(np.stack([np.asarray(x) for x in L.f]), case[f'{tag}_f']) < TOL, tag
        L.sweep.compute_residual()
        ref = float(case[f'{tag}_res_full_abs'])
        assert abs(L.status.residual - ref) <= 1e-8 * abs(ref) + 1e-11, tag
        L.sweep.compute_end_point()
        assert rel_err(L.uend.get(), case[f'{tag}_uend_0']) < TOL, tag

    check('k0')
    for k in range(1, meta['nsweeps'] + 1):
        L.sweep.update_nodes()
        check(f'k{k}')


def test_allencahn_3d_vs_oracle():
    """3-D variant (formulas of AllenCahn_MPIFFT.py; not importable as reference): device vs the NumPy oracle."""
    from oracle import sdc_oracle as O
    from pysdc_amd.problems import allencahn_imex

    for nv, it in (((16, 16, 16), 'sphere'), ((32, 32), 'circle')):
        P = allencahn_imex(nvars=nv, eps=0.08, radius=0.25, init_type=it)
        Po = O.AllenCahnND(nvars=nv, eps=0.08, radius=0.25, init_type=it)
        u = P.u_exact(0.0)
        assert rel_err(u.get(), Po.u_exact(0.0)) < 1e-14
        f, fo = P.eval_f(u, 0.0), Po.eval_f(Po.u_exact(0.0), 0.0)
        assert rel_err(f.impl.get(), fo[0]) < 1e-11
        assert rel_err(f.expl.get(), fo[1]) < 1e-13
        sol = P.solve_system(u, 1e-3, u, 0.0)
        assert rel_err(sol.get(), Po.solve_system(Po.u_exact(0.0), 1e-3, None, 0.0)) < 1e-13


def _rank_thread(world, rank, name, fname, out, errors, skip_residual=False):
    import traceback

    from tests import _fake_dist as FD

    try:
        FD.bind(world, rank)
        from pysdc_amd.controller import controller_dist
        from pysdc_amd.stats import get_sorted
        from tests.test_gpu_plugin import description_from

        case = load_cases(fname)[name]
        meta = case['meta']
        if 'iorder' in meta or meta.get('transfer'):      # multi-level description
            desc = _description(meta, meta['level_params'], meta.get('iorder', 6), meta.get('rorder', 2))
            if not meta.get('periodic', True):
                desc['space_transfer_params']['periodic'] = False
            if meta.get('transfer') == 'mesh_to_mesh_fft2d':
                from pysdc_amd.transfer import mesh_to_mesh_fft2d

                desc['space_transfer_class'], desc['space_transfer_params'] = mesh_to_mesh_fft2d, {}
        else:
            desc = description_from(meta)
        if skip_residual:
            desc['sweeper_params']['skip_residual_computation'] = ('IT_CHECK', 'IT_FINE', 'IT_DOWN', 'IT_UP', 'IT_COARSE')
        C = controller_dist(dict(logger_level=40, comm_wire='shm', **meta['controller_params']), desc, dist=FD)
        P = C.S.levels[0].prob
        u0 = P.u_init
        u0[:] = case['u0']
        uend, stats = C.run(u0, meta['t0'], meta['Tend'])
        niter = get_sorted(stats, type='niter', sortby='time')
        out[rank] = dict(uend=uend.get(), t=[t for t, _ in niter], n=[v for _, v in niter], two_hop=C.two_hop_calls, bcast=C.bcast_two_hop_calls,
                         overlap=C._overlap)
    except Exception:  # noqa: BLE001
        errors.append(traceback.format_exc())
        try:
            world.barrier.abort()
        except Exception:  # noqa: BLE001
            pass


@pytest.mark.parametrize('name,fname,size', [('fixedK_2d_P4', 'runs_relay.npz', 4), ('fixedK_2d_P3', 'runs_relay.npz', 3),
                                             ('alltodone_2d_P4', 'runs_relay.npz', 4), ('fixedK_2d_P4_tail', 'runs_relay.npz', 4),
                                             ('mssdc_P2_jac', 'runs.npz', 2), ('mssdc_P2_gs', 'runs.npz', 2),
                                             ('mssdc_P4_jac', 'runs.npz', 4), ('fixedK_3d_P2', 'runs.npz', 2),
                                             ('pfasst_heat2d_P2', 'runs_ml.npz', 2), ('pfasst_heat2d_P4', 'runs_ml.npz', 4),
                                             ('pfasst_heat2d_P4_all_to_done', 'runs_ml.npz', 4),
                                             ('pfasst_heat2d_M53_P2', 'runs_ml.npz', 2), ('pfasst_forced2d_P2', 'runs_ml.npz', 2),
                                             ('pfasst_heat3d_P2', 'runs_ml.npz', 2), ('ac2d_pfasst_P2', 'runs_ac.npz', 2),
                                             ('ac2d_fft2d_pfasst_P2', 'runs_ac_fft.npz', 2),
                                             ('t6a_pfasst_P2', 'runs_ml_dirichlet.npz', 2),
                                             ('t6a_pfasst_P4', 'runs_ml_dirichlet.npz', 4),
                                             ('forced2d_run_P2', 'runs.npz', 2), ('mssdc_P4_gs', 'runs.npz', 4),
                                             ('dirichlet_heat1d_P2', 'runs_dirichlet.npz', 2),
                                             ('skip_all_2d_P3', 'runs_skip.npz', 3), ('skip_check_2d_P2', 'runs_skip.npz', 2),
                                             ('fixedK_2d_P8', 'runs_relay8.npz', 8), ('alltodone_2d_P8', 'runs_relay8.npz', 8),
                                             ('fixedK_2d_P3_nsweeps2', 'runs_nsweeps2.npz', 3),
                                             ('fixedK_2d_P2_nsweeps2', 'runs_nsweeps2.npz', 2)])
def test_time_parallel_controller_on_device_levels(name, fname, size):
    """controller_dist with DEVICE levels and several ranks on one GPU: the ranks are threads, torch.distributed is
    replaced by an in-process stand-in (tests/_fake_dist.py), everything else - early end value, hand-over posted on
    a side stream, two-hop relay, replace_u0 with kept residual fields, deferred f[0], advance - is the code that
    runs over RCCL.  Against golden serial runs of the reference."""
    import threading

    from tests import _fake_dist as FD

    case = load_cases(fname)[name]
    world = FD.World(size)
    out, errors = {}, []
    threads = [threading.Thread(target=_rank_thread, args=(world, r, name, fname, out, errors)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[0]
    assert len(out) == size
    times = np.concatenate([out[r]['t'] for r in range(size)])
    niter = np.concatenate([out[r]['n'] for r in range(size)])
    order = np.argsort(times)
    assert list(niter[order]) == list(case['niter'])
    np.testing.assert_allclose(times[order], case['niter_t'], rtol=0, atol=1e-14)
    for r in range(size):
        assert rel_err(out[r]['uend'], case['uend']) < TOL
    if 'nsweeps2' in name:   # a lone send between the two sweeps of an iteration: no early end value under it
        assert not any(out[r]['overlap'] for r in range(size))
    elif name.startswith(('fixedK_2d', 'alltodone')):
        assert all(out[r]['overlap'] for r in range(size))
        if size > 2:
            assert all(out[r]['two_hop'] > 0 for r in range(size))
            assert all(out[r]['bcast'] > 0 for r in range(size))


@pytest.mark.parametrize('name,fname,size', [('fixedK_2d_P4', 'runs_relay.npz', 4), ('fixedK_2d_P3', 'runs_relay.npz', 3),
                                             ('fixedK_3d_P2', 'runs.npz', 2)])
def test_time_parallel_skip_residual_computation(name, fname, size):
    """runs with a fixed number of sweeps and skip_residual_computation for every stage: the sweeps only move the cached
    transforms (no residual, no kept residual fields), the hand-over still travels early - same end values as the
    reference's run."""
    import threading

    from tests import _fake_dist as FD

    case = load_cases(fname)[name]
    world = FD.World(size)
    out, errors = {}, []
    threads = [threading.Thread(target=_rank_thread, args=(world, r, name, fname, out, errors, True)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[0]
    assert len(out) == size
    times = np.concatenate([out[r]['t'] for r in range(size)])
    niter = np.concatenate([out[r]['n'] for r in range(size)])
    assert list(niter[np.argsort(times)]) == list(case['niter'])
    for r in range(size):
        assert rel_err(out[r]['uend'], case['uend']) < TOL
        assert out[r]['overlap']


@pytest.mark.parametrize('nranks,M,size,prob', [(3, 3, 64, 'heat_unforced'), (8, 5, 64, 'heat_unforced'), (4, 5, 256, 'heat_unforced'),
                                                 (3, 5, 64, 'advdiff'), (4, 3, 128, 'advection')])
def test_time_parallel_controller_64cubed_matches_serial_emulation(nranks, M, size, prob):
    """three / eight ranks on 64^3 and four on 256^3 (the fused spectral sweep kernel, norm passes, kept residual
    fields; at 256^3 the kernels run long enough for ordering mistakes between the streams to show) through
    controller_dist with the in-process stand-in, against controller_nonMPI emulating the processes; two blocks, the
    second one partially filled."""
    import os
    import threading

    from pysdc_amd.controller import controller_nonMPI

    n = int(os.environ.get('PYSDC_TP_N', size))      # (one-off runs at other sizes: PYSDC_TP_N=512 pytest -k 64cubed;
    #                                                   scripts/tp_check.py N RANKS M does the same from the shell)
    from pysdc_amd.stats import get_sorted
    from tests import _fake_dist as FD

    # (also the IMEX sweeper on advection-diffusion - complex explicit symbol, iterates stored - and implicit advection:
    # the same spectra-on-the-wire hand-over and residual update by one more field through the inverse passes)
    pp = {'heat_unforced': dict(nvars=[n, n, n], nu=0.1, freq=2), 'advdiff': dict(nvars=[n, n, n], nu=0.02, c=1.0, freq=2),
          'advection': dict(nvars=[n, n, n], c=1.0, freq=2)}[prob]
    sw = dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='IE')
    if prob == 'advdiff':
        sw['QE'] = 'EE'
    meta = dict(prob=prob, prob_params=pp, sweeper='imex_1st_order' if prob == 'advdiff' else 'generic_implicit',
                sweeper_params=sw, level_params=dict(dt=2e-3, restol=-1),
                maxiter=3, controller_params={}, t0=0.0, Tend=2e-3 * (2 * nranks - nranks // 2))
    from pysdc_amd.synth import init_field
    from tests.test_gpu_plugin import description_from

    u0h = init_field((n, n, n), 2, 1e-2, 3)
    C = controller_nonMPI(nranks, dict(logger_level=40), description_from(meta))
    P = C.MS[0].levels[0].prob
    u0 = P.u_init
    u0[:] = u0h
    ref, rstats = C.run(u0, meta['t0'], meta['Tend'])
    ref = ref.get()
    del C, P, u0
    world = FD.World(nranks)
    out, errors = {}, []

    def rank_main(rank):
        import traceback

        try:
            FD.bind(world, rank)
            from pysdc_amd.controller import controller_dist

            Cd = controller_dist(dict(logger_level=40, comm_wire='shm'), description_from(meta), dist=FD)
            Pd = Cd.S.levels[0].prob
            v = Pd.u_init
            v[:] = u0h
            uend, stats = Cd.run(v, meta['t0'], meta['Tend'])
            out[rank] = (uend.get(), Cd.two_hop_calls, Cd._overlap and Cd.spectral_wire)   # (spectra on the wire, also between blocks)
        except Exception:  # noqa: BLE001
            errors.append(traceback.format_exc())
            try:
                world.barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[0]
    for r in range(nranks):
        assert rel_err(out[r][0], ref) < 1e-12
        assert out[r][1] > 0 and out[r][2]


def test_config5_pfasst_four_ranks_match_serial_emulation():
    """BASELINE config 5 at reduced size: Allen-Cahn 3-D (64^3 / 32^3, M = 3 on both levels), two-level PFASST with
    burn-in over four time ranks - controller_dist on device levels with the in-process stand-in against
    controller_nonMPI emulating the four processes (same iteration counts, same end value)."""
    import threading

    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.problems import allencahn_imex
    from pysdc_amd.stats import get_sorted
    from pysdc_amd.sweepers import imex_1st_order
    from pysdc_amd.transfer import mesh_to_mesh
    from tests import _fake_dist as FD

    n, size = 64, 4
    desc = dict(problem_class=allencahn_imex,
                problem_params=dict(nvars=[(n, n, n), (n // 2, n // 2, n // 2)], eps=0.04, radius=0.25, init_type='sphere'),
                sweeper_class=imex_1st_order, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU', QE='EE'),
                level_params=dict(dt=1e-3, restol=1e-8, nsweeps=1), step_params=dict(maxiter=20),
                space_transfer_class=mesh_to_mesh, space_transfer_params=dict(iorder=6, rorder=2, periodic=True))
    cp = dict(logger_level=40, predict_type='pfasst_burnin')
    C = controller_nonMPI(size, dict(cp), desc)
    P = C.MS[0].levels[0].prob
    ref, rstats = C.run(P.u_exact(0.0), 0.0, 8e-3)
    ref = ref.get()
    ref_niter = [v for _, v in get_sorted(rstats, type='niter', sortby='time')]
    world = FD.World(size)
    out, errors = {}, []

    def rank_main(rank):
        import traceback

        try:
            FD.bind(world, rank)
            from pysdc_amd.controller import controller_dist

            Cd = controller_dist(dict(cp, comm_wire='shm'), desc, dist=FD)
            Pd = Cd.S.levels[0].prob
            uend, stats = Cd.run(Pd.u_exact(0.0), 0.0, 8e-3)
            niter = get_sorted(stats, type='niter', sortby='time')
            out[rank] = (uend.get(), [t for t, _ in niter], [v for _, v in niter])
        except Exception:  # noqa: BLE001
            errors.append(traceback.format_exc())
            try:
                world.barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors[0]
    times = np.concatenate([out[r][1] for r in range(size)])
    niter = np.concatenate([out[r][2] for r in range(size)])
    assert list(niter[np.argsort(times)]) == ref_niter
    for r in range(size):
        assert rel_err(out[r][0], ref) < 1e-11


@pytest.mark.parametrize('fname,name', [('runs_ml.npz', 'mlsdc_heat2d'), ('runs_ml.npz', 'pfasst_heat2d_P4'),
                                        ('runs_ml.npz', 'pfasst_forced2d_P2'), ('runs_ml.npz', 'mlsdc_heat2d_M53'),
                                        ('runs_ml.npz', 'pfasst_heat3d_P2'), ('runs_ac.npz', 'ac2d_pfasst_P2')])
def test_multilevel_runs_inside_foreign_levels(fname, name):
    """MLSDC / PFASST with level containers that are NOT pysdc_amd.level.Level (tests/_plain_level.py: plain lists that
    reset_level replaces, frozen attribute set - the semantics of the reference's core/level.py): the sweepers keep a
    ForeignLevelState per level and adopt the lists the FAS transfer writes into (u, f, tau, uold, fold)."""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.level import ForeignLevelState
    from pysdc_amd.stats import get_sorted
    from tests._plain_level import PlainStep, PlainLevel

    case = load_cases(fname)[name]
    meta = case['meta']
    desc = _description(meta, meta['level_params'], meta.get('iorder', 6), meta.get('rorder', 2))
    desc['step_class'] = PlainStep
    C = controller_nonMPI(meta['num_procs'], dict(logger_level=40, **meta['controller_params']), desc)
    assert all(type(L) is PlainLevel for L in C.MS[0].levels)
    P = C.MS[0].levels[0].prob
    u0 = P.u_init
    u0[:] = case['u0']
    uend, stats = C.run(u0, meta['t0'], meta['Tend'])
    assert all(isinstance(L.sweep._dev(), ForeignLevelState) for L in C.MS[0].levels)
    niter = get_sorted(stats, type='niter', sortby='time')
    assert [v for _, v in niter] == list(case['niter'])
    assert rel_err(uend.get(), case['uend']) < TOL
