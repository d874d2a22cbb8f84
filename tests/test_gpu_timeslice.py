"""GPU: what a time-parallel level does with a u[0] that is replaced between sweeps (controller_MPI.py:218-233, :574-583;
include/sdcmi.h: sdc_set_timeslice_options) - iterates recomputed from the start values received so far (the trail) instead
of stored, the last inverse pass of a residual put off until the new start value is there (one pass then yields the node norms
before and after the receive), the last node's spectrum written first.  Every combination must reproduce the plain data flow
(iterates stored, every pass at once): node values and end values to round-off of the multipliers, residual norms of both
kinds, the records of queued residuals, the end value of a multi-rank run against the serial emulation."""
import os

import numpy as np
import pytest

from pysdc_amd import lib as L
from tests import _gpu as G

pytestmark = pytest.mark.gpu


def _engine(nvars, M, opts):
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    e = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1), M)
    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    qi = np.zeros_like(c.Qmat)
    qi[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
    e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
    assert e.spectral_handover_ok()
    L.check(e.lib.sdc_set_wire_spectral(e.ctx, 1), e.ctx)
    e.set_early_end_point(True)
    e.set_timeslice_options(*opts)
    return e


def _slice(nvars, M, opts, iters, dt, blocking=False, receive=None):
    """one time slice: predict, then per iteration sweep -> end value (spectrum) -> residual -> the received start value (here:
    the slice's own end value, or nothing when receive[k] is False) -> residual.  Returns everything an observer can see."""
    import torch

    from pysdc_amd.hip_mesh import _CAI
    from pysdc_amd.synth import init_field

    e = _engine(nvars, M, opts)
    e.upload(L.SLOT_U, 0, init_field(nvars, 2, 1e-2, 3))
    nspec = 2 * (nvars[0] // 2 + 1) * int(np.prod(nvars[1:]))
    e.predict(0.0, dt)
    e.profile_read()
    e.profile_enable(True)
    out = dict(resA=[], resB=[], sent=[])
    r0, n0 = e.residual(dt)
    out['res0'] = (r0, n0)
    futs = []
    for k in range(iters):
        e.sweep(0.0, dt)
        e.end_point(dt, False)
        src = torch.as_tensor(_CAI(e.end_spectrum(), nspec, e), device='cuda')
        out['sent'].append(src.cpu().numpy().copy())
        if blocking:
            out['resA'].append(e.residual(dt))
        else:
            futs.append(('resA', e.residual_post(dt)))
        if receive is None or receive[k]:
            dst = torch.as_tensor(_CAI(e.spectrum_inbox(), nspec, e), device='cuda')
            e.invalidate_spectra(8)     # (what the communicator does: the end value existed for the wire only)
            dst.copy_(src)
            torch.cuda.synchronize()    # (the copy runs on torch's stream, the engine on its own)
            e.replace_u0_spectrum()
        if blocking:
            out['resB'].append(e.residual(dt))
        else:
            futs.append(('resB', e.residual_post(dt)))
    for key, f in futs:
        out[key].append((f.result(), f.norms.copy()))
    out['launches'] = {k: v[1] for k, v in e.profile_read().items()}
    e.profile_enable(False)
    out['bytes_sweeping'] = e.device_bytes    # (before anybody looks at a node value)
    out['u'] = e.download_u()
    e.end_point(dt, False)
    out['uend'] = e.download(L.SLOT_UEND)
    out['bytes'] = e.device_bytes
    e.close()
    return out


def _same(a, b, scale, what):
    assert len(a) == len(b), what
    for k, ((ra, na), (rb, nb)) in enumerate(zip(a, b)):
        tol = 1e-6 * abs(rb) + 2e-12 * scale
        assert abs(ra - rb) <= tol, (what, k, ra, rb)
        assert np.all(np.abs(na - nb) <= 1e-6 * np.abs(nb) + 2e-12 * scale), (what, k, na, nb)


# (trail_sources, defer_last_pass: 0 at once / 1 put off, behind the next sweep where possible / 2 put off only, split_send)
PLAIN = (0, 0, False)
FLOWS = [(0, 1, False), (0, 2, False), (5, 1, False), (5, 1, True), (5, 0, False), (5, 2, True), (2, 1, False), (2, 1, True), (8, 1, True)]


# (1024^2 x 3 and 512^3 x 3: the trail launch has a wave to spare and takes the difference of the last two start values along as
# one more line - SpecArgs::dz; 512 x 5 fills its workgroup: the difference goes through a launch of its own)
@pytest.mark.parametrize('nvars,M', [((512, 512), 5), ((1024, 1024), 3), ((512, 512, 512), 5), ((512, 512, 512), 3),
                                     ((512, 512), 1), ((512, 512), 2), ((1024, 1024), 4)])   # (every node count the launches are built for)
def test_every_data_flow_of_a_time_slice_reproduces_stored_iterates(nvars, M):
    dt = 2e-3 * (512.0 / nvars[0]) ** 2 * 40
    iters = 4 if len(nvars) == 3 else 6
    ref = _slice(nvars, M, PLAIN, iters, dt, blocking=True)
    scale = float(np.max(np.abs(ref['u'])))
    flows = FLOWS[:3] if len(nvars) == 3 else FLOWS
    for opts in flows:
        for blocking in ((False,) if len(nvars) == 3 else (False, True)):
            got = _slice(nvars, M, opts, iters, dt, blocking=blocking)
            tag = (opts, blocking)
            assert np.max(np.abs(got['u'] - ref['u'])) <= 1e-12 * scale, tag
            assert np.max(np.abs(got['uend'] - ref['uend'])) <= 1e-12 * scale, tag
            for k in range(iters):
                assert np.max(np.abs(got['sent'][k] - ref['sent'][k])) <= 1e-12 * np.max(np.abs(ref['sent'][k])), (tag, k)
            _same(got['resA'], ref['resA'], scale, (tag, 'before the receive'))
            _same(got['resB'], ref['resB'], scale, (tag, 'after the receive'))
            assert got['res0'][0] == ref['res0'][0]
            if M == 3 and opts[0] >= 4 and opts[1] == 1 and not blocking:
                # the difference of two start values went through the next sweep's trail launch, not through a launch of its
                # own, wherever a trail launch followed (sweeps 2 .. 4 of the slice)
                assert got['launches'].get('fft_z_diff[1]', 0) <= iters - 3, got['launches']
    # (the residual does shrink: the comparison above is not one of zeros)
    assert ref['resB'][-1][0] < 0.5 * ref['resB'][0][0]


def test_a_slice_that_does_not_receive_every_time():
    """first rank of a block (never receives), a predecessor that is done (controller_MPI.py:247: no receive from then on),
    and a mix: the trail only grows when a start value arrives"""
    nvars, M, dt, iters = (512, 512), 5, 0.08, 6
    for receive in ([False] * iters, [True, True, False, False, True, False], [False, True, True, True, True, True]):
        ref = _slice(nvars, M, PLAIN, iters, dt, blocking=True, receive=receive)
        scale = float(np.max(np.abs(ref['u'])))
        for opts in ((0, 1, False), (5, 1, False), (5, 1, True), (3, 2, False)):
            got = _slice(nvars, M, opts, iters, dt, receive=receive)
            assert np.max(np.abs(got['u'] - ref['u'])) <= 1e-12 * scale, (receive, opts)
            _same(got['resA'], ref['resA'], scale, (receive, opts, 'A'))
            _same(got['resB'], ref['resB'], scale, (receive, opts, 'B'))


def test_iterates_that_are_not_stored_leave_the_node_slabs_unmapped():
    """a slice on the trail keeps start values, not iterates: no node spectra, no U[1..M], no F (mapped on first touch) - until
    somebody looks at a node value"""
    nvars, M = (512, 512, 512), 5
    os.environ['SDC_LAZY_MIN_BYTES'] = '1048576'
    try:
        lean = _slice(nvars, M, (5, 2, False), 3, 0.08)
        fat = _slice(nvars, M, PLAIN, 3, 0.08, blocking=True)
    finally:
        del os.environ['SDC_LAZY_MIN_BYTES']
    field = 8 * int(np.prod(nvars))
    spec = field * (nvars[0] + 2) / nvars[0]
    # while it sweeps, the lean slice holds: U[0], one end-value buffer, M work spectra, the start values of the trail (3),
    # the last node's spectrum, the inbox, two spare spectra of the joint pass - no U[1..M], no F, no node spectra
    assert lean["bytes_sweeping"] <= 2.05 * field + (M + 9) * spec, lean['bytes_sweeping'] / field
    assert lean['bytes'] >= lean['bytes_sweeping'] + (2 * M + 1) * field      # the download mapped U[1..M] and F
    # the plain flow stores its iterates: M - 1 node spectra more, the trail's extra start values and spare buffers less
    assert fat['bytes_sweeping'] >= 2 * field + (2 * M + 2) * spec, fat['bytes_sweeping'] / field


@pytest.mark.parametrize('env', [{}, {'PYSDC_AMD_TRAIL': '0', 'PYSDC_AMD_SPLIT_SEND': '0'}, {'PYSDC_AMD_TRAIL': '2', 'PYSDC_AMD_DEFER_X': '2'},
                                 {'PYSDC_AMD_TRAIL': '0', 'PYSDC_AMD_SPLIT_SEND': '0', 'PYSDC_AMD_DEFER_X': '0'},
                                 {'PYSDC_AMD_TRAIL': '5', 'PYSDC_AMD_SPLIT_SEND': '0', 'PYSDC_AMD_DEFER_X': '1'},   # (eight ranks' default)
                                 {'PYSDC_AMD_DEFER_X': '1'}])
def test_three_ranks_at_512cubed_match_the_serial_emulation(env):
    """controller_dist, three thread ranks on the one GPU over the shared-memory wire, heat 512^3 (the size from which iterates
    are recomputed from mode pairs), six iterations per block (more start values than a trail of two holds), two blocks with
    the second one partially filled - against controller_nonMPI emulating the ranks"""
    import threading
    import traceback

    import torch

    from pysdc_amd.controller import controller_dist, controller_nonMPI
    from pysdc_amd.synth import init_field
    from tests import _fake_dist as FD
    from tests._cases import rel_err
    from tests.test_gpu_plugin import description_from

    n, nranks, M, dt = 512, 3, 5, 2e-3
    meta = dict(prob='heat_unforced', prob_params=dict(nvars=[n, n, n], nu=0.1, freq=2), sweeper='generic_implicit',
                sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='IE'), level_params=dict(dt=dt, restol=-1),
                maxiter=6, controller_params={}, t0=0.0, Tend=dt * (2 * nranks - 1))
    u0h = init_field((n, n, n), 2, 1e-2, 3)
    C = controller_nonMPI(nranks, dict(logger_level=40), description_from(meta))
    u0 = C.MS[0].levels[0].prob.u_init
    u0[:] = u0h
    ref, rstats = C.run(u0, meta['t0'], meta['Tend'])
    ref = ref.get()

    def by_time_and_iter(stats, kind):
        return {(round(k.time / dt), k.iter): float(v) for k, v in stats.items() if k.type == kind}

    ref_res = {kind: by_time_and_iter(rstats, kind) for kind in ('residual_post_sweep', 'residual_post_iteration')}
    del C, u0
    import gc

    gc.collect()                # (three 512^3 levels of the emulation: their engines go before the ranks' ones come)
    torch.cuda.empty_cache()
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    world = FD.World(nranks)
    out, errors = {}, []

    def rank_main(rank):
        try:
            FD.bind(world, rank)
            Cd = controller_dist(dict(logger_level=40, comm_wire='shm'), description_from(meta), dist=FD)
            v = Cd.S.levels[0].prob.u_init
            v[:] = u0h
            uend, stats = Cd.run(v, meta['t0'], meta['Tend'])
            res = {kind: by_time_and_iter(stats, kind) for kind in ('residual_post_sweep', 'residual_post_iteration')}
            out[rank] = (uend.get(), Cd.spectral_wire, res)
            Cd.close()
        except Exception:  # noqa: BLE001
            errors.append(traceback.format_exc())
            try:
                world.barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    try:
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    gc.collect()
    assert not errors, errors[0]
    for r in range(nranks):
        assert rel_err(out[r][0], ref) < 1e-12, r
        assert out[r][1]
        # the residuals the hooks logged - after every sweep (against the old start value) and after every receive - are the
        # serial emulation's, entry by entry
        for kind, mine in out[r][2].items():
            assert len(mine) > 0
            for key, v in mine.items():
                want = ref_res[kind][key]
                assert abs(v - want) <= 1e-6 * abs(want) + 1e-13, (r, kind, key, v, want)


def test_a_serial_run_through_the_plug_in_classes_never_makes_the_node_fields_real():
    """controller -> sweeper -> problem classes, hooks that fetch (not read) level fields: a run that stays in Fourier space holds
    u[0], f[0], two end-value buffers, the work spectra and the spectra of the start value and of the last node - not U[1..M],
    F[1..M] or the node spectra (86 instead of 232 GB at 1024^3, M = 5); reading one node value allocates them"""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.hooks import Hooks
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    class Fetch(Hooks):
        def post_iteration(self, step, level_number):
            L_ = step.levels[level_number]
            assert len(L_.u) == 4 and L_.u[2] is not None and L_.f[3] is not None      # handed out, not dereferenced

    n, M = 128, 3
    os.environ['SDC_LAZY_MIN_BYTES'] = '4096'
    try:
        desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=2),
                    sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='IE'),
                    level_params=dict(dt=1e-2, restol=-1), step_params=dict(maxiter=4))
        C = controller_nonMPI(1, dict(logger_level=40, hook_class=[Fetch]), desc)
        Lv = C.MS[0].levels[0]
        uend, _ = C.run(Lv.prob.u_exact(0.0), 0.0, 3e-2)
        field = 8 * n**3
        spec = field * (n + 2) / n
        lean = Lv.engine.device_bytes
        assert lean <= 4.1 * field + (M + 3) * spec, lean / field
        assert np.all(np.isfinite(uend.get()))
        _ = Lv.u[2].get()                      # somebody looks at a node value
        assert Lv.engine.device_bytes >= lean + 2 * M * field
    finally:
        del os.environ['SDC_LAZY_MIN_BYTES']
