"""GPU: the transport behind the C-ABI (include/sdcmi.h sdc_comm_*, pysdc_amd/comm.py).  On the one-GPU box a
communicator of ONE rank carries the hand-over to itself over RCCL (ncclSend / ncclRecv to the own rank inside one
group): that drives librccl binding, communicator setup, the message stream, the event ordering against the engine's
stream, the inbox -> sdc_replace_u0 path and the UEND write fence on real hardware.  Several ranks on the one GPU - as
threads and as separate processes - run the same calls over the shared-memory wire: direct and two-hop hand-over, mesh
broadcast, levels that share a communicator, and `bench.py --gpus 2` from its self-launching parent down to the JSON
line.  The two-rank RCCL test needs two GPUs (skipped on the one-GPU box, runs wherever the suite is given more)."""
import numpy as np
import pytest

from pysdc_amd import lib as L
from tests import _gpu as G

pytestmark = pytest.mark.gpu


def _engine(n=32, M=3):
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=0.1), M)
    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    qi = np.zeros_like(c.Qmat)
    qi[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
    e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
    return e


@pytest.mark.parametrize('chunk', [0, 1000])
def test_single_rank_handover_through_rccl(chunk):
    from pysdc_amd.comm import DeviceComm

    e = _engine()
    comm = DeviceComm(e, 1, 0)
    assert comm.info()['wire'] == 'rccl'
    if chunk:
        comm.set_chunk(chunk)
    rng = np.random.default_rng(1)
    u0 = rng.standard_normal(e.nvars)
    dt = 1e-3
    e.upload(L.SLOT_U, 0, u0)
    e.predict(0.0, dt)
    for k in range(3):
        e.sweep(0.0, dt)
        e.residual(dt)
        e.end_point(dt, False)
        uend = e.download(L.SLOT_UEND)
        comm.exchange(send_to=0, recv_from=0)          # the own end value becomes the new u[0]
        assert np.array_equal(e.download(L.SLOT_U, 0), uend), k
        # the residual against the new u[0] equals a fresh evaluation on an engine that was given it by upload
        res, _ = e.residual(dt)
        ref = _engine()
        ref.upload(L.SLOT_U, 0, uend)
        for m in range(1, e.M + 1):
            ref.upload(L.SLOT_U, m, e.download(L.SLOT_U, m))
            ref.upload(L.SLOT_F, m, e.download(L.SLOT_F, m))
        rref, _ = ref.residual(dt)
        assert abs(res - rref) <= 1e-9 * abs(rref) + 1e-13, (k, res, rref)
        ref.close()
    # broadcast of a slab field with one rank is the identity and leaves the engine consistent
    comm.bcast(L.SLOT_UEND, 0, root=0)
    comm.sync()
    comm.close()
    e.close()


def _two_rank_worker(rank, port, out):
    import os

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    from pysdc_amd.comm import DeviceComm, torch_host_bcast

    e = _engine()
    comm = DeviceComm(e, 2, rank, host_bcast=torch_host_bcast)
    val = np.full(e.nvars, float(rank + 1))
    e.upload(L.SLOT_UEND, 0, val)
    e.upload(L.SLOT_U, 0, np.zeros(e.nvars))
    comm.exchange(send_to=1 if rank == 0 else None, recv_from=0 if rank == 1 else None)
    comm.bcast(L.SLOT_UEND, 0, root=1)
    out.put((rank, float(e.download(L.SLOT_U, 0).max()), float(e.download(L.SLOT_UEND).max())))
    comm.close()
    dist.destroy_process_group()


def test_two_ranks_over_xgmi():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    import torch.multiprocessing as mp

    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 29611, out)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(out.get(timeout=180) for _ in range(2))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():   # never leave a rank behind that may hold the GPUs
                p.kill()
    assert res == [(0, 0.0, 2.0), (1, 1.0, 2.0)]


def _thread_ranks(P, body):
    """P ranks as threads of this process on the one GPU; body(rank, uid) -> result"""
    import threading
    import traceback

    from pysdc_amd.comm import shm_unique_id

    uid = shm_unique_id()
    out, errors = [None] * P, []

    def run(r):
        try:
            out[r] = body(r, uid)
        except Exception:  # noqa: BLE001
            errors.append(traceback.format_exc())

    ts = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(240)
    assert not errors, errors[0]
    return out


@pytest.mark.parametrize('P,relay,active', [(2, True, 2), (3, True, 3), (4, True, 4), (4, False, 4), (5, True, 3), (8, True, 8)])
def test_lockstep_handover_and_broadcast_between_ranks_on_one_device(P, relay, active):
    """uend(r) -> u[0](r + 1) for all active ranks at once (sdc_comm_handover_post / _complete: direct for two ranks, two
    hops through all ranks beyond), three rounds over the same mailboxes, then the end-of-block broadcast from the last
    active rank (mesh scatter + all-gather beyond two ranks): every value bit for bit, counters as expected"""
    from pysdc_amd.comm import DeviceComm

    n = 24  # 24^3 = 13824 values: not a multiple of 5 or 7, so pieces of unequal length occur

    def field(r, k):
        return np.random.default_rng(100 * k + r).standard_normal((n, n, n))

    def body(r, uid):
        e = _engine(n=n)
        comm = DeviceComm(e, P, r, uid=uid)
        comm.set_relay(relay)
        assert comm.info()['wire'] == 'shm'
        got = []
        for k in range(3):
            e.upload(L.SLOT_UEND, 0, field(r, k))
            e.upload(L.SLOT_U, 0, np.zeros((n, n, n)))
            comm.handover_post(active)
            comm.handover_complete()
            got.append(e.download(L.SLOT_U, 0))
        e.upload(L.SLOT_UEND, 0, field(r, 9))
        comm.bcast(L.SLOT_UEND, 0, root=active - 1)
        end = e.download(L.SLOT_UEND)
        info = comm.info()
        comm.sync()
        comm.close()
        e.close()
        return got, end, info

    res = _thread_ranks(P, body)
    for r, (got, end, info) in enumerate(res):
        for k in range(3):
            want = field(r - 1, k) if 1 <= r < active else np.zeros((n, n, n))
            assert np.array_equal(got[k], want), (r, k)
        assert np.array_equal(end, field(active - 1, 9)), r
        assert info['two_hop_handovers'] == (3 if relay and active > 2 and r < active else 0)
        assert info['mesh_broadcasts'] == (1 if relay and P > 2 else 0)


@pytest.mark.parametrize('share,n', [(0.45, 40), (0.9, 24), (0.3, 96)])
def test_two_ranks_share_the_message_between_the_wire_and_host_memory(share, n):
    """two ranks: the tail of every lock-step hand-over travels through pinned host memory on helper threads (HostPipe: a ring
    of slots in shared memory) beside the direct message - several rounds over the same pipe, every value bit for bit; the
    96^3 field is larger than the ring, so slots are reused within one message"""
    import os

    from pysdc_amd.comm import DeviceComm

    os.environ['SDC_PIPE_CHUNK'] = '4096' if n == 96 else '1000000'     # (slots of 32 KB: ~65 trips round the ring per message)
    os.environ['SDC_PIPE_MIN_BYTES'] = '0'                              # (the path is for messages of tens of MB by default)

    def field(r, k):
        return np.random.default_rng(10 * k + r).standard_normal((n, n, n))

    def body(r, uid):
        e = _engine(n=n)
        comm = DeviceComm(e, 2, r, uid=uid)
        comm.set_host_share(share)
        got = []
        for k in range(4):
            e.upload(L.SLOT_UEND, 0, field(r, k))
            e.upload(L.SLOT_U, 0, np.zeros((n, n, n)))
            comm.handover_post(2)
            comm.handover_complete()
            got.append(e.download(L.SLOT_U, 0))
        comm.sync()
        comm.close()
        e.close()
        return got

    try:
        res = _thread_ranks(2, body)
    finally:
        del os.environ['SDC_PIPE_CHUNK']
        del os.environ['SDC_PIPE_MIN_BYTES']
    for k in range(4):
        assert np.array_equal(res[0][k], np.zeros((n, n, n))) and np.array_equal(res[1][k], field(0, k)), k


def test_levels_of_one_rank_share_a_communicator():
    """two levels per rank (fine owns, coarse attaches: PFASST sends on every level, controller_MPI.py:702-768): messages of
    both levels travel over the one communicator in the order they are posted"""
    from pysdc_amd.comm import DeviceComm

    def body(r, uid):
        fine, coarse = _engine(n=32), _engine(n=16)
        cf = DeviceComm(fine, 2, r, uid=uid)
        cc = DeviceComm.attach(coarse, cf)
        for e, v in ((fine, 1.0 + r), (coarse, 10.0 + r)):
            e.upload(L.SLOT_UEND, 0, np.full(e.nvars, v))
            e.upload(L.SLOT_U, 0, np.zeros(e.nvars))
        to, frm = (1, None) if r == 0 else (None, 0)
        cf.exchange(send_to=to, recv_from=frm)      # fine first ...
        cc.exchange(send_to=to, recv_from=frm)      # ... then coarse: a lone send each on rank 0, both in flight together
        out = float(fine.download(L.SLOT_U, 0).max()), float(coarse.download(L.SLOT_U, 0).max())
        cc.close()
        cf.close()
        fine.close()
        coarse.close()
        return out

    assert _thread_ranks(2, body) == [(0.0, 0.0), (1.0, 10.0)]


@pytest.mark.parametrize('ranks,n,steps,warmup', [(2, 256, 2, 1), (8, 256, 1, 1)])
def test_bench_ranks_on_one_gpu_end_to_end(tmp_path, ranks, n, steps, warmup):
    """`python bench.py --gpus 2` as the driver starts it (no launcher, no environment): the parent starts the rank
    processes, they rendezvous over gloo, the state vectors travel through the C-ABI communicator (shared-memory wire,
    both ranks on GPU 0), rank 0's JSON line comes back.  The end value equals controller_nonMPI emulating the two
    ranks; every rank reports its own iterations."""
    import json
    import os
    import subprocess
    import sys

    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit
    from pysdc_amd.stats import get_sorted
    import ctypes as C

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / 'uend.npy')
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(ranks), '--n', str(n), '--backend', 'gloo', '--same-device',
           '--steps', str(steps), '--warmup', str(warmup), '--dump-end-value', dump, '--job-timeout', '600']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=root)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert res.returncode == 0 and lines, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    rec = json.loads(lines[-1])
    assert 'error' not in rec, rec
    assert rec['n_gpus'] == ranks and rec['steps'] == steps and rec['scaling'] == 'weak'
    assert rec['value'] == pytest.approx(ranks * steps / (rec['ms_per_step'] * 1e-3 * steps), rel=1e-3)   # (the line rounds)
    assert len(lines[-1]) < 6000
    pr = rec['per_rank']     # (compact: one list per quantity, one entry per rank; the full per-rank records are in the side file)
    assert len(pr['seconds']) == ranks and pr['sweeps'] == [4 * steps] * ranks and pr['wire'] == 'shm'
    assert rec['niter'] == 4 and rec['finite']
    # the multi-rank path was checked against the serial emulation on a small grid before the run, in its default mode
    assert 'mode: default (64^3 and 512^2 checks vs serial emulation' in rec['config']['time_parallel'], rec['config']
    # the same time steps by the serial controller emulating the ranks
    dt = 1e-3 * (512.0 / n) ** 2
    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=2, order=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=5, quad_type='RADAU-RIGHT', QI='IE'),
                level_params=dict(dt=dt, restol=-1.0, nsweeps=1), step_params=dict(maxiter=4))
    ctrl = controller_nonMPI(ranks, dict(logger_level=40), desc)
    lvl = ctrl.MS[0].levels[0]
    u0 = lvl.prob.u_init
    L.check(lvl.engine.lib.sdc_init_field(lvl.engine.ctx, u0.ptr, (C.c_int * 3)(2, 2, 2), 1e-3, 0), lvl.engine.ctx)
    ref, stats = ctrl.run(u0, 0.0, ranks * dt * (warmup + steps))
    assert [v for _, v in get_sorted(stats, type='niter')] == [4] * (ranks * (warmup + steps))
    ref, got = ref.get(), np.load(dump)
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))


def test_bench_config5_pfasst_on_eight_ranks_is_the_golden_run(tmp_path):
    """BASELINE config 5 verbatim through `bench.py --workload allencahn --gpus 8` (eight rank processes on GPU 0 over the
    shared-memory wire: launcher, rendezvous, the two-level wire validation against the serial emulation, controller_dist's
    PFASST with burn-in - fine and coarse hand-overs, the `done` chain): with the reference's 2-D reaction term and circle
    extruded along z (--ac-variant ref2d) and restol 1e-8 the run IS the golden 8-process PFASST case of the reference
    (tests/golden/runs_cfg5.npz: cfg5_ac2d_pfasst_P8) - iteration counts of every rank bit-exact, the end value in every
    z-plane <= 1e-10 (what controller_nonMPI(8) reproduces in tests/test_gpu_configs.py)"""
    import json
    import os
    import subprocess
    import sys

    from tests._cases import load_cases

    import torch

    if torch.cuda.mem_get_info()[0] < 60e9:
        pytest.skip('not enough HBM for eight two-level slices')
    case = load_cases('runs_cfg5.npz')['cfg5_ac2d_pfasst_P8']
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump, start = str(tmp_path / 'uend.npy'), str(tmp_path / 'u0.npy')
    np.save(start, case['u0'])        # (the reference's circle with the seeded perturbation the golden run started from)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--workload', 'allencahn', '--ac-variant', 'ref2d', '--gpus', '8',
           '--backend', 'gloo', '--same-device', '--restol', '1e-8', '--steps', '1', '--warmup', '0', '--dump-end-value', dump,
           '--start-plane-file', start, '--job-timeout', '900']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200, env=env, cwd=root)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert res.returncode == 0 and lines, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    rec = json.loads(lines[-1])
    assert 'error' not in rec, rec
    assert rec['n_gpus'] == 8 and rec['finite'] and 'two-level PFASST' in rec['config']['time_parallel']
    assert 'mode: default (64^3 / 32^3 two-level checks vs serial emulation' in rec['config']['time_parallel'], rec['config']
    assert rec['per_rank']['sweeps'] == [int(v) for v in case['niter']] == [7, 7, 8, 8, 8, 9, 9, 9]
    got = np.load(dump)
    scale = float(np.max(np.abs(case['uend'])))
    assert got.shape == (256, 256, 256) and float(np.max(np.abs(got - case['uend'][:, :, None]))) < 1e-10 * scale


@pytest.mark.parametrize('forced', [1, 2, 3])
def test_wire_validation_falls_back_mode_by_mode(forced):
    """bench.py's validate_wire drops a wire mode that does not reproduce the serial emulation for the next, more conservative
    one: with the first `forced` modes declared mismatching, the run lands on the following mode (fields on the wire and every
    hand-over sent; then no two-hop relay either), says so in its line - and ends with ONE JSON error when no mode is left"""
    import json
    import os
    import subprocess
    import sys

    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '3', '--n', '64', '--backend', 'gloo', '--same-device', '--steps', '1',
           '--warmup', '1', '--job-timeout', '600']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['PYSDC_BENCH_FORCE_WIRE_MISMATCH'] = str(forced)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=root)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert lines, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    rec = json.loads(lines[-1])
    if forced >= len(bench.WIRE_MODES):
        # (the launcher reports the rank that gave up; the rank's own line - why - is in its output tail)
        assert res.returncode != 0 and 'error' in rec and 'no wire mode reproduces the serial emulation' in json.dumps(rec), rec
        return
    assert res.returncode == 0 and 'error' not in rec, rec
    assert f'mode: {bench.WIRE_MODES[forced][0]}' in rec['config']['time_parallel'], rec['config']
    assert rec['n_gpus'] == 3 and rec['finite'] and rec['niter'] == 4


def test_bench_reports_a_failed_launch_as_json(tmp_path):
    """a multi-rank job that cannot start (two ranks on one GPU over RCCL is refused before anything runs) ends with ONE
    JSON error line and a non-zero exit code - no hang, no partial output"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--n', '64', '--same-device'],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, cwd=root)
    assert res.returncode != 0
    assert 'error' in json.loads(res.stdout.strip().splitlines()[-1])
