"""GPU: the RCCL transport behind the C-ABI (include/sdcmi.h sdc_comm_*, pysdc_amd/comm.py).  On the one-GPU box a
communicator of ONE rank carries the hand-over to itself (ncclSend / ncclRecv to the own rank inside one group): that
drives librccl binding, communicator setup, the message stream, the event ordering against the engine's stream, the
inbox -> sdc_replace_u0 path and the UEND write fence on real hardware.  The two-rank test needs two GPUs (it is
skipped on the one-GPU box and runs wherever the suite is given a multi-GPU node)."""
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
    from pysdc_amd.comm import RcclComm

    e = _engine()
    comm = RcclComm(e, 1, 0)
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
    from pysdc_amd.comm import RcclComm, torch_host_bcast

    e = _engine()
    comm = RcclComm(e, 2, rank, host_bcast=torch_host_bcast)
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
