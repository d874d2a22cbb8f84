"""Index arithmetic of controller_dist's mesh-aware messages for up to 8 ranks (the gloo tests run 2-4 processes):
the two-hop hand-over uend(r) -> u[0](r+1) and the scatter + all-gather broadcast move exactly the bytes of the
direct copy for every message length, including lengths that do not divide by the number of pieces and lengths
shorter than the number of ranks."""
import threading
import types

import numpy as np
import pytest
import torch

from tests import _queue_dist as QD


class _Buf:
    def __init__(self, t):
        self.t = t

    def as_torch(self):
        return self.t


def _bare_controller(rank, size):
    from pysdc_amd.controller import controller_dist

    C = object.__new__(controller_dist)
    C.dist, C.comm, C.rank, C.size, C.relay = QD, None, rank, size, True
    C._two_hop_calls = C._bcast_two_hop_calls = 0
    C._comms, C._abi = None, False
    C._relay_stage = None
    C.S = types.SimpleNamespace(status=types.SimpleNamespace(iter=3))
    return C


def _run(size, fn):
    world, errors = QD.World(size), []

    def main(rank):
        try:
            QD.bind(world, rank)
            fn(rank)
        except Exception:  # noqa: BLE001
            import traceback

            errors.append(traceback.format_exc())

    threads = [threading.Thread(target=main, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert not errors, errors[0]


@pytest.mark.parametrize('size', [3, 4, 5, 8])
@pytest.mark.parametrize('n', [1, 2, 5, 7, 8, 64, 257, 1000])
def test_broadcast_over_the_mesh_moves_every_element(size, n):
    for root in sorted({0, size - 1, size // 2}):
        data = torch.arange(n, dtype=torch.float64) + 1000.0 * root + 0.5
        got = {}

        def rank_main(rank, root=root, data=data):
            C = _bare_controller(rank, size)
            buf = data.clone() if rank == root else torch.full((n,), -1.0, dtype=torch.float64)
            C.broadcast(_Buf(buf.reshape(1, n) if n > 1 else buf), root)
            got[rank] = (buf, C.bcast_two_hop_calls)

        _run(size, rank_main)
        for rank in range(size):
            assert torch.equal(got[rank][0], data), (size, n, root, rank)
            assert got[rank][1] == 1


@pytest.mark.parametrize('size', [3, 4, 6, 8])
@pytest.mark.parametrize('n', [1, 3, 7, 8, 9, 100, 1023])
def test_two_hop_hand_over_is_the_direct_copy(size, n):
    src = [torch.from_numpy(np.random.default_rng(10 * size + r).standard_normal(n)) for r in range(size)]
    got = {}

    def rank_main(rank):
        C = _bare_controller(rank, size)
        inbox = torch.full((n,), -7.0, dtype=torch.float64)
        L = types.SimpleNamespace(uend=_Buf(src[rank].clone()), u=[_Buf(inbox)])
        for req in C._two_hop_ops(L, size, _Buf(inbox) if rank >= 1 else None):
            req.wait()
        got[rank] = (inbox, L.uend.t)

    _run(size, rank_main)
    for rank in range(size):
        assert torch.equal(got[rank][1], src[rank])                    # the end value itself is untouched
        if rank >= 1:
            assert torch.equal(got[rank][0], src[rank - 1]), (size, n, rank)
