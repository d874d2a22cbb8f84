"""RCCL transport of the time-parallel hand-over through the C-ABI (include/sdcmi.h: sdc_comm_*).

The role of the reference's ``NCCLComm`` (pySDC/helpers/NCCL_communicator.py:7-20: an MPI communicator whose data
calls go to NCCL) for the one exchange this path has: rank 0 makes the unique id, a HOST-side communicator ships its
128 bytes to the other ranks (``host_bcast``; mpi4py ``comm.bcast`` in the reference, ``torch.distributed`` over gloo
here), every rank then joins with ``ncclCommInitRank`` on its own GPU.  Messages are posted on a stream of the
engine's own and ordered against its kernels by events; nothing here blocks the host."""
import ctypes as C

from pysdc_amd import lib as L
from pysdc_amd.errors import ParameterError


def unique_id():
    """128 bytes identifying a new communicator; call on ONE rank and ship the result to the others"""
    buf = C.create_string_buffer(128)
    L.check(L.load().sdc_comm_unique_id(buf), None)
    return buf.raw


def torch_host_bcast(uid, root=0, group=None):
    """ship the id with torch.distributed (any backend that moves host objects, e.g. gloo)"""
    import torch.distributed as dist

    box = [uid]
    dist.broadcast_object_list(box, src=root, group=group)
    return box[0]


class RcclComm:
    """one time rank's end of the communicator, bound to that rank's SweepEngine (fine level)"""

    def __init__(self, engine, nranks, rank, uid=None, host_bcast=None):
        if uid is None:
            if nranks > 1 and host_bcast is None:
                raise ParameterError('more than one rank: pass the unique id or a host_bcast(uid_or_None) -> uid callable')
            uid = unique_id() if rank == 0 else None
            if host_bcast is not None:
                uid = host_bcast(uid)
        if len(uid) != 128:
            raise ParameterError('the unique id has 128 bytes')
        self.engine, self.size, self.rank = engine, int(nranks), int(rank)
        L.check(engine.lib.sdc_comm_init(engine.ctx, uid, self.size, self.rank), engine.ctx)

    def _e(self, rc):
        L.check(rc, self.engine.ctx)

    def exchange(self, send_to=None, recv_from=None):
        """send UEND to ``send_to`` and / or receive the new u[0] from ``recv_from`` as one group
        (controller_MPI.py:218-305 send_full + recv_full); None skips a direction"""
        self._e(self.engine.lib.sdc_comm_exchange(self.engine.ctx, -1 if send_to is None else int(send_to),
                                                  -1 if recv_from is None else int(recv_from)))

    def send_uend(self, peer):
        self._e(self.engine.lib.sdc_send_uend(self.engine.ctx, int(peer)))

    def recv_u0(self, peer):
        self._e(self.engine.lib.sdc_recv_u0(self.engine.ctx, int(peer)))

    def bcast(self, slot, m=0, root=0):
        """one slab field of rank ``root`` to all ranks, in place (controller_MPI.py:125-130)"""
        self._e(self.engine.lib.sdc_bcast(self.engine.ctx, int(slot), int(m), int(root)))

    def set_chunk(self, doubles_per_piece):
        self._e(self.engine.lib.sdc_comm_set_chunk(self.engine.ctx, int(doubles_per_piece)))

    def sync(self):
        self._e(self.engine.lib.sdc_comm_sync(self.engine.ctx))

    def close(self):
        if self.engine is not None and self.engine.ctx:
            self.engine.lib.sdc_comm_destroy(self.engine.ctx)
        self.engine = None
