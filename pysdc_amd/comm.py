"""Transport of the time-parallel hand-over through the C-ABI (include/sdcmi.h: sdc_comm_*).

The role of the reference's ``NCCLComm`` (pySDC/helpers/NCCL_communicator.py:7-20: an MPI communicator whose data
calls go to NCCL) for the exchanges this path has: rank 0 makes the unique id, a HOST-side communicator ships its
128 bytes to the other ranks (``host_bcast``; mpi4py ``comm.bcast`` in the reference, ``torch.distributed`` over gloo
here), every rank then joins on its own GPU.  Two wires behind the same calls: RCCL over xGMI (one process per GPU), and
host mailboxes in shared memory for ranks RCCL cannot connect (several ranks on one GPU, ranks that are threads of one
process) - chosen by the unique id (``rccl_unique_id`` / ``shm_unique_id``).  Messages are posted on a stream of the
engine's own and ordered against its kernels by events."""
import ctypes as C
import os
import secrets

from pysdc_amd import lib as L
from pysdc_amd.errors import ParameterError


def rccl_unique_id():
    """128 bytes identifying a new RCCL communicator; call on ONE rank and ship the result to the others"""
    buf = C.create_string_buffer(128)
    L.check(L.load().sdc_comm_unique_id(buf), None)
    return buf.raw


unique_id = rccl_unique_id


def shm_unique_id():
    """128 bytes naming a fresh set of shared-memory mailboxes (the second wire); call on ONE rank"""
    name = f'shm:{os.getpid()}-{secrets.token_hex(8)}'.encode()
    return name + b'\0' * (128 - len(name))


def make_unique_id(wire):
    if wire == 'rccl':
        return rccl_unique_id()
    if wire == 'shm':
        return shm_unique_id()
    raise ParameterError(f"wire must be 'rccl' or 'shm', got {wire!r}")


def torch_host_bcast(uid, root=0, group=None, dist=None):
    """ship the id with torch.distributed (any backend that moves host objects, e.g. gloo)"""
    if dist is None:
        import torch.distributed as dist

    box = [uid]
    dist.broadcast_object_list(box, src=root, group=group)
    return box[0]


def ranks_share_host_memory(rank, host_bcast, host_all_ok):
    """do all ranks see the same /dev/shm?  The pinned-host second path of two-rank hand-overs (sdc_comm_set_host_share) is a
    ring in POSIX shared memory: it only exists between ranks of ONE host.  Rank 0 creates a small file holding a nonce and
    ships name and nonce with `host_bcast(obj_or_None) -> obj`; every rank looks for it; `host_all_ok(flag) -> bool` tells
    whether all of them found it.  Both calls are collective and are made by every rank whatever happens locally (a rank
    that cannot create or read the file says 'no' in the second one); a failure of the calls themselves means 'no'."""
    import os
    import secrets

    created, probe = None, (None, None)
    if rank == 0:
        try:
            name, nonce = f'/dev/shm/pysdc_amd.probe.{os.getpid()}.{secrets.token_hex(8)}', secrets.token_hex(16)
            with open(name, 'w') as f:
                f.write(nonce)
            created, probe = name, (name, nonce)
        except OSError:
            pass
    try:
        name, nonce = host_bcast(probe if rank == 0 else None)
        mine = False
        if name:
            try:
                with open(name) as f:
                    mine = f.read() == nonce
            except OSError:
                pass
        return bool(host_all_ok(mine))
    except Exception:  # noqa: BLE001  (the host group itself failed: the path stays off)
        return False
    finally:
        if created:
            try:
                os.unlink(created)
            except OSError:
                pass


class DeviceComm:
    """one time rank's end of the communicator, bound to one level's SweepEngine.  The fine level owns the
    communicator (``DeviceComm(engine, P, r, ...)``); the coarser levels of the same rank share it (``attach``)."""

    def __init__(self, engine, nranks, rank, uid=None, host_bcast=None, wire='rccl', _owner=None):
        self.engine = engine
        if _owner is not None:
            self.size, self.rank = _owner.size, _owner.rank
            L.check(engine.lib.sdc_comm_attach(engine.ctx, _owner.engine.ctx), engine.ctx)
            return
        if uid is None:
            if nranks > 1 and host_bcast is None:
                raise ParameterError('more than one rank: pass the unique id or a host_bcast(uid_or_None) -> uid callable')
            uid = make_unique_id(wire) if rank == 0 else None
            if host_bcast is not None:
                uid = host_bcast(uid)
        if len(uid) != 128:
            raise ParameterError('the unique id has 128 bytes')
        self.size, self.rank = int(nranks), int(rank)
        L.check(engine.lib.sdc_comm_init(engine.ctx, uid, self.size, self.rank), engine.ctx)

    @classmethod
    def attach(cls, engine, owner):
        return cls(engine, owner.size, owner.rank, _owner=owner)

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

    def handover_post(self, nactive):
        """lock-step runs: uend(r) -> u[0](r + 1) for all ``nactive`` ranks, posted behind the end value only"""
        self._e(self.engine.lib.sdc_comm_handover_post(self.engine.ctx, int(nactive)))

    def handover_complete(self):
        self._e(self.engine.lib.sdc_comm_handover_complete(self.engine.ctx))

    def bcast(self, slot, m=0, root=0):
        """one slab field of rank ``root`` to all ranks, in place (controller_MPI.py:125-130)"""
        self._e(self.engine.lib.sdc_bcast(self.engine.ctx, int(slot), int(m), int(root)))

    def bcast_buffer(self, ptr, n, root=0):
        """any device buffer of n doubles, in place"""
        self._e(self.engine.lib.sdc_comm_bcast_buffer(self.engine.ctx, ptr, int(n), int(root)))

    def bcast_end_spectrum(self, root):
        """the end value of a block as its half spectrum, into the other ranks' spectrum inboxes"""
        self._e(self.engine.lib.sdc_comm_bcast_end_spectrum(self.engine.ctx, int(root)))

    def set_chunk(self, doubles_per_piece):
        self._e(self.engine.lib.sdc_comm_set_chunk(self.engine.ctx, int(doubles_per_piece)))

    def set_format(self, spectra):
        """lock-step hand-overs carry half spectra instead of fields (levels that sweep in Fourier space)"""
        self._e(self.engine.lib.sdc_comm_set_format(self.engine.ctx, int(bool(spectra))))

    def set_relay(self, on):
        self._e(self.engine.lib.sdc_comm_set_relay(self.engine.ctx, int(bool(on))))

    def set_host_share(self, share):
        """two ranks: this fraction of every lock-step hand-over goes through pinned host memory beside the direct message
        (include/sdcmi.h: sdc_comm_set_host_share)"""
        self._e(self.engine.lib.sdc_comm_set_host_share(self.engine.ctx, float(share)))

    def info(self):
        rank, size = C.c_int(), C.c_int()
        hops, mesh = C.c_ulonglong(), C.c_ulonglong()
        kind = C.create_string_buffer(16)
        self._e(self.engine.lib.sdc_comm_info(self.engine.ctx, C.byref(rank), C.byref(size), C.byref(hops), C.byref(mesh),
                                              kind))
        return dict(rank=rank.value, size=size.value, two_hop_handovers=hops.value, mesh_broadcasts=mesh.value,
                    wire=kind.value.decode())

    def sync(self):
        self._e(self.engine.lib.sdc_comm_sync(self.engine.ctx))

    def close(self):
        if self.engine is not None and self.engine.ctx:
            self.engine.lib.sdc_comm_destroy(self.engine.ctx)
        self.engine = None


RcclComm = DeviceComm  # (the name of the first version: one wire)
