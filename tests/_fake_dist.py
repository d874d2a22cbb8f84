"""In-process stand-in for the part of torch.distributed that pysdc_amd.controller.controller_dist uses: every rank is
a THREAD of one process, all ranks share one GPU.  Device levels hand their state over through the C-ABI communicator
itself (controller parameter comm_wire='shm': the shared-memory mailbox wire of sdc_comm_*, which works between threads
like between processes), so this stand-in only carries what torch.distributed carries in a real run: the unique id,
convergence flags, step counts, barriers.  (Its point-to-point operations - snapshots with stream-ordered events - still
serve controllers whose levels are not device levels.)"""
import queue
import threading

import torch

_local = threading.local()


class _Work:
    def __init__(self, fn=None):
        self._fn = fn

    def wait(self):
        if self._fn is not None:
            fn, self._fn = self._fn, None
            fn()
        return True


class P2POp:
    def __init__(self, op, tensor, peer, group=None, tag=0):
        self.op, self.tensor, self.peer, self.group, self.tag = op, tensor, peer, group, tag


def isend(*a, **k):  # only used as a marker inside P2POp
    raise NotImplementedError


def irecv(*a, **k):
    raise NotImplementedError


class World:
    def __init__(self, size):
        self.size = size
        self.q = {}
        self.lock = threading.Lock()
        self.barrier = threading.Barrier(size, timeout=120)
        self.red = [None] * size
        self.bc = {}

    def chan(self, kind, src, dst):
        with self.lock:
            return self.q.setdefault((kind, src, dst), queue.Queue())


def bind(world, rank):
    _local.world, _local.rank = world, rank


def get_rank(group=None):
    return _local.rank


def get_world_size(group=None):
    return _local.world.size


def get_backend(group=None):
    return 'in-process'


def new_group(backend=None, **kwargs):
    return ('group', backend)


def _put(kind, tensor, dst):
    snap = tensor.detach().clone()              # taken on the sender's current stream
    ev = None
    if snap.is_cuda:
        ev = torch.cuda.Event()
        ev.record()
    _local.world.chan(kind, _local.rank, dst).put((snap, ev))
    return ev


def _get(kind, tensor, src):
    snap, ev = _local.world.chan(kind, src, _local.rank).get(timeout=120)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
    tensor.copy_(snap)
    if snap.is_cuda:
        # the snapshot was allocated on the SENDER's stream: its block must not be handed out again before the copy
        # above, which runs on this stream, is done
        snap.record_stream(torch.cuda.current_stream())


def send(tensor, dst, group=None, tag=0):
    _put('flag', tensor, dst)


def recv(tensor, src, group=None, tag=0):
    _get('flag', tensor, src)


def all_reduce(tensor, group=None, op=None):
    w = _local.world
    w.red[_local.rank] = tensor.detach().cpu().clone()
    w.barrier.wait()
    total = sum(w.red[1:], w.red[0].clone())
    w.barrier.wait()
    tensor.copy_(total.to(tensor.device))


def broadcast(tensor, src, group=None):
    w = _local.world
    if _local.rank == src:
        snap = tensor.detach().clone()
        if snap.is_cuda:
            torch.cuda.current_stream().synchronize()
        w.bc['v'] = snap
    w.barrier.wait()
    if _local.rank != src:
        tensor.copy_(w.bc['v'])
        if tensor.is_cuda:
            torch.cuda.current_stream().synchronize()
    w.barrier.wait()


def broadcast_object_list(box, src=0, group=None):
    """host objects (the unique id of the C-ABI communicator) from rank `src` to every rank"""
    w = _local.world
    if _local.rank == src:
        w.bc['obj'] = list(box)
    w.barrier.wait()
    if _local.rank != src:
        box[:] = w.bc['obj']
    w.barrier.wait()


def barrier(group=None):
    _local.world.barrier.wait()


def batch_isend_irecv(ops):
    """sends are posted at once (snapshot on the caller's current stream); receives complete in wait(), on the
    stream that is current THEN - like a ProcessGroupNCCL work object"""
    works = []
    for op in ops:
        if op.op is isend:
            ev = _put('data', op.tensor, op.peer)
            # waiting for a send = the waiting stream may not touch the buffer before the snapshot was taken
            works.append(_Work((lambda e=ev: torch.cuda.current_stream().wait_event(e)) if ev is not None else None))
    for op in ops:
        if op.op is irecv:
            works.append(_Work(lambda t=op.tensor, s=op.peer: _get('data', t, s)))
    return works
