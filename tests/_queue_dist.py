"""A minimal in-memory stand-in for the torch.distributed calls controller_dist's relay code makes (P2POp, isend, irecv,
batch_isend_irecv, broadcast), for CPU tensors and ranks that are threads of one process: lets the index arithmetic
of the two-hop hand-over and of the scatter + all-gather broadcast be checked for 3 .. 8 ranks without a process
group (test infrastructure only)."""
import collections
import threading

_local = threading.local()


class World:
    def __init__(self, size):
        self.size = size
        self.cond = threading.Condition()
        self.box = collections.defaultdict(collections.deque)   # (src, dst) -> messages in posting order
        self.bcast = {}
        self.barrier = threading.Barrier(size)


def bind(world, rank):
    _local.world, _local.rank = world, rank


def get_rank(group=None):
    return _local.rank


def get_world_size(group=None):
    return _local.world.size


def isend(*a, **k):   # markers only: P2POp carries them
    raise NotImplementedError


def irecv(*a, **k):
    raise NotImplementedError


class P2POp:
    def __init__(self, op, tensor, peer, group=None, tag=0):
        self.op, self.tensor, self.peer, self.tag = op, tensor, peer, tag


class _Done:
    def wait(self):
        return True


class _Recv:
    def __init__(self, world, src, dst, tensor, tag):
        self.world, self.key, self.tensor, self.tag = world, (src, dst), tensor, tag

    def wait(self):
        w = self.world
        with w.cond:
            ok = w.cond.wait_for(lambda: len(w.box[self.key]) > 0, timeout=30)
            assert ok, f'no message {self.key}'
            tag, data = w.box[self.key].popleft()
        assert tag == self.tag, (tag, self.tag)
        assert data.numel() == self.tensor.numel(), (self.key, data.numel(), self.tensor.numel())
        self.tensor.copy_(data)
        return True


def batch_isend_irecv(ops):
    w, r = _local.world, _local.rank
    reqs = []
    for op in ops:
        if op.op is isend:
            with w.cond:
                w.box[(r, op.peer)].append((op.tag, op.tensor.clone()))
                w.cond.notify_all()
            reqs.append(_Done())
        else:
            reqs.append(_Recv(w, op.peer, r, op.tensor, op.tag))
    return reqs


def broadcast(tensor, src, group=None):
    w, r = _local.world, _local.rank
    if r == src:
        w.bcast['data'] = tensor.clone()
    w.barrier.wait()
    if r != src:
        tensor.copy_(w.bcast['data'])
    w.barrier.wait()
