"""Device datatype: the behaviour of pySDC's ``mesh`` / ``imex_mesh``
(/root/reference/pySDC/implementations/datatype_classes/mesh.py:12-190) on an MI355X buffer.

A ``hip_mesh`` either owns its storage (a torch float64 tensor used purely as a device allocation) or is a
non-owning *view* of one field of a SweepEngine slab (``L.u[m]`` / ``L.f[m]``, SURVEY.md 8b "Ownership").
Arithmetic runs in HIP kernels through the C-ABI (``sdc_vec_*``); ``abs()`` is the global max norm returned
as a Python float exactly like ``mesh.__abs__`` (mesh.py:65-83), including NaN propagation."""
import ctypes as C

import numpy as np

from pysdc_amd import lib as L
from pysdc_amd.errors import DataError

_F64 = np.dtype('float64')


def _torch():
    import torch

    return torch


class _CAI:
    """holder exposing __cuda_array_interface__ so torch can wrap library-owned device memory (RCCL P2P)."""

    def __init__(self, ptr, n, keep):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}
        self._keep = keep


class device_buffer:
    """raw device allocation holding the bytes of a host array (index / weight tables of the transfer kernels)"""

    def __init__(self, host):
        h = np.ascontiguousarray(host)
        self.nbytes = int(h.nbytes)
        self._alloc()
        self._upload(h)

    def _alloc(self):
        torch = _torch()
        self._buf = torch.empty(max(self.nbytes, 1), dtype=torch.uint8, device='cuda')
        self.ptr = self._buf.data_ptr()

    def _upload(self, h):
        _check_hip(_hip().hipMemcpy(C.c_void_p(self.ptr), h.ctypes.data_as(C.c_void_p), self.nbytes, 1))


def _torch_key(key):
    """an advanced-indexing key with its arrays on the device"""
    torch = _torch()

    def conv(k):
        if isinstance(k, hip_mesh):
            return k.as_torch().reshape(k.shape)
        if isinstance(k, (np.ndarray, list)):
            a = np.asarray(k)
            return torch.as_tensor(a, device='cuda')
        return k

    return tuple(conv(k) for k in key) if isinstance(key, tuple) else conv(key)


class hip_mesh:
    comm = None
    xp = None
    __array_priority__ = 1000  # ndarray * hip_mesh -> hip_mesh.__rmul__

    _lineage = None  # (engine, generation) of the end value this object is an unmodified copy of, or None

    def __init__(self, init=None, val=0.0, *, _ptr=None, _shape=None, _keep=None, _on_write=None, _on_access=None):
        self._buf = None
        self._keep = _keep
        self._on_write = _on_write  # slab views tell their level that device state changed
        self._on_access = _on_access  # ... and ask it to store deferred node fields before the memory is used
        if _ptr is not None:  # non-owning view
            self.shape = tuple(_shape)
            self.size = int(np.prod(self.shape))
            self._p = _ptr if callable(_ptr) else int(_ptr)   # (callable: asked for when the memory is first used - a slab
            return                                             #  block that nobody touches is never allocated)
        if isinstance(init, hip_mesh):
            self.shape = init.shape
            self.size = init.size
            self._alloc()
            _chk(L.load().sdc_vec_copy(None, self.size, init.ptr, self.ptr))
        elif isinstance(init, tuple) and len(init) == 3 and isinstance(init[2], np.dtype):
            if init[2] != _F64:
                raise DataError(f'hip_mesh holds float64 data, got {init[2]}')
            shape = init[0]
            self.shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
            self.size = int(np.prod(self.shape))
            self._alloc()
            if val is not None:  # val=None: contents undefined (the caller overwrites every element)
                _chk(L.load().sdc_vec_fill(None, self.size, float(val), self.ptr))
            type(self).comm = init[1]
        else:
            raise NotImplementedError(type(init))

    def _alloc(self):
        torch = _torch()
        self._buf = torch.empty(self.size, dtype=torch.float64, device='cuda')
        self.ptr = self._buf.data_ptr()

    # ---- views / host access -------------------------------------------------------------------------
    @property
    def ptr(self):
        if self._on_access is not None:
            self._on_access()
        self._lineage = None  # the raw address leaves this object: it may be written behind our back
        if callable(self._p):
            self._p = int(self._p())
        return self._p

    @ptr.setter
    def ptr(self, value):
        self._p = int(value)

    @classmethod
    def view(cls, ptr, shape, keep=None, on_write=None, on_access=None):
        return cls(_ptr=ptr, _shape=shape, _keep=keep, _on_write=on_write, _on_access=on_access)

    def _wrote(self):
        self._lineage = None  # (no longer the value an engine handed out: see controller_nonMPI.run)
        if self._on_write is not None:
            self._on_write()

    @property
    def dtype(self):
        return _F64

    @property
    def ndim(self):
        return len(self.shape)

    def get(self):
        """host copy (cupy_mesh idiom: ``uend.get()``, projects/GPU/heat.py:61-94)."""
        torch = _torch()
        torch.cuda.synchronize()
        out = np.empty(self.size, dtype=np.float64)
        _check_hip(_hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), self.size * 8, 2))
        return out.reshape(self.shape)

    def set(self, host):
        h = np.ascontiguousarray(host, dtype=np.float64).reshape(-1)
        if h.size != self.size:
            raise DataError(f'size mismatch: {h.size} vs {self.size}')
        _check_hip(_hip().hipMemcpy(C.c_void_p(self.ptr), h.ctypes.data_as(C.c_void_p), self.size * 8, 1))
        self._wrote()

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)

    def flatten(self):
        return hip_mesh.view(self.ptr, (self.size,), keep=self, on_write=self._on_write, on_access=self._on_access)

    def as_torch(self):
        """torch tensor aliasing this buffer (for torch.distributed send/recv over RCCL)."""
        self._lineage = None  # whoever holds the tensor may write through it
        if self._buf is not None:
            return self._buf
        torch = _torch()
        return torch.as_tensor(_CAI(self.ptr, self.size, self), device='cuda')

    # ---- indexing (mesh IS an ndarray, datatype_classes/mesh.py:12-60: u[i], u[..., j], u[1:-1], u[mask]) ------------------
    @staticmethod
    def _whole(key):
        return key is Ellipsis or (isinstance(key, slice) and key == slice(None))

    def _box(self, key):
        """basic indexing resolved against self.shape: (start, step, count) per axis and the shape of the result (integer
        axes dropped), or None when the key needs advanced indexing (index arrays, masks, np.newaxis)"""
        key = key if isinstance(key, tuple) else (key,)
        if any(not (isinstance(k, (int, np.integer, slice)) or k is Ellipsis) or isinstance(k, (bool, np.bool_)) for k in key):
            return None
        if sum(k is Ellipsis for k in key) > 1:
            raise IndexError("an index can only have a single ellipsis ('...')")
        nd = len(self.shape)
        given = sum(k is not Ellipsis for k in key)
        if given > nd:
            raise IndexError(f'too many indices for array: array is {nd}-dimensional, but {given} were indexed')
        full = []
        for k in key:
            full += [slice(None)] * (nd - given) if k is Ellipsis else [k]
        full += [slice(None)] * (nd - len(full))
        start, step, count, out_shape = [], [], [], []
        for ax, (k, n) in enumerate(zip(full, self.shape)):
            if isinstance(k, slice):
                a, b, c = k.indices(n)
                cnt = len(range(a, b, c))
                start.append(a if cnt else 0)
                step.append(c)
                count.append(cnt)
                out_shape.append(cnt)
            else:
                i = int(k)
                if not -n <= i < n:
                    raise IndexError(f'index {i} is out of bounds for axis {ax} with size {n}')
                start.append(i % n)
                step.append(1)
                count.append(1)
        return start, step, count, tuple(out_shape)

    def _box_call(self, box, compact_ptr, direction, value=0.0):
        start, step, count, _ = box
        nd = len(self.shape)
        arr = lambda v: (C.c_longlong * nd)(*[int(x) for x in v])   # noqa: E731
        _chk(L.load().sdc_vec_box(None, nd, arr(self.shape), arr(start), arr(step), arr(count), self.ptr, compact_ptr, direction,
                                  float(value)))

    def __setitem__(self, key, value):
        if getattr(self, '_strided_copy', False):
            raise DataError('this field is the COPY a strided index of a device field returned (ndarray would have handed out '
                            'a view): write through the field itself, x[key] = value')
        if not self._whole(key):
            box = self._box(key)
            if box is None:   # index arrays / masks: through a torch view of the same memory
                t = self.as_torch().reshape(self.shape)
                t[_torch_key(key)] = value.as_torch().reshape(value.shape) if isinstance(value, hip_mesh) else _torch().as_tensor(
                    np.asarray(value, dtype=np.float64), device='cuda')
            elif np.isscalar(value):
                self._box_call(box, None, 2, value)
            else:
                shape = box[3]
                if isinstance(value, hip_mesh):
                    if value.shape != shape and value.size != int(np.prod(shape)):
                        raise DataError(f'could not broadcast input of shape {value.shape} into shape {shape}')
                    src = value
                else:
                    src = hip_mesh((shape if shape else (1,), None, _F64), val=None)
                    src.set(np.broadcast_to(np.asarray(value, dtype=np.float64), shape if shape else (1,)))
                self._box_call(box, src.ptr, 1)
            self._wrote()
            return
        if isinstance(value, hip_mesh):
            if value.size != self.size:
                raise DataError(f'size mismatch: {value.size} vs {self.size}')
            if value.ptr != self.ptr:
                _chk(L.load().sdc_vec_copy(None, self.size, value.ptr, self.ptr))
        elif np.isscalar(value):
            _chk(L.load().sdc_vec_fill(None, self.size, float(value), self.ptr))
        else:
            self.set(np.broadcast_to(np.asarray(value, dtype=np.float64), self.shape))
        self._wrote()

    def __getitem__(self, key):
        """whole field: the object itself.  Integers / slices: the selected box gathered on the device into a NEW hip_mesh
        (a copy - where ndarray hands out a view; write through `x[key] = ...`), or a Python float when every axis got an
        integer.  Index arrays and masks: the same through a torch view of the memory."""
        if self._whole(key):
            return self
        box = self._box(key)
        if box is None:
            t = self.as_torch().reshape(self.shape)[_torch_key(key)]
            if t.ndim == 0:
                return float(t.item())
            out = hip_mesh((tuple(t.shape), None, _F64), val=None)
            out.as_torch().reshape(t.shape).copy_(t)
            return out
        shape = box[3]
        off = self._contiguous_offset(box)
        if off is not None and shape and int(np.prod(shape)):
            # a box that is ONE contiguous piece of memory (leading-axis integers / unit-step slices, whole trailing axes): a
            # view, like ndarray's - `v = u[1:-1]; v[:] = x` and `u[0][...] = x` write the field itself
            return hip_mesh.view(self.ptr + 8 * off, shape, keep=self, on_write=self._on_write, on_access=self._on_access)
        out = hip_mesh((shape if shape else (1,), None, _F64), val=None)
        if out.size:
            self._box_call(box, out.ptr, 0)
        if shape:
            out._strided_copy = True   # (where ndarray hands out a strided view: writes through it would be lost - they raise)
        return out if shape else float(out.get()[0])

    def _contiguous_offset(self, box):
        """element offset of the box when it is one contiguous piece of this (C-ordered) field, else None"""
        start, step, count, _ = box
        if any(c > 1 and st != 1 for st, c in zip(step, count)):
            return None
        nd = len(self.shape)
        ax = nd - 1
        while ax >= 0 and start[ax] == 0 and count[ax] == self.shape[ax]:   # whole trailing axes
            ax -= 1
        if any(count[a] != 1 for a in range(ax)):   # at most ONE partial axis; everything before it a single index
            return None
        off, stride = 0, 1
        for a in range(nd - 1, -1, -1):
            off += start[a] * stride
            stride *= self.shape[a]
        return off

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return (self[i] for i in range(self.shape[0]))

    def reshape(self, *shape):
        """another shape over the same memory (ndarray.reshape of a contiguous array: a view)"""
        shape = shape[0] if len(shape) == 1 and not np.isscalar(shape[0]) else shape
        shape = tuple(int(v) for v in shape)
        if -1 in shape:
            known = int(np.prod([v for v in shape if v != -1]))
            shape = tuple(self.size // max(known, 1) if v == -1 else v for v in shape)
        if int(np.prod(shape)) != self.size:
            raise ValueError(f'cannot reshape array of size {self.size} into shape {shape}')
        return hip_mesh.view(self.ptr, shape, keep=self, on_write=self._on_write, on_access=self._on_access)

    def ravel(self):
        return self.flatten()

    def max(self):
        return float(self.as_torch().max().item())

    def min(self):
        return float(self.as_torch().min().item())

    def sum(self):
        return float(self.as_torch().sum().item())

    def mean(self):
        return float(self.as_torch().mean().item())

    # ---- arithmetic (mesh.py:49-63: results keep the datatype) ------------------------------------------
    def _new_like(self):
        out = hip_mesh.__new__(type(self))
        out._keep = None
        out._on_write = out._on_access = None
        out.shape, out.size = self.shape, self.size
        hip_mesh._alloc(out)
        return out

    def _axpby(self, a, x, b, y, out):
        _chk(L.load().sdc_vec_axpby(None, self.size, float(a), None if x is None else x.ptr, float(b),
                                    None if y is None else y.ptr, out.ptr))
        return out

    def _coerce(self, other):
        if isinstance(other, hip_mesh):
            if other.size != self.size:
                raise DataError(f'size mismatch: {other.size} vs {self.size}')
            return other
        tmp = self._new_like()
        tmp[:] = other
        return tmp

    def _constant(self, value):
        """a field filled with one value, made on the device (no host array, no PCIe)"""
        tmp = self._new_like()
        _chk(L.load().sdc_vec_fill(None, self.size, float(value), tmp.ptr))
        return tmp

    def __add__(self, o):
        if np.isscalar(o):
            o = self._constant(o)
        return self._axpby(1.0, self, 1.0, self._coerce(o), self._new_like())

    __radd__ = __add__

    def __sub__(self, o):
        if np.isscalar(o):
            o = self._constant(o)
        return self._axpby(1.0, self, -1.0, self._coerce(o), self._new_like())

    def __rsub__(self, o):
        if np.isscalar(o):
            o = self._constant(o)
        return self._axpby(-1.0, self, 1.0, self._coerce(o), self._new_like())

    def __mul__(self, a):
        if not np.isscalar(a):
            raise NotImplementedError('hip_mesh * non-scalar')
        return self._axpby(float(a), self, 0.0, None, self._new_like())

    __rmul__ = __mul__

    def __truediv__(self, a):
        if not np.isscalar(a):
            raise NotImplementedError('hip_mesh / non-scalar')
        return self._axpby(1.0 / float(a), self, 0.0, None, self._new_like())

    def __neg__(self):
        return self._axpby(-1.0, self, 0.0, None, self._new_like())

    def __iadd__(self, o):
        self._axpby(1.0, self, 1.0, self._coerce(o), self)
        self._wrote()
        return self

    def __isub__(self, o):
        self._axpby(1.0, self, -1.0, self._coerce(o), self)
        self._wrote()
        return self

    def __imul__(self, a):
        self._axpby(float(a), self, 0.0, None, self)
        self._wrote()
        return self

    def iaxpy(self, a, x):
        """self += a * x in one pass (no temporary for a * x)"""
        self._axpby(1.0, self, float(a), self._coerce(x), self)
        self._wrote()
        return self

    def __abs__(self):
        out = C.c_double()
        _chk(L.load().sdc_vec_amax(None, self.size, self.ptr, C.byref(out)))
        return float(out.value)

    def copy(self):
        return type(self)(self) if type(self) is hip_mesh else hip_mesh(self)

    # ---- communication (mesh.py:85-125) through torch.distributed (RCCL on GPUs) -----------------------
    def isend(self, dest=None, tag=None, comm=None):
        import torch.distributed as dist

        return dist.isend(self.as_torch(), dst=dest, group=comm, tag=tag or 0)

    def irecv(self, source=None, tag=None, comm=None):
        import torch.distributed as dist

        return dist.irecv(self.as_torch(), src=source, group=comm, tag=tag or 0)

    def bcast(self, root=None, comm=None):
        import torch.distributed as dist

        dist.broadcast(self.as_torch(), src=root, group=comm)
        return self

    def __reduce__(self):
        # controller_nonMPI clones steps with dill.copy and only falls back to re-instantiation on
        # PicklingError / TypeError / ValueError (controller_nonMPI.py:37-45): device buffers do not pickle
        raise TypeError('hip_mesh holds device memory and cannot be pickled')


class hip_imex_mesh:
    """two components ``impl`` / ``expl`` in one buffer (datatype_classes/mesh.py:166-173)."""

    components = ['impl', 'expl']

    def __init__(self, init=None, val=0.0, *, _parts=None):
        if _parts is not None:
            self.impl, self.expl = _parts
        elif isinstance(init, hip_imex_mesh):
            self.impl, self.expl = hip_mesh(init.impl), hip_mesh(init.expl)
        elif isinstance(init, tuple):
            self.impl, self.expl = hip_mesh(init, val), hip_mesh(init, val)
        else:
            raise NotImplementedError(type(init))
        self.shape = (2,) + self.impl.shape

    @classmethod
    def view(cls, ptr_impl, ptr_expl, shape, keep=None, on_write=None, on_access=None):
        return cls(_parts=(hip_mesh.view(ptr_impl, shape, keep, on_write, on_access),
                           hip_mesh.view(ptr_expl, shape, keep, on_write, on_access)))

    def get(self):
        return np.stack([self.impl.get(), self.expl.get()])

    def __array__(self, dtype=None, copy=None):
        return self.get()

    def __setitem__(self, key, value):
        if isinstance(value, hip_imex_mesh):
            self.impl[:] = value.impl
            self.expl[:] = value.expl
        elif np.isscalar(value):
            self.impl[:] = value
            self.expl[:] = value
        else:
            v = np.asarray(value)
            self.impl[:] = v[0]
            self.expl[:] = v[1]

    def __getitem__(self, key):
        if key == 0:
            return self.impl
        if key == 1:
            return self.expl
        return self

    # component-wise arithmetic (MultiComponentMesh keeps ndarray arithmetic, datatype_classes/mesh.py:128-173)
    def _zip(self, other, op):
        if isinstance(other, hip_imex_mesh):
            return hip_imex_mesh(_parts=(op(self.impl, other.impl), op(self.expl, other.expl)))
        return hip_imex_mesh(_parts=(op(self.impl, other), op(self.expl, other)))

    def __add__(self, o):
        return self._zip(o, lambda a, b: a + b)

    __radd__ = __add__

    def __sub__(self, o):
        return self._zip(o, lambda a, b: a - b)

    def __mul__(self, a):
        return self._zip(a, lambda x, y: x * y)

    __rmul__ = __mul__

    def __iadd__(self, o):
        if isinstance(o, hip_imex_mesh):
            self.impl += o.impl
            self.expl += o.expl
        else:
            self.impl += o
            self.expl += o
        return self

    def iaxpy(self, a, x):
        self.impl.iaxpy(a, x.impl if isinstance(x, hip_imex_mesh) else x)
        self.expl.iaxpy(a, x.expl if isinstance(x, hip_imex_mesh) else x)
        return self

    def __isub__(self, o):
        if isinstance(o, hip_imex_mesh):
            self.impl -= o.impl
            self.expl -= o.expl
        else:
            self.impl -= o
            self.expl -= o
        return self


# ---- raw HIP runtime access for host copies (same runtime instance the engine uses) ---------------------
_hiprt = None


def _hip():
    global _hiprt
    if _hiprt is None:
        L.load()
        for name in ('libamdhip64.so.7', 'libamdhip64.so'):
            try:
                _hiprt = C.CDLL(name, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
        if _hiprt is None:
            from pysdc_amd.errors import EngineError

            raise EngineError('cannot find the HIP runtime (libamdhip64)')
        _hiprt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hiprt.hipMemcpy.restype = C.c_int
    return _hiprt


def _check_hip(rc):
    if rc != 0:
        from pysdc_amd.errors import EngineError

        raise EngineError(f'HIP runtime error {rc}')


def _chk(rc):
    L.check(rc, None)
