"""TEST-ONLY stand-in for the HIP library on a host without a GPU.

``host_device()`` patches, for the duration of a test, the three places where pysdc_amd touches the device:
``pysdc_amd.lib.load`` (the C-ABI handle: only the ``sdc_vec_*`` entry points the datatype uses are emulated, on
host memory), ``hip_mesh`` allocation / host copies, and ``SweepEngine`` (replaced by ``HostEngine``: same methods,
slabs in NumPy arrays, numerics delegated to the CPU oracle).  It exists so that the HOST-SIDE plumbing - sweeper /
problem / datatype classes, ``ForeignLevelState``, slab-backed lists - can run inside the REFERENCE's own ``Step`` and
controllers in the build container (tests/test_reference_stack.py).  It is not a CPU fallback of the product: nothing
under pysdc_amd/ imports it, and without these patches the product raises (no GPU, or no libsdcmi.so)."""
import contextlib
import ctypes as C

import numpy as np

from oracle import sdc_oracle as O


def _addr(p):
    if p is None:
        return None
    if isinstance(p, C.c_void_p):
        return p.value
    return int(p)


def _arr(p, n):
    return np.ctypeslib.as_array((C.c_double * int(n)).from_address(_addr(p)))


class HostLib:
    """the subset of include/sdcmi.h the datatype classes call, on host pointers"""

    def sdc_vec_copy(self, ctx, n, x, y):
        _arr(y, n)[:] = _arr(x, n)
        return 0

    def sdc_vec_fill(self, ctx, n, a, y):
        _arr(y, n)[:] = a
        return 0

    def sdc_vec_axpby(self, ctx, n, a, x, b, y, z):
        out = np.zeros(int(n)) if x is None and y is None else None
        if x is not None and y is not None:
            out = a * _arr(x, n) + b * _arr(y, n)
        elif x is not None:
            out = a * _arr(x, n)
        elif y is not None:
            out = b * _arr(y, n)
        _arr(z, n)[:] = out
        return 0

    def sdc_vec_amax(self, ctx, n, x, out):
        v = _arr(x, n)
        out._obj.value = float(np.max(np.abs(v))) if not np.isnan(v).any() else float('nan')
        return 0

    def sdc_vec_box(self, ctx, ndim, shape, start, step, count, field, compact, direction, value):
        shp = [int(shape[d]) for d in range(ndim)]
        f = _arr(field, int(np.prod(shp))).reshape(shp)
        cnt = [int(count[d]) for d in range(ndim)]
        if int(np.prod(cnt)) == 0:
            return 0
        key = tuple(np.arange(cnt[d]) * int(step[d]) + int(start[d]) for d in range(ndim))
        sel = np.ix_(*key)
        if direction == 0:
            _arr(compact, int(np.prod(cnt)))[:] = f[sel].reshape(-1)
        elif direction == 1:
            f[sel] = _arr(compact, int(np.prod(cnt))).reshape(cnt)
        else:
            f[sel] = value
        return 0

    def sdc_last_error(self, ctx):
        return b'host stand-in'

    engine = None   # the HostEngine this handle belongs to (entry points that take a context)

    def sdc_defer_f0(self, ctx):
        """F[0] = f(U[0]) after U[0] was replaced (include/sdcmi.h: put off on the device, evaluated at once here)"""
        e = self.engine
        g0 = e.gvals[0] if e.gvals is not None else 0.0
        e.eval_f(e.U[0].ctypes.data, g0, e.F[0, 0].ctypes.data, e.F[0, 1].ctypes.data if e.ncomp == 2 else None)
        return 0

    # space transfer (include/sdcmi.h: sdc_transfer_apply[_batch]): the 1-D operator given as fixed-width rows, stored
    # entry-major ([width][n_out]), applied along every axis
    def sdc_transfer_apply_batch(self, stream, nfields, ndim, n_out, n_in, width, idx, w, src, dst):
        ix = np.ctypeslib.as_array((C.c_int32 * (width * n_out)).from_address(_addr(idx))).reshape(width, n_out)
        wt = _arr(w, width * n_out).reshape(width, n_out)
        M = np.zeros((n_out, n_in))
        for k in range(width):
            for i in range(n_out):
                M[i, ix[k, i]] += wt[k, i]
        fin = _arr(src, nfields * n_in**ndim).reshape((nfields,) + (n_in,) * ndim)
        out = fin
        for ax in range(1, ndim + 1):
            out = np.moveaxis(np.tensordot(M, out, axes=([1], [ax])), 0, ax)
        _arr(dst, nfields * n_out**ndim)[:] = out.reshape(-1)
        return 0

    def sdc_transfer_apply(self, stream, ndim, n_out, n_in, width, idx, w, src, dst):
        return self.sdc_transfer_apply_batch(stream, 1, ndim, n_out, n_in, width, idx, w, src, dst)


class _Prob:
    """oracle-style problem (eval_f / solve_system on ndarrays) built from what the engine was configured with"""

    def __init__(self, eng):
        self.e = eng
        self.imex = eng.ncomp == 2
        self.nvars = eng.nvars

    def u_init(self):
        return np.zeros(self.nvars)

    def f_init(self):
        return np.zeros((2,) + self.nvars) if self.imex else np.zeros(self.nvars)

    def _apply(self, which, u):
        off, w = self.e.stencil[which]
        out = np.zeros_like(u)
        for ax in range(u.ndim):
            for s, ws in zip(off, w):
                out += ws * np.roll(u, -s, axis=ax)
        return out

    def eval_f(self, u, t):
        if not self.imex:
            return self._apply(0, u)
        f = np.zeros((2,) + self.nvars)
        f[0] = self._apply(0, u)
        if 1 in self.e.stencil:
            f[1] = self._apply(1, u)
        elif self.e.profile is not None:
            f[1] = self.e.profile * self.e.g_of(t)
        return f

    def solve_system(self, rhs, factor, u0, t):
        n = self.nvars[0]
        off, w = self.e.stencil[0]
        k = np.arange(n)
        lam1 = sum(ws * np.exp(2j * np.pi * k * s / n) for s, ws in zip(off, w))
        lam = 0
        for ax in range(len(self.nvars)):
            shape = [1] * len(self.nvars)
            shape[ax] = n
            lam = lam + lam1.reshape(shape)
        return np.real(np.fft.ifftn(np.fft.fftn(rhs) / (1.0 - factor * lam)))


class HostEngine:
    """pysdc_amd.engine.SweepEngine's method surface over NumPy slabs (periodic finite-difference problems)"""

    SLOT_U, SLOT_F, SLOT_TAU, SLOT_UEND = 0, 1, 2, 3

    def __init__(self, nvars, num_nodes, ncomp=1, device=None, stream=0):
        self.nvars = (nvars,) if isinstance(nvars, int) else tuple(int(v) for v in nvars)
        self.ndim, self.n = len(self.nvars), self.nvars[0]
        self.N = int(np.prod(self.nvars))
        self.M, self.ncomp = int(num_nodes), int(ncomp)
        self.U = np.zeros((self.M + 1, self.N))
        self.F = np.zeros((self.M + 1, self.ncomp, self.N))
        self.TAU = np.zeros((self.M, self.N))
        self.UEND = np.zeros(self.N)
        self.stencil, self.profile, self.gvals, self.times = {}, None, None, None
        self.tau_active = False
        self.lib = HostLib()
        self.lib.engine = self
        self.ctx = None
        self.calls = []

    def close(self):
        pass

    # ---- setup -------------------------------------------------------------------------------------------
    def set_coeffs(self, Qmat, QI, QE, nodes, weights):
        self.coll = O.Coll(np.array(nodes), np.array(weights), np.array(Qmat), np.array(QI),
                           None if QE is None else np.array(QE))

    def set_stencil(self, which, offsets, weights):
        self.stencil[which] = ([int(o) for o in offsets], [float(x) for x in weights])

    def set_forcing_profile(self, profile):
        self.profile = np.array(profile, dtype=float).reshape(self.nvars)

    def set_forcing_values(self, g):
        self.gvals = [float(x) for x in g]

    def g_of(self, t):
        return self.gvals[int(np.argmin([abs(t - s) for s in self.times]))]

    def set_solver(self, kind, rtol=1e-12, maxiter=10000):
        assert kind == 'direct'

    def set_tau_active(self, active):
        self.tau_active = bool(active)

    def _noop(self, *a, **k):
        pass

    invalidate_spectra = set_fused_residual = set_skip_residual = set_deferred = set_unlocked = _noop
    set_spectral_reuse = materialize = sync = profile_enable = set_keep_residual_fields = set_early_end_point = _noop

    # ---- data --------------------------------------------------------------------------------------------
    def _slab(self, slot, m=0, comp=0):
        if slot == self.SLOT_U:
            return self.U[m]
        if slot == self.SLOT_F:
            return self.F[m, comp]
        if slot == self.SLOT_TAU:
            return self.TAU[m]
        return self.UEND

    def ptr(self, slot, m=0, comp=0):
        return self._slab(slot, m, comp).ctypes.data

    def uend_address(self):
        return self.UEND.ctypes.data

    def upload(self, slot, m, host, comp=0):
        self._slab(slot, m, comp)[:] = np.asarray(host, dtype=float).reshape(-1)

    def download(self, slot, m=0, comp=0):
        return self._slab(slot, m, comp).reshape(self.nvars).copy()

    # ---- the sweep path, through the oracle's restatement of the reference ---------------------------------------
    def _level(self, t, dt):
        L = O.Level(_Prob(self), self.coll, dt)
        L.time = t
        L.u = [self.U[m].reshape(self.nvars) for m in range(self.M + 1)]
        shape = ((2,) + self.nvars) if self.ncomp == 2 else self.nvars
        L.f = [self.F[m].reshape(shape) for m in range(self.M + 1)]
        L.tau = [self.TAU[m].reshape(self.nvars) if self.tau_active else None for m in range(self.M)]
        L.unlocked = True
        self.times = [t] + [t + dt * tau for tau in self.coll.nodes]
        return L

    def _store(self, L):
        for m in range(self.M + 1):
            self.U[m] = np.asarray(L.u[m]).reshape(-1)
            self.F[m] = np.asarray(L.f[m]).reshape(self.ncomp, -1)

    def predict(self, t, dt, guess='spread', fill_u=0.0, fill_f=0.0):
        self.calls.append('predict')
        L = self._level(t, dt)
        O.predict(L, guess)
        self._store(L)
        self._t, self._dt = t, dt

    def sweep(self, t, dt):
        self.calls.append('sweep')
        L = self._level(t, dt)
        O.sweep(L)
        self._store(L)
        self._t, self._dt = t, dt

    def set_lazy_predictor_residual(self, on):
        pass      # (nothing is put off on the host)

    # what the product's own controllers use on their Level objects (Level.advance, run-to-run tokens)
    def advance(self):
        self.U[0][:] = self.UEND

    def end_value_generation(self):
        return 0      # (no token: every run loads its start value)

    def residual_deferred(self):
        return False

    def residual(self, dt, residual_type='full_abs'):
        self.calls.append('residual')
        L = self._level(getattr(self, '_t', 0.0), dt)
        L.residual_type = residual_type
        norms = O.compute_residual(L)
        return L.status_residual, np.array(norms)

    integrals_written = False

    def residual_post(self, dt, residual_type='full_abs', restol=-1.0, integrals=None):
        """the queued form of the device engine (SweepEngine.residual_post): on the host the number is there at once; the
        quadrature sums are written too when asked for"""
        from pysdc_amd.engine import ResidualFuture

        res, norms = self.residual(dt, residual_type)
        self.integrals_written = integrals is not None
        if integrals is not None:
            self.integrate(dt, integrals)
        return ResidualFuture.ready(res, norms, restol)

    def end_point(self, dt, do_coll_update):
        self.calls.append('end_point')
        L = self._level(getattr(self, '_t', 0.0), dt)
        O.compute_end_point(L, do_coll_update)
        self.UEND[:] = L.uend.reshape(-1)

    def integrate(self, dt, dst_ptrs):
        L = self._level(getattr(self, '_t', 0.0), dt)
        for p, v in zip(dst_ptrs, O.integrate(L)):
            _arr(p, self.N)[:] = v.reshape(-1)

    def eval_f(self, u_ptr, g_t, fi_ptr, fe_ptr=None):
        u = _arr(u_ptr, self.N).reshape(self.nvars)
        P = _Prob(self)
        _arr(fi_ptr, self.N)[:] = P._apply(0, u).reshape(-1)
        if fe_ptr is not None:
            if 1 in self.stencil:
                _arr(fe_ptr, self.N)[:] = P._apply(1, u).reshape(-1)
            else:
                _arr(fe_ptr, self.N)[:] = (self.profile * g_t).reshape(-1)

    def solve(self, rhs_ptr, factor, out_ptr, guess_ptr=None):
        rhs = _arr(rhs_ptr, self.N).reshape(self.nvars)
        _arr(out_ptr, self.N)[:] = _Prob(self).solve_system(rhs, factor, None, 0.0).reshape(-1)

    def vec_copy(self, n, x, y):
        self.lib.sdc_vec_copy(None, n, x, y)

    def vec_fill(self, n, a, y):
        self.lib.sdc_vec_fill(None, n, a, y)

    def work_counters(self):
        return dict(newton=0, rhs=0, failed=0, CG=0)

    @property
    def device_bytes(self):
        return 0


@contextlib.contextmanager
def host_device():
    from pysdc_amd import engine as E, hip_mesh as HM, level as LV, lib as LB

    saved = (LB.load, HM.hip_mesh._alloc, HM.hip_mesh.get, HM.hip_mesh.set, E.SweepEngine, LV.SweepEngine)
    saved_buf = (HM.device_buffer._alloc, HM.device_buffer._upload)

    def _balloc(self):
        self._buf = np.zeros(max(self.nbytes, 1), dtype=np.uint8)
        self.ptr = self._buf.ctypes.data

    def _bupload(self, h):
        self._buf[:self.nbytes] = np.frombuffer(h.tobytes(), dtype=np.uint8)

    HM.device_buffer._alloc, HM.device_buffer._upload = _balloc, _bupload
    hostlib = HostLib()

    def _alloc(self):
        self._buf = np.empty(self.size, dtype=np.float64)
        self.ptr = self._buf.ctypes.data

    def _get(self):
        return _arr(self.ptr, self.size).reshape(self.shape).copy()

    def _set(self, host):
        _arr(self.ptr, self.size)[:] = np.asarray(host, dtype=float).reshape(-1)
        self._wrote()

    LB.load = lambda: hostlib
    HM.hip_mesh._alloc, HM.hip_mesh.get, HM.hip_mesh.set = _alloc, _get, _set
    E.SweepEngine = LV.SweepEngine = HostEngine
    try:
        yield
    finally:
        LB.load, HM.hip_mesh._alloc, HM.hip_mesh.get, HM.hip_mesh.set, E.SweepEngine, LV.SweepEngine = saved
        HM.device_buffer._alloc, HM.device_buffer._upload = saved_buf
