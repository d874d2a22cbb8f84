"""Thin object wrapper around one libsdcmi context = the device state of one pySDC Level."""
import collections
import ctypes as C
import weakref

import numpy as np

from pysdc_amd import lib as L
from pysdc_amd.errors import ParameterError


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class ResidualFuture:
    """A residual that is on its way (include/sdcmi.h: sdc_residual_post): the device work is queued, the number - and the
    node norms, and `residual <= restol` as the device found it - are read from pinned host memory when somebody asks.
    Stands for its number wherever one is needed (float(); calling it returns the number, which is how LevelStatus
    evaluates what it holds).  ``fetch(block)`` returns (residual, norms, converged) or None while the record is not there."""

    queued = True    # (the work is on the stream already: nothing about it has to happen before the state changes)
    __slots__ = ('_fetch_fn', '_value', '_norms', '_converged', 'restol', '__weakref__')

    def __init__(self, fetch, restol=-1.0):
        self._fetch_fn, self._value, self._norms, self._converged = fetch, None, None, None
        self.restol = restol    # the tolerance the device's flag is taken against (a reader with another one compares itself)

    @classmethod
    def ready(cls, value, norms, restol=-1.0):
        """a residual that is known already (engines without a queue: tests/_host_engine.py)"""
        f = cls(None, restol)
        f._value, f._norms, f._converged = float(value), np.array(norms, dtype=float), bool(restol >= 0 and value <= restol)
        return f

    def _fetch(self, block):
        got = self._fetch_fn(block)
        if got is None:
            return False
        self._value, self._norms, self._converged = got
        self._fetch_fn = None
        return True

    def done(self):
        """has the record arrived?  Never waits."""
        return self._value is not None or self._fetch(False)

    def result(self):
        if self._value is None:
            self._fetch(True)
        return self._value

    @property
    def norms(self):
        self.result()
        return self._norms

    @property
    def converged(self):
        """residual <= restol, taken on the device against the tolerance of sdc_set_restol"""
        self.result()
        return self._converged

    def __float__(self):
        return float(self.result())

    def __call__(self):
        return self.result()

    def __repr__(self):
        return f'ResidualFuture({self._value if self._value is not None else "pending"})'


class SweepEngine:
    """Device slabs U[(M+1)][N], F[(M+1)][ncomp][N], TAU[M][N], UEND[N] plus the sweep entry points.

    nvars: tuple of equal even ints (ndim <= 3); ncomp: 1 (implicit) or 2 (IMEX)."""

    def __init__(self, nvars, num_nodes, ncomp=1, device=None, stream=0):
        self.lib = L.load()
        if device is None:  # one process per GPU: follow torch's current device (LOCAL_RANK)
            try:
                import torch

                device = torch.cuda.current_device() if torch.cuda.is_available() else 0
            except Exception:  # pragma: no cover
                device = 0
        nvars = (nvars,) if isinstance(nvars, int) else tuple(int(v) for v in nvars)
        if len(set(nvars)) != 1:
            raise ParameterError('need a square domain, got %s' % (nvars,))
        self.nvars, self.ndim, self.n = nvars, len(nvars), nvars[0]
        self.N = int(np.prod(nvars))
        self.M, self.ncomp, self.device = int(num_nodes), int(ncomp), int(device)
        self.ctx = C.c_void_p()
        L.check(self.lib.sdc_ctx_create(C.byref(self.ctx), device, self.ndim, self.n, self.M, self.ncomp,
                                        C.c_void_p(stream)))
        self.tau_active = False
        self._futures = collections.deque()   # (ticket, weak reference) of residuals on their way (residual_post)
        self._issued = 0                      # last ticket of the residual ring handed out (posted or blocking)
        self._restol_sent = None

    def close(self):
        if getattr(self, 'ctx', None):
            for _, ref in list(getattr(self, '_futures', ())):   # residuals still on their way are collected while the records exist
                fut = ref()
                if fut is not None:
                    try:
                        fut.result()
                    except Exception:  # noqa: BLE001
                        pass
            self.lib.sdc_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        L.check(rc, self.ctx)

    # ---- setup ----
    def set_coeffs(self, Qmat, QI, QE, nodes, weights):
        Qm = np.ascontiguousarray(Qmat, dtype=np.float64)
        qi = np.ascontiguousarray(QI, dtype=np.float64)
        qe = None if QE is None else np.ascontiguousarray(QE, dtype=np.float64)
        nd = np.ascontiguousarray(nodes, dtype=np.float64)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if Qm.shape != (self.M + 1, self.M + 1) or qi.shape != Qm.shape:
            raise ParameterError(f'coefficient matrices must be {(self.M + 1, self.M + 1)}')
        self._chk(self.lib.sdc_set_coeffs(self.ctx, _dptr(Qm), _dptr(qi), None if qe is None else _dptr(qe),
                                          _dptr(nd), _dptr(w)))

    def set_stencil(self, which, offsets, weights):
        off = (C.c_int * len(offsets))(*[int(o) for o in offsets])
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self._chk(self.lib.sdc_set_stencil(self.ctx, which, len(offsets), off, _dptr(w)))

    def set_odd_interior(self, n_interior):
        """compact interior fields of a dirichlet-zero level inside slab fields of the odd extension's size
        (include/sdcmi.h: sdc_set_odd_interior)"""
        self._chk(self.lib.sdc_set_odd_interior(self.ctx, int(n_interior)))

    def set_banded_operator(self, cols, weights):
        """row table of a 1-D operator on a bounded grid (include/sdcmi.h: sdc_set_banded_operator)"""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if cols.shape != w.shape or cols.ndim != 2:
            raise ParameterError('cols and weights must be (n_interior, width) tables')
        self._chk(self.lib.sdc_set_banded_operator(self.ctx, cols.shape[0], cols.shape[1],
                                                   cols.ctypes.data_as(C.POINTER(C.c_int)), _dptr(w)))

    def set_symbol(self, which, table):
        t = np.ascontiguousarray(np.asarray(table, dtype=np.complex128)).view(np.float64)
        assert t.size == 2 * self.n
        self._chk(self.lib.sdc_set_symbol(self.ctx, which, _dptr(t)))

    def set_reaction(self, kind, p0, p1=0.0, nu=2):
        self._chk(self.lib.sdc_set_reaction(self.ctx, int(kind), float(p0), float(p1), int(nu)))

    def set_forcing_profile(self, profile):
        p = np.ascontiguousarray(profile, dtype=np.float64).reshape(-1)
        assert p.size == self.N
        self._chk(self.lib.sdc_set_forcing_profile(self.ctx, _dptr(p)))

    def set_forcing_values(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        assert g.size == self.M + 1
        self._chk(self.lib.sdc_set_forcing_values(self.ctx, _dptr(g)))

    def invalidate_spectra(self, which=3):
        self._chk(self.lib.sdc_invalidate_spectra(self.ctx, int(which)))

    def set_fused_residual(self, on):
        self._chk(self.lib.sdc_set_fused_residual(self.ctx, int(bool(on))))

    def set_skip_residual(self, on):
        """no stage after a sweep computes the residual (sweeper parameter skip_residual_computation)"""
        self._chk(self.lib.sdc_set_skip_residual(self.ctx, int(bool(on))))

    def set_virtual_sweeps(self, max_sweeps):
        """sweeps per step whose iterate is not stored but recomputed from the transform of u[0] (0: off)"""
        self._chk(self.lib.sdc_set_virtual_sweeps(self.ctx, int(max_sweeps)))

    def set_multiplier_table(self, from_sweep):
        """first sweep of a step that takes the node multipliers from a table instead of replaying earlier sweeps (0: never)"""
        self._chk(self.lib.sdc_set_multiplier_table(self.ctx, int(from_sweep)))

    def set_lazy_predictor_residual(self, on):
        """sdc_predict puts the norm-only transform behind the residual of a spread state off until sdc_residual asks for it"""
        self._chk(self.lib.sdc_set_lazy_predictor_residual(self.ctx, int(bool(on))))

    def residual_deferred(self):
        """True while residual() would have to do that transform first (include/sdcmi.h: sdc_residual_deferred)"""
        rc = self.lib.sdc_residual_deferred(self.ctx)
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def set_deferred(self, on):
        """leave F[1..M] / the spread copies unwritten until they are read (include/sdcmi.h: sdc_set_deferred)"""
        self._chk(self.lib.sdc_set_deferred(self.ctx, int(bool(on))))

    def set_timeslice_options(self, trail_sources=0, defer_last_pass=1, split_send=False):
        """how a time-parallel level deals with a u[0] that is replaced between sweeps (include/sdcmi.h); defer_last_pass: 0 =
        every pass at once, 1 = put off and, where nobody asks, run behind the next sweep's first launches, 2 = put off only"""
        self._chk(self.lib.sdc_set_timeslice_options(self.ctx, int(trail_sources), int(defer_last_pass), int(bool(split_send))))

    def set_early_end_point(self, on):
        self._chk(self.lib.sdc_set_early_end_point(self.ctx, int(bool(on))))

    def stream_wait_uend(self, stream_handle):
        """make another HIP stream (raw handle, e.g. torch.cuda.Stream().cuda_stream) wait until UEND is complete"""
        self._chk(self.lib.sdc_stream_wait_uend(self.ctx, C.c_void_p(int(stream_handle))))

    def set_keep_residual_fields(self, on):
        self._chk(self.lib.sdc_set_keep_residual_fields(self.ctx, int(bool(on))))

    def replace_u0(self, src_ptr):
        """u[0] <- src with the node norms of the residual updated in the same pass when the sweep kept the residual
        fields (include/sdcmi.h: sdc_replace_u0)"""
        self._chk(self.lib.sdc_replace_u0(self.ctx, src_ptr))

    def spectral_handover_ok(self):
        """this level sweeps in Fourier space: its hand-over may carry spectra (include/sdcmi.h: sdc_spectral_handover_ok)"""
        return bool(self.lib.sdc_spectral_handover_ok(self.ctx))

    def end_spectrum(self, stream_handle=None):
        p = self.lib.sdc_end_spectrum(self.ctx, C.c_void_p(int(stream_handle)) if stream_handle else None)
        if not p:
            self._chk(L.ERR_STATE)
        return p

    def spectrum_inbox(self):
        p = self.lib.sdc_spectrum_inbox(self.ctx)
        if not p:
            self._chk(L.ERR_NOMEM)
        return p

    def start_from_spectrum(self):
        self._chk(self.lib.sdc_start_from_spectrum(self.ctx))

    def replace_u0_spectrum(self):
        self._chk(self.lib.sdc_replace_u0_spectrum(self.ctx))

    def advance(self):
        """u[0] <- uend for the next time step on this level (include/sdcmi.h: sdc_advance)"""
        self._chk(self.lib.sdc_advance(self.ctx))

    def materialize(self, slot=-1, m=-1):
        self._chk(self.lib.sdc_materialize(self.ctx, int(slot), int(m)))

    def set_unlocked(self, unlocked=True):
        self._chk(self.lib.sdc_set_unlocked(self.ctx, int(bool(unlocked))))

    def set_spectral_reuse(self, on):
        self._chk(self.lib.sdc_set_spectral_reuse(self.ctx, int(bool(on))))

    def set_tau_active(self, active):
        self._chk(self.lib.sdc_set_tau_active(self.ctx, int(bool(active))))
        self.tau_active = bool(active)

    # ---- data ----
    def ptr(self, slot, m=0, comp=0):
        p = self.lib.sdc_slot_ptr(self.ctx, slot, m, comp)
        if not p:
            raise ParameterError(f'bad slot ({slot}, {m}, {comp})')
        return p

    def uend_address(self):
        """where the end value lies right now (include/sdcmi.h: sdc_uend_address; changes with every advance())"""
        return self.lib.sdc_uend_address(self.ctx)

    def end_value_generation(self):
        """> 0 while the end value still is the last node of the cached iterate (include/sdcmi.h)"""
        return int(self.lib.sdc_end_value_generation(self.ctx))

    def upload(self, slot, m, host, comp=0):
        h = np.ascontiguousarray(host, dtype=np.float64).reshape(-1)
        if h.size != self.N:
            raise ParameterError(f'expected {self.N} values, got {h.size}')
        self._chk(self.lib.sdc_upload(self.ctx, slot, m, comp, _dptr(h)))

    def download(self, slot, m=0, comp=0):
        out = np.empty(self.N, dtype=np.float64)
        self._chk(self.lib.sdc_download(self.ctx, slot, m, comp, _dptr(out)))
        return out.reshape(self.nvars)

    def download_u(self):
        return np.stack([self.download(L.SLOT_U, m) for m in range(self.M + 1)])

    def download_f(self):
        if self.ncomp == 1:
            return np.stack([self.download(L.SLOT_F, m) for m in range(self.M + 1)])
        return np.stack([np.stack([self.download(L.SLOT_F, m, c) for c in range(2)]) for m in range(self.M + 1)])

    # ---- sweep path ----
    def predict(self, t, dt, guess='spread', fill_u=0.0, fill_f=0.0):
        self._chk(self.lib.sdc_predict(self.ctx, t, dt, L.GUESS[guess], fill_u, fill_f))

    def sweep(self, t, dt):
        self._chk(self.lib.sdc_sweep(self.ctx, t, dt))

    def residual(self, dt, residual_type='full_abs'):
        if residual_type not in L.RES_TYPES:
            raise ParameterError(
                f'residual_type = {residual_type} not implemented, choose '
                f'full_abs, last_abs, full_rel or last_rel instead'
            )
        norms = np.zeros(self.M)
        res = C.c_double()
        self._retire_old_tickets()
        try:
            self._chk(self.lib.sdc_residual(self.ctx, dt, L.RES_TYPES[residual_type], _dptr(norms), C.byref(res)))
        finally:
            self._issued = int(self.lib.sdc_residual_last_ticket(self.ctx))   # (a ticket of the same ring, taken even by a call that failed)
        return res.value, norms

    def residual_route(self, dt):
        """where a residual asked for now would come from (include/sdcmi.h: sdc_residual_route); 3 = a pass over F in real
        space, the only route that can bring the quadrature sums along"""
        rc = self.lib.sdc_residual_route(self.ctx, dt)
        if rc < 0:
            self._chk(rc)
        return rc

    def residual_post(self, dt, residual_type='full_abs', restol=-1.0, integrals=None):
        """queue the residual of the current state and return a ResidualFuture at once (no synchronisation); restol is the
        tolerance the device takes its `converged` flag against.  integrals: M device addresses that receive the quadrature
        sums dt Q F (integrate()'s result) in the same pass when the residual is reduced from F in real space;
        `self.integrals_written` says whether they were"""
        if residual_type not in L.RES_TYPES:
            raise ParameterError(
                f'residual_type = {residual_type} not implemented, choose '
                f'full_abs, last_abs, full_rel or last_rel instead'
            )
        if restol != self._restol_sent:
            self._chk(self.lib.sdc_set_restol(self.ctx, float(restol)))
            self._restol_sent = restol
        self._retire_old_tickets()
        t = C.c_ulonglong()
        self.integrals_written = False
        if integrals is not None:
            wrote = C.c_int()
            dst = (C.c_void_p * self.M)(*[C.c_void_p(int(p)) for p in integrals])
            self._chk(self.lib.sdc_residual_post_integrals(self.ctx, dt, L.RES_TYPES[residual_type], dst, C.byref(wrote),
                                                           C.byref(t)))
            self.integrals_written = bool(wrote.value)
        else:
            self._chk(self.lib.sdc_residual_post(self.ctx, dt, L.RES_TYPES[residual_type], C.byref(t)))
        ticket = self._issued = t.value

        def fetch(block, self=self, ticket=ticket):
            if not self.ctx:
                raise RuntimeError('the engine of this residual is closed')
            norms = np.zeros(self.M)
            res, conv, ready = C.c_double(), C.c_int(), C.c_int()
            self._chk(self.lib.sdc_residual_wait(self.ctx, ticket, int(block), _dptr(norms), C.byref(res), C.byref(conv),
                                                 C.byref(ready)))
            return (res.value, norms, bool(conv.value)) if ready.value else None

        fut = ResidualFuture(fetch, restol)
        self._futures.append((ticket, weakref.ref(fut)))
        return fut

    def _retire_old_tickets(self, keep=200):
        """the library keeps the last 256 residual records: whoever still holds a ticket that the NEXT post would bring
        within 56 records of being overwritten gets its values now (its launch is long done).  Blocking residual() calls use
        tickets too (sdc_residual = post + wait), so the distance is counted in tickets, not in pending futures."""
        while self._futures and self._futures[0][0] <= self._issued + 1 - keep:
            _, ref = self._futures.popleft()
            old = ref()
            if old is not None:
                old.result()

    def end_point(self, dt, do_coll_update):
        self._chk(self.lib.sdc_end_point(self.ctx, dt, int(bool(do_coll_update))))

    def integrate(self, dt, dst_ptrs):
        arr = (C.c_void_p * self.M)(*dst_ptrs)
        self._chk(self.lib.sdc_integrate(self.ctx, dt, arr))

    # ---- problem-level ----
    def eval_f(self, u_ptr, g_t, fi_ptr, fe_ptr=None):
        self._chk(self.lib.sdc_eval_f(self.ctx, u_ptr, g_t, fi_ptr, fe_ptr))

    def eval_f_many(self, u_ptrs, fi_ptrs, fe_ptrs=None, g_ts=None):
        """eval_f of several fields in one pass of the launches (include/sdcmi.h: sdc_eval_f_batch)"""
        nf = len(u_ptrs)
        arr = lambda ps: (C.c_void_p * nf)(*[C.c_void_p(int(p)) for p in ps])   # noqa: E731
        g = None if g_ts is None else _dptr(np.ascontiguousarray(g_ts, dtype=np.float64))
        self._chk(self.lib.sdc_eval_f_batch(self.ctx, nf, arr(u_ptrs), g, arr(fi_ptrs), None if fe_ptrs is None else arr(fe_ptrs)))

    def solve(self, rhs_ptr, factor, out_ptr, guess_ptr=None):
        self._chk(self.lib.sdc_solve(self.ctx, rhs_ptr, factor, guess_ptr, out_ptr))

    def set_problem_vdp(self, mu, newton_tol, newton_maxiter):
        self._chk(self.lib.sdc_set_problem_vdp(self.ctx, float(mu), float(newton_tol), int(newton_maxiter)))

    def set_vdp_block_solver(self, kind):
        """'closed_form' (vector ALUs) or 'mfma' (matrix cores): include/sdcmi.h sdc_set_vdp_block_solver"""
        self._chk(self.lib.sdc_set_vdp_block_solver(self.ctx, {'closed_form': 0, 'mfma': 1}[kind]))

    def work_counters(self):
        out = (C.c_ulonglong * 5)()
        self._chk(self.lib.sdc_work_counters(self.ctx, out))
        return dict(newton=int(out[0]), rhs=int(out[1]), failed=int(out[2]), CG=int(out[3]), GMRES=int(out[4]))

    def set_solver(self, kind, rtol=1e-12, maxiter=10000):
        """'direct' (exact Fourier solve), 'CG' or 'GMRES' (include/sdcmi.h: sdc_set_solver)"""
        self._chk(self.lib.sdc_set_solver(self.ctx, {'direct': 0, 'CG': 1, 'GMRES': 2}[kind], float(rtol), int(maxiter)))

    # ---- vectors ----
    def vec_copy(self, n, x, y):
        self._chk(self.lib.sdc_vec_copy(self.ctx, n, x, y))

    def vec_fill(self, n, a, y):
        self._chk(self.lib.sdc_vec_fill(self.ctx, n, a, y))

    def vec_axpby(self, n, a, x, b, y, z):
        self._chk(self.lib.sdc_vec_axpby(self.ctx, n, a, x, b, y, z))

    def vec_amax(self, n, x):
        out = C.c_double()
        self._chk(self.lib.sdc_vec_amax(self.ctx, n, x, C.byref(out)))
        return out.value

    # ---- misc ----
    def sync(self):
        self._chk(self.lib.sdc_sync(self.ctx))

    def timer_begin(self):
        self._chk(self.lib.sdc_timer_begin(self.ctx))

    def timer_end(self):
        ms = C.c_double()
        self._chk(self.lib.sdc_timer_end(self.ctx, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._chk(self.lib.sdc_profile_enable(self.ctx, int(on)))

    def profile_read(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = np.zeros(cap)
        calls = (C.c_int * cap)()
        count = C.c_int()
        self._chk(self.lib.sdc_profile_read(self.ctx, cap, names, _dptr(ms), calls, C.byref(count)))
        return {names[i].decode(): (float(ms[i]), int(calls[i])) for i in range(count.value)}

    @property
    def device_bytes(self):
        return int(self.lib.sdc_ctx_bytes(self.ctx))
