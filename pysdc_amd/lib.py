"""ctypes binding of libsdcmi.so (C-ABI: include/sdcmi.h).  No fallback: if the HIP library is missing or a
call fails this raises - the product path never computes on the CPU."""
import ctypes as C
import os

from pysdc_amd.errors import CommunicationError, EngineError, ParameterError, ProblemError, UnlockError

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PYSDC_AMD_LIB: another build of the same library, e.g. one compiled with experiment macros - scripts/README.md)
LIB_PATH = os.environ.get('PYSDC_AMD_LIB') or os.path.join(_HERE, 'libsdcmi.so')

SLOT_U, SLOT_F, SLOT_TAU, SLOT_UEND, SLOT_WORK = 0, 1, 2, 3, 4
RES_TYPES = {'full_abs': 0, 'last_abs': 1, 'full_rel': 2, 'last_rel': 3}
GUESS = {'spread': 0, 'copy': 1, 'zero': 2, 'random': 3}
EXPL_NONE, EXPL_STENCIL, EXPL_FORCING, EXPL_REACTION, EXPL_SYMBOL = 0, 1, 2, 3, 4

ERR_PARAM, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED, ERR_NOMEM, ERR_NEWTON, ERR_COMM = -1, -2, -3, -4, -5, -6, -7

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/sdcmi.h declares (tests/test_abi.py checks)
PROTOTYPES = {
    'sdc_version': (C.c_int, []),
    'sdc_ctx_create': (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    'sdc_ctx_destroy': (C.c_int, [_vp]),
    'sdc_last_error': (C.c_char_p, [_vp]),
    'sdc_ctx_bytes': (C.c_size_t, [_vp]),
    'sdc_set_coeffs': (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp]),
    'sdc_set_stencil': (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), _dp]),
    'sdc_set_symbol': (C.c_int, [_vp, C.c_int, _dp]),
    'sdc_set_reaction': (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_int]),
    'sdc_set_expl_kind': (C.c_int, [_vp, C.c_int]),
    'sdc_set_forcing_profile': (C.c_int, [_vp, _dp]),
    'sdc_set_forcing_values': (C.c_int, [_vp, _dp]),
    'sdc_slot_ptr': (_vp, [_vp, C.c_int, C.c_int, C.c_int]),
    'sdc_uend_address': (_vp, [_vp]),
    'sdc_end_value_generation': (C.c_longlong, [_vp]),
    'sdc_upload': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    'sdc_download': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    'sdc_set_tau_active': (C.c_int, [_vp, C.c_int]),
    'sdc_invalidate_spectra': (C.c_int, [_vp, C.c_int]),
    'sdc_set_spectral_reuse': (C.c_int, [_vp, C.c_int]),
    'sdc_set_unlocked': (C.c_int, [_vp, C.c_int]),
    'sdc_set_fused_residual': (C.c_int, [_vp, C.c_int]),
    'sdc_set_skip_residual': (C.c_int, [_vp, C.c_int]),
    'sdc_set_virtual_sweeps': (C.c_int, [_vp, C.c_int]),
    'sdc_set_multiplier_table': (C.c_int, [_vp, C.c_int]),
    'sdc_set_lazy_predictor_residual': (C.c_int, [_vp, C.c_int]),
    'sdc_set_restol': (C.c_int, [_vp, C.c_double]),
    'sdc_residual_post': (C.c_int, [_vp, C.c_double, C.c_int, C.POINTER(C.c_ulonglong)]),
    'sdc_residual_post_integrals': (C.c_int, [_vp, C.c_double, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int),
                                              C.POINTER(C.c_ulonglong)]),
    'sdc_residual_wait': (C.c_int, [_vp, C.c_ulonglong, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'sdc_residual_deferred': (C.c_int, [_vp]),
    'sdc_residual_route': (C.c_int, [_vp, C.c_double]),
    'sdc_set_timeslice_options': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    'sdc_residual_last_ticket': (C.c_ulonglong, [_vp]),
    'sdc_set_deferred': (C.c_int, [_vp, C.c_int]),
    'sdc_set_solver': (C.c_int, [_vp, C.c_int, C.c_double, C.c_int]),
    'sdc_set_banded_operator': (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), _dp]),
    'sdc_solve_jacobian': (C.c_int, [_vp, _vp, C.c_double, _vp, _vp]),
    'sdc_advance': (C.c_int, [_vp]),
    'sdc_defer_f0': (C.c_int, [_vp]),
    'sdc_set_keep_residual_fields': (C.c_int, [_vp, C.c_int]),
    'sdc_set_early_end_point': (C.c_int, [_vp, C.c_int]),
    'sdc_stream_wait_uend': (C.c_int, [_vp, _vp]),
    'sdc_replace_u0': (C.c_int, [_vp, _vp]),
    'sdc_spectral_handover_ok': (C.c_int, [_vp]),
    'sdc_set_wire_spectral': (C.c_int, [_vp, C.c_int]),
    'sdc_end_spectrum': (_vp, [_vp, _vp]),
    'sdc_spectrum_inbox': (_vp, [_vp]),
    'sdc_replace_u0_spectrum': (C.c_int, [_vp]),
    'sdc_start_from_spectrum': (C.c_int, [_vp]),
    'sdc_comm_unique_id': (C.c_int, [C.c_char_p]),
    'sdc_comm_init': (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    'sdc_comm_attach': (C.c_int, [_vp, _vp]),
    'sdc_comm_destroy': (C.c_int, [_vp]),
    'sdc_comm_exchange': (C.c_int, [_vp, C.c_int, C.c_int]),
    'sdc_send_uend': (C.c_int, [_vp, C.c_int]),
    'sdc_recv_u0': (C.c_int, [_vp, C.c_int]),
    'sdc_bcast': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    'sdc_comm_handover_post': (C.c_int, [_vp, C.c_int]),
    'sdc_comm_handover_complete': (C.c_int, [_vp]),
    'sdc_comm_bcast_buffer': (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    'sdc_comm_bcast_end_spectrum': (C.c_int, [_vp, C.c_int]),
    'sdc_comm_set_chunk': (C.c_int, [_vp, C.c_size_t]),
    'sdc_comm_set_relay': (C.c_int, [_vp, C.c_int]),
    'sdc_comm_set_host_share': (C.c_int, [_vp, C.c_double]),
    'sdc_comm_set_format': (C.c_int, [_vp, C.c_int]),
    'sdc_comm_info': (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_ulonglong),
                                C.POINTER(C.c_ulonglong), C.c_char_p]),
    'sdc_comm_selftest': (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    'sdc_comm_sync': (C.c_int, [_vp]),
    'sdc_fft_prolong': (C.c_int, [_vp, _vp, _vp, _vp, C.c_double]),
    'sdc_materialize': (C.c_int, [_vp, C.c_int, C.c_int]),
    'sdc_init_field': (C.c_int, [_vp, _vp, C.POINTER(C.c_int), C.c_double, C.c_ulonglong]),
    'sdc_predict': (C.c_int, [_vp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]),
    'sdc_sweep': (C.c_int, [_vp, C.c_double, C.c_double]),
    'sdc_residual': (C.c_int, [_vp, C.c_double, C.c_int, _dp, _dp]),
    'sdc_end_point': (C.c_int, [_vp, C.c_double, C.c_int]),
    'sdc_integrate': (C.c_int, [_vp, C.c_double, C.POINTER(_vp)]),
    'sdc_eval_f': (C.c_int, [_vp, _vp, C.c_double, _vp, _vp]),
    'sdc_eval_f_batch': (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _dp, C.POINTER(_vp), C.POINTER(_vp)]),
    'sdc_solve': (C.c_int, [_vp, _vp, C.c_double, _vp, _vp]),
    'sdc_vec_copy': (C.c_int, [_vp, C.c_size_t, _vp, _vp]),
    'sdc_vec_fill': (C.c_int, [_vp, C.c_size_t, C.c_double, _vp]),
    'sdc_vec_axpby': (C.c_int, [_vp, C.c_size_t, C.c_double, _vp, C.c_double, _vp, _vp]),
    'sdc_vec_amax': (C.c_int, [_vp, C.c_size_t, _vp, _dp]),
    'sdc_vec_box': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                              C.POINTER(C.c_longlong), _vp, _vp, C.c_int, C.c_double]),
    'sdc_set_problem_vdp': (C.c_int, [_vp, C.c_double, C.c_double, C.c_int]),
    'sdc_set_vdp_block_solver': (C.c_int, [_vp, C.c_int]),
    'sdc_work_counters': (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    'sdc_transfer_apply_batch': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    'sdc_transfer_apply_batch_acc': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    'sdc_transfer_apply': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    'sdc_transfer_apply_nested': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    'sdc_odd_mirror': (C.c_int, [_vp, _vp, C.c_int]),
    'sdc_set_odd_interior': (C.c_int, [_vp, C.c_int]),
    'sdc_odd_extend': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    'sdc_odd_extract': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    'sdc_sync': (C.c_int, [_vp]),
    'sdc_timer_begin': (C.c_int, [_vp]),
    'sdc_timer_end': (C.c_int, [_vp, _dp]),
    'sdc_profile_enable': (C.c_int, [_vp, C.c_int]),
    'sdc_profile_read': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load libsdcmi.so (once).  torch, when installed, is imported first so that the library binds to the
    one HIP runtime already in the process (same SONAME, libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback.'
        )
    try:
        import torch  # noqa: F401  (plumbing only: shares the HIP runtime / streams)
    except Exception:  # pragma: no cover
        pass
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise EngineError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EngineError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc == 0:
        return
    lib = load()
    msg = lib.sdc_last_error(ctx)
    msg = msg.decode() if msg else f'libsdcmi error {rc}'
    if rc == ERR_PARAM:
        raise ParameterError(msg)
    if rc == ERR_STATE:
        raise UnlockError(msg)
    if rc == ERR_NEWTON:
        raise ProblemError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    if rc == ERR_COMM:
        raise CommunicationError(msg)
    raise EngineError(msg)
