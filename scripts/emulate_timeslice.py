"""Device work of ONE time slice of a multi-GPU multi-step SDC run, measured on a single GPU: per iteration
sweep -> end point -> u[0] replaced by a received value (here: a local buffer) -> residual against it.  The message
itself (8 N bytes over xGMI) is not part of this number.  Usage: python scripts/emulate_timeslice.py [n] [iters]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pysdc_amd import lib as L
from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS
from pysdc_amd.engine import SweepEngine
from pysdc_amd.hip_mesh import _CAI
from pysdc_amd import fd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M = 5
dt = 2.5e-4 * (1024.0 / n) ** 2
c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
QI = np.zeros_like(c.Qmat)
QI[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
out = {}
copies = int(sys.argv[3]) if len(sys.argv) > 3 else 8       # the "message": that many 8 N byte device copies
side = torch.cuda.Stream()
msg_ms = float(os.environ.get('EMU_MSG_MS', '0'))      # spectral variant: the message as a wait of that length on its own stream
sleep_cycles_per_ms = 0.0
if msg_ms > 0:
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    t_ = time.perf_counter()
    torch.cuda._sleep(200_000_000)
    torch.cuda.synchronize()
    sleep_cycles_per_ms = 200_000_000 / (1e3 * (time.perf_counter() - t_))
variants = tuple(os.environ.get('EMU_VARIANTS', 'spectral,overlap,fields,recompute').split(','))
for keep in [dict(spectral='spectral', overlap='overlap', fields=True, recompute=False)[v] for v in variants]:
    e = SweepEngine((n, n, n), M)
    e.set_coeffs(c.Qmat, QI, None, c.nodes, c.weights)
    e.set_stencil(0, *fd.periodic_operator_stencil(2, 2, 'center', 1.0 / n, 0.1))
    e.set_keep_residual_fields(bool(keep) and keep != 'spectral')
    e.set_early_end_point(keep in ('overlap', 'spectral'))
    if keep == 'spectral':
        # the wire carries spectra: what a communicator in spectral format does, with device copies standing in for the
        # message (include/sdcmi.h: sdc_end_spectrum -> sdc_spectrum_inbox -> sdc_replace_u0_spectrum)
        assert e.spectral_handover_ok()
        L.check(e.lib.sdc_set_wire_spectral(e.ctx, 1), e.ctx)
    freq = (C.c_int * 3)(2, 2, 2)
    L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 1e-3, 0), e.ctx)
    e.invalidate_spectra(1)
    inbox = torch.empty(e.N, dtype=torch.float64, device='cuda')
    e.vec_copy(e.N, e.ptr(L.SLOT_U, 0), inbox.data_ptr())
    e.predict(0.0, dt)
    e.residual(dt)
    e.sweep(0.0, dt)                        # warm-up: allocates the spectral cache
    e.residual(dt)
    e.end_point(dt, False)
    if keep != 'spectral':                  # (a FIELD as the new start value: the spectral slice never sees one - it would make
        e.replace_u0(inbox.data_ptr())      #  the engine store the iterate and allocate the node spectra for nothing)
        e.residual(dt)
    uend = torch.as_tensor(_CAI(e.ptr(L.SLOT_UEND), e.N, e), device='cuda')
    nspec = 2 * (n // 2 + 1) * n * n
    futures, inbox_free, sleeps = [], None, []
    if keep == 'spectral':
        opts = [int(v) for v in os.environ.get('EMU_OPTS', '5,1,0').split(',')]   # trail sources, deferred last pass, split send
        e.set_timeslice_options(*opts)
    # (one untimed iteration first: buffers that only this data flow needs - the spectrum inbox, the difference spectrum - are
    # allocated by its first use, and an 8.6 GB hipMalloc can take 100 ms on a fresh box)
    K = int(os.environ.get('EMU_SWEEPS', '4'))    # sweeps per block (the bench's 4): every block starts from a spread predictor
    for k in range(iters + (K if keep == 'spectral' else 1)):
        if k == (K if keep == 'spectral' else 1):
            for f in futures:
                f.result()
            futures = []
            e.profile_read()
            e.profile_enable(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if keep == 'spectral' and k % K == 0:
            # a new block: the slice starts from the end value of the block before as a spectrum (here: its own), predicts,
            # and - like every rank of a block - skips the first hand-over (all of them hold the same start value)
            e.end_point(dt, False)
            e.advance()
            e.predict(0.0, dt)
            futures.append(e.residual_post(dt))
        e.sweep(0.0, dt)
        if keep == 'spectral':
            e.end_point(dt, False)          # put off: the end value is the last node's spectrum
            src = torch.as_tensor(_CAI(e.end_spectrum(side.cuda_stream), nspec, e), device='cuda')
            dst = torch.as_tensor(_CAI(e.spectrum_inbox(), nspec, e), device='cuda')
            e.invalidate_spectra(8)         # (what the communicator does: the end value existed for the wire only)
            with torch.cuda.stream(side):   # the message, posted behind the launch that wrote the last node's spectrum only
                if inbox_free is not None:  # (... and behind the engine's last use of the buffer it lands in: sdc_comm's inbox_free)
                    side.wait_event(inbox_free)
                if msg_ms > 0:              # a message that takes msg_ms to arrive and uses none of this GPU's bandwidth on the way
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                    torch.cuda._sleep(int(msg_ms * sleep_cycles_per_ms))
                    ev[1].record()
                    sleeps.append(ev)       # (the spin counts shader clocks, which drop under load: what it took is measured)
                for _ in range(copies):
                    dst.copy_(src)
            futures.append(e.residual_post(dt))   # IT_FINE: queued; its last pass waits for the receive (defer_last_pass)
            torch.cuda.current_stream().wait_stream(side)
            e.replace_u0_spectrum()         # what arrives
            inbox_free = torch.cuda.Event()
            inbox_free.record()
            futures.append(e.residual_post(dt))   # IT_CHECK
            continue
        if keep == 'overlap':
            e.end_point(dt, False)          # free: the sweep produced UEND right after the spectral update
            e.stream_wait_uend(side.cuda_stream)
            with torch.cuda.stream(side):   # the message, posted behind UEND only
                for _ in range(copies):
                    inbox.copy_(uend)
            e.residual(dt)                  # IT_FINE, while the message travels
            torch.cuda.current_stream().wait_stream(side)
        else:
            e.residual(dt)                  # IT_FINE
            e.end_point(dt, False)          # what is sent
            e.materialize(L.SLOT_UEND, 0)   # (a raw pointer is read below: the end value has to be there for real)
            for _ in range(copies):
                inbox.copy_(uend)
        e.replace_u0(inbox.data_ptr())      # what arrives
        e.residual(dt)                      # IT_CHECK
    for f in futures:                       # (whoever logged them collects them at the end of the run)
        f.result()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / iters
    prof = e.profile_read()
    dev_bytes = e.device_bytes
    if keep == 'spectral':
        e.end_point(dt, False)
    e.materialize(L.SLOT_UEND, 0)
    if keep == 'spectral':   # the new start value, brought back to real space, is the last end value
        check = float(torch.max(torch.abs(torch.as_tensor(_CAI(e.ptr(L.SLOT_U, 0), e.N, e), device='cuda') - uend)))
    else:
        check = float(torch.max(torch.abs(inbox - uend)))       # the last message is the last end value
    out[{'spectral': 'spectra_on_the_wire', 'overlap': 'overlapped_message', True: 'kept_residual_fields',
         False: 'recomputed_residual'}[keep]] = {
        'ms_per_iteration': 1e3 * el, 'message_copies': copies, 'message_ms': msg_ms,
        'message_ms_measured': round(sum(a.elapsed_time(b) for a, b in sleeps[-iters:]) / max(1, len(sleeps[-iters:])), 2) if sleeps else 0.0, 'options': os.environ.get('EMU_OPTS', '5,1,0') if keep == 'spectral' else None,
        'device_bytes': dev_bytes, 'inbox_minus_uend': check, 'kernels_ms': {k: round(v[0] / v[1], 2) for k, v in prof.items() if v[1]},
        'kernels_ms_per_iteration': {k: round(v[0] / iters, 2) for k, v in prof.items() if v[1]}}
    e.close()
print(json.dumps(out))
