"""GPU diagnostics: where does host time go, and first per-kernel timings."""
import sys, time
t0 = time.time()
import numpy as np
sys.path.insert(0, '.')
from pysdc_amd import lib as L
from pysdc_amd.engine import SweepEngine
from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS
from pysdc_amd.fd import periodic_operator_stencil, grid_1d
print('imports', time.time() - t0, flush=True)


def mk(nvars, M=5, QI='IE'):
    t = time.time()
    e = SweepEngine(nvars, M, 1)
    print('  create', nvars, time.time() - t, 'bytes', e.device_bytes / 1e9, flush=True)
    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    qi = np.zeros_like(c.Qmat)
    qi[1:, 1:] = QDELTA_GENERATORS[QI](qGen=c.generator, tLeft=0).genCoeffs()
    e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
    dx, _ = grid_1d(nvars[0], 'periodic')
    t = time.time()
    e.set_stencil(0, *periodic_operator_stencil(2, 2, 'center', dx, 0.1))
    print('  set_stencil', time.time() - t, flush=True)
    return e


for nv in [(8, 8, 8), (16, 16), (64,)]:
    t = time.time()
    e = mk(nv)
    e.upload(L.SLOT_U, 0, np.random.default_rng(0).standard_normal(nv))
    t1 = time.time(); e.predict(0.0, 1e-3); e.sync(); print('  predict', time.time() - t1)
    t1 = time.time(); e.sweep(0.0, 1e-3); e.sync(); print('  sweep1', time.time() - t1)
    t1 = time.time(); e.sweep(0.0, 1e-3); e.sync(); print('  sweep2', time.time() - t1)
    t1 = time.time(); r = e.residual(1e-3); print('  residual', time.time() - t1, r[0])
    t1 = time.time(); e.download_u(); print('  download_u', time.time() - t1)
    t1 = time.time(); e.close(); print('  close', time.time() - t1)
    print(nv, 'total', time.time() - t, flush=True)

for n in (256, 512):
    dt = 1e-3 * (512 / n) ** 2
    e = mk((n, n, n))
    N = n**3
    # device-side init: sin mode + noise is not needed for timing; use host upload of random data in chunks
    rng = np.random.default_rng(0)
    e.upload(L.SLOT_U, 0, rng.standard_normal(N))
    e.predict(0.0, dt); e.sync()
    e.sweep(0.0, dt); e.sync()
    e.profile_enable(True)
    for _ in range(3):
        e.sweep(0.0, dt)
        e.residual(dt)
    prof = e.profile_read()
    e.profile_enable(False)
    tot = 0
    for k, (ms, calls) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print(f'  {k:14s} {ms / calls:9.3f} ms/call x{calls}')
        tot += ms
    print(f'n={n}: profiled total per sweep+residual {tot / 3:.2f} ms')
    e.timer_begin()
    K = 5
    for _ in range(K):
        e.sweep(0.0, dt)
    ms = e.timer_end() / K
    B = 8 * N * 16
    print(f'n={n}: sweep {ms:.2f} ms  -> floor-bytes {B / 1e9:.2f} GB -> {B / ms / 1e6:.1f} GB/s (floor-only); '
          f'71 word-passes -> {8 * N * 71 / ms / 1e6:.1f} GB/s')
    e.close()
