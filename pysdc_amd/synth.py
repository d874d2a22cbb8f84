"""Host equivalent of the device-side synthetic field generator (``sdc_init_field``, include/sdcmi.h):
prod_d sin(pi*freq_d*x_d) on the reference's grid (generic_ND_FD.py:171-180) + amp * N(0,1) drawn from
splitmix64(seed, i) and Box-Muller.  Used by tests to check the device generator and by the CPU baseline
so that CPU and GPU legs see the same input without shipping multi-GB arrays."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def init_field(nvars, freq, amp=0.0, seed=0):
    nvars = (nvars,) if isinstance(nvars, int) else tuple(nvars)
    ndim, n = len(nvars), nvars[0]
    freq = (freq,) * ndim if isinstance(freq, int) else tuple(freq)
    x = np.arange(n) * (1.0 / n)
    if ndim == 1:
        v = np.sin(np.pi * freq[0] * x)
    elif ndim == 2:
        v = np.sin(np.pi * freq[0] * x[None, :]) * np.sin(np.pi * freq[1] * x[:, None])
    else:
        v = (np.sin(np.pi * freq[0] * x[None, :, None]) * np.sin(np.pi * freq[1] * x[:, None, None])
             * np.sin(np.pi * freq[2] * x[None, None, :]))
    v = np.array(v, dtype=np.float64)
    if amp != 0.0:
        i = np.arange(v.size, dtype=np.uint64)
        with np.errstate(over='ignore'):
            base = (np.uint64(seed) * np.uint64(0x100000001B3)) & _M64
            h1 = _splitmix64((base + np.uint64(2) * i) & _M64)
            h2 = _splitmix64((base + np.uint64(2) * i + np.uint64(1)) & _M64)
        u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
        u2 = ((h2 >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
        g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        v = v + amp * g.reshape(v.shape)
    return v
