"""Achievable HBM bandwidth on this GPU with plain streaming kernels of the library (to quote beside the nominal
8 TB/s): fill (write only), amax (read only), device-to-device copy (1R:1W), axpby (2R:1W).  8.6 GB operands."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pysdc_amd import lib as Lb

lib = Lb.load()
n = 1024**3
x = torch.empty(n, dtype=torch.float64, device='cuda')
y = torch.empty(n, dtype=torch.float64, device='cuda')
z = torch.empty(n, dtype=torch.float64, device='cuda')
out = C.c_double()


def run(name, fn, nbytes, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    return name, {'ms': 1e3 * dt, 'GB/s': nbytes / dt / 1e9}


res = dict([
    run('fill (write only)', lambda: lib.sdc_vec_fill(None, n, 1.0, x.data_ptr()), 8 * n),
    run('fill y', lambda: lib.sdc_vec_fill(None, n, 2.0, y.data_ptr()), 8 * n),
    run('amax (read only)', lambda: lib.sdc_vec_amax(None, n, x.data_ptr(), C.byref(out)), 8 * n),
    run('copy (1R:1W)', lambda: lib.sdc_vec_copy(None, n, x.data_ptr(), z.data_ptr()), 16 * n),
    run('axpby (2R:1W)', lambda: lib.sdc_vec_axpby(None, n, 1.0, x.data_ptr(), 2.0, y.data_ptr(), z.data_ptr()), 24 * n),
    run('torch copy_ (1R:1W)', lambda: z.copy_(x), 16 * n),
])
print(json.dumps(res, indent=1))
