import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from pysdc_amd import lib as L
from pysdc_amd.engine import SweepEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
print('free', torch.cuda.mem_get_info()[0] / 1e9, flush=True)
e = SweepEngine((n, n, n), 5)
print('created, bytes', e.device_bytes / 1e9, 'free', torch.cuda.mem_get_info()[0] / 1e9, flush=True)
p = e.ptr(L.SLOT_U, 1)
print('mapped, bytes', e.device_bytes / 1e9, 'free', torch.cuda.mem_get_info()[0] / 1e9, flush=True)
e.sync()
print('amax', e.vec_amax(e.N, e.ptr(L.SLOT_F, 3)), flush=True)
e.close()
print('closed, free', torch.cuda.mem_get_info()[0] / 1e9, flush=True)
