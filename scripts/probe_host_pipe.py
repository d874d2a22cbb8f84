"""Rate of the second path of a two-rank hand-over (csrc/comm.hpp: HostPipe - pinned host memory beside the one xGMI link),
measured with two thread ranks on ONE GPU: the sender's device-to-host copies and the receiver's host-to-device copies run at
the same time over the box's host link (full duplex), as they would over the two links of two GPUs.  The share that stays
on the "wire" is a device-to-device copy through the shared-memory mailbox here and says nothing about xGMI.
Usage: python scripts/probe_host_pipe.py [n] [share] [rounds]"""
import json
import os
import sys
import threading
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pysdc_amd import lib as L
from pysdc_amd.comm import DeviceComm, shm_unique_id
from tests import _gpu as G
from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.999
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
uid = shm_unique_id()
out, errors = [None, None], []


def body(r):
    try:
        e = G.engine_for('heat_unforced', dict(nvars=(n, n, n), nu=0.1), 1)
        c = CollBase(1, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
        qi = np.zeros_like(c.Qmat)
        qi[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        comm = DeviceComm(e, 2, r, uid=uid)
        comm.set_host_share(share)
        e.vec_fill(e.N, 1.0 + r, e.ptr(L.SLOT_UEND, 0))
        times = []
        for k in range(rounds + 1):
            comm.sync()
            t0 = time.perf_counter()
            comm.handover_post(2)
            comm.handover_complete()
            comm.sync()
            e.sync()
            times.append(time.perf_counter() - t0)
        ok = float(np.max(np.abs(e.download(L.SLOT_U, 0) - (1.0 if r == 1 else 0.0)))) if r == 1 else 0.0
        out[r] = (times[1:], ok)
        comm.close()
        e.close()
    except Exception:  # noqa: BLE001
        errors.append(traceback.format_exc())


ts = [threading.Thread(target=body, args=(r,)) for r in range(2)]
for t in ts:
    t.start()
for t in ts:
    t.join(600)
if errors:
    print(errors[0])
    sys.exit(1)
nbytes = 8.0 * n**3
best = min(out[1][0])
print(json.dumps({'n': n, 'message_GB': nbytes / 1e9, 'host_share': share, 'rounds': rounds,
                  'seconds_per_handover': [round(t, 4) for t in out[1][0]], 'best_s': round(best, 4),
                  'GBps_of_the_whole_message': round(nbytes / best / 1e9, 1),
                  'GBps_host_path': round(share * nbytes / best / 1e9, 1), 'max_error_on_receiver': out[1][1]}))
