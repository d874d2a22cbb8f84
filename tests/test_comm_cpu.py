"""The exchange patterns of the C-ABI communicator (include/sdcmi.h: sdc_comm_*) without a GPU: sdc_comm_selftest runs
the direct hand-over, the two-hop hand-over (controller_MPI.py:218-305 carried by all ranks together) and the mesh
broadcast (controller_MPI.py:125-130) on plain host buffers over the shared-memory mailbox wire and checks every value
that arrives - ranks as threads of this process and as separate processes, message lengths that do not divide by the
number of ranks, several rounds over the same mailboxes."""
import os
import subprocess
import sys
import threading

import pytest

from pysdc_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _threads(P, n, what, arg, rounds=3):
    lib = L.load()
    job = f'cpu{os.getpid()}_{P}_{n}_{what}_{arg}'.encode()
    rcs, errs = [None] * P, [None] * P

    def rank(r):
        rcs[r] = lib.sdc_comm_selftest(job, P, r, n, what, arg, rounds)
        if rcs[r] != 0:
            errs[r] = lib.sdc_last_error(None)

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    return rcs, errs


@pytest.mark.parametrize('P', [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize('n', [1, 7, 64, 1000, 12345])
def test_patterns_between_threads(P, n):
    for what, arg in ((0, 0), (1, 0), (1, max(1, P - 1)), (2, 0), (2, P - 1), (2, P // 2)):
        rcs, errs = _threads(P, n, what, arg)
        assert rcs == [0] * P, (what, arg, errs)
    assert not [f for f in os.listdir('/dev/shm') if f.startswith(f'sdcmi.cpu{os.getpid()}_')]   # mailboxes removed


def test_patterns_between_processes():
    """world size 2 and 3 as real processes (what the one-GPU rehearsal of bench.py uses between its rank processes)"""
    code = ('import sys; sys.path.insert(0, %r); from pysdc_amd import lib as L; lib = L.load(); '
            'job, P, r = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3]); '
            'rc = [lib.sdc_comm_selftest(job + str(w).encode(), P, r, 100003, w, a, 4) for w, a in ((0, 0), (1, 0), (2, P - 1))]; '
            'print(rc, lib.sdc_last_error(None)); sys.exit(0 if rc == [0, 0, 0] else 1)') % ROOT
    for P in (2, 3):
        procs = [subprocess.Popen([sys.executable, '-c', code, f'proc{os.getpid()}_{P}_', str(P), str(r)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(P)]
        outs = [p.communicate(timeout=120)[0] for p in procs]
        assert [p.returncode for p in procs] == [0] * P, outs


def test_a_missing_peer_times_out_with_an_error():
    lib = L.load()
    os.environ['SDC_COMM_TIMEOUT'] = '0.3'
    try:
        rc = lib.sdc_comm_selftest(f'lonely{os.getpid()}'.encode(), 2, 1, 16, 0, 0, 1)   # rank 1 waits for a rank 0 that never comes
    finally:
        del os.environ['SDC_COMM_TIMEOUT']
    assert rc == L.ERR_COMM
    assert b'timed out' in lib.sdc_last_error(None)
    for f in os.listdir('/dev/shm'):
        if f.startswith(f'sdcmi.lonely{os.getpid()}'):
            os.unlink(os.path.join('/dev/shm', f))


def test_same_host_probe_of_the_pinned_host_path():
    """the pinned-host share of two-rank hand-overs is switched on only between ranks that have been seen to share /dev/shm
    (pysdc_amd/controller.py: _ranks_share_host): the probe file is found by a rank of this host, not by one that looks for it
    somewhere else; any failure on the way means 'no'"""
    import os

    from pysdc_amd.comm import ranks_share_host_memory

    assert ranks_share_host_memory(0, lambda obj: obj, lambda flag: flag) is True
    assert ranks_share_host_memory(0, lambda obj: ('/dev/shm/pysdc_amd.probe.none', obj[1]), lambda flag: flag) is False
    assert ranks_share_host_memory(0, lambda obj: (obj[0], 'another nonce'), lambda flag: flag) is False
    assert ranks_share_host_memory(1, lambda obj: ('/dev/shm/pysdc_amd.probe.none', 'x'), lambda flag: flag) is False

    def broken(obj):
        raise RuntimeError('no host group')

    assert ranks_share_host_memory(0, broken, lambda flag: flag) is False
    assert not [f for f in os.listdir('/dev/shm') if f.startswith('pysdc_amd.probe.')]
