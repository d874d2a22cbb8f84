"""bench.py as its own launcher, without a GPU: `python bench.py --gpus 2` starts the rank processes itself; here they cannot
initialise a device, so the job must end quickly with ONE JSON error line (rank output tails included) and a non-zero exit
code - the behaviour the driver relies on when a multi-GPU run cannot start.  (The successful path runs in the GPU suite:
tests/test_gpu_comm.py::test_bench_two_ranks_on_one_gpu_end_to_end.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch

    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason='the failure path is what a box without GPUs shows')
def test_launcher_reports_ranks_that_cannot_start():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--n', '64', '--backend', 'gloo',
                          '--same-device', '--job-timeout', '120'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert 'error' in rec and rec['n_gpus'] == 2 and set(rec['rank_output_tails']) == {'rank0', 'rank1'}


def test_flag_combinations_that_cannot_work_are_refused_at_once():
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--same-device'], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120, cwd=ROOT)
    assert res.returncode == 2 and 'error' in json.loads(res.stdout.strip().splitlines()[-1])
