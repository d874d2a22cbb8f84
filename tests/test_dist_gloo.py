"""CPU, world_size 2, gloo: the one-time-step-per-rank controller (pysdc_amd.controller.controller_dist)
against golden multi-step runs of the reference's serial controller (tests/golden/runs.npz).  The reference
applies the same oracle: MPI and non-MPI controllers must agree (tests/test_tutorials/test_step_6.py:20-42)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests._cases import load_cases, make_oracle_problem, make_oracle_coll, rel_err


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, outdir, fname='runs.npz', relay='1'):
    import torch.distributed as dist

    os.environ['PYSDC_AMD_RELAY'] = relay
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import sdc_oracle as O
        from pysdc_amd.controller import controller_dist
        from pysdc_amd.stats import get_sorted
        from tests._oracle_step import OracleStep, np_mesh

        case = load_cases(fname)[name]
        meta = case['meta']
        lp = meta['level_params']
        coll = make_oracle_coll(case)

        def factory():
            prob = make_oracle_problem(meta['prob'], meta['prob_params'])
            return O.Level(prob, coll, lp['dt'], restol=lp.get('restol', -1.0), nsweeps=lp.get('nsweeps', 1))

        desc = dict(step_class=OracleStep, oracle_level_factory=factory, level_params=lp,
                    step_params=dict(maxiter=meta['maxiter']))
        if world > 1:   # only the world group carries a run (peers are global ranks): sub-groups are refused, on every rank
            from pysdc_amd.errors import ParameterError

            sub = dist.new_group(ranks=[0])
            try:
                controller_dist(dict(logger_level=40), desc, comm=sub)
                raise AssertionError('sub-group accepted')
            except ParameterError:
                pass
        C = controller_dist(dict(logger_level=40, **meta['controller_params']), desc)
        shape = factory().prob.nvars
        u0 = np_mesh(np.array(case['u0']).reshape(shape))
        uend, stats = C.run(u0, meta['t0'], meta['Tend'])
        niter = get_sorted(stats, type='niter', sortby='time')
        np.savez(os.path.join(outdir, f'r{rank}.npz'), uend=np.asarray(uend), t=[t for t, _ in niter],
                 n=[v for _, v in niter], two_hop=getattr(C, 'two_hop_calls', 0),
                 bcast=getattr(C, 'bcast_two_hop_calls', 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['mssdc_P2_jac', 'mssdc_P2_gs', 'fixedK_3d_P2', 'forced2d_run_P2'])
def test_two_ranks_match_serial_golden(name):
    case = load_cases('runs.npz')[name]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), name, d), nprocs=2, join=True)
        r = [np.load(os.path.join(d, f'r{k}.npz')) for k in range(2)]
    times = np.concatenate([r[0]['t'], r[1]['t']])
    niter = np.concatenate([r[0]['n'], r[1]['n']])
    order = np.argsort(times)
    assert list(niter[order]) == list(case['niter'])                 # bit-exact iteration counts
    np.testing.assert_allclose(times[order], case['niter_t'], rtol=0, atol=1e-14)
    for k in range(2):                                                # every rank holds the broadcast end value
        assert rel_err(r[k]['uend'], case['uend']) < 1e-13


@pytest.mark.parametrize('relay', ['1', '0'])
@pytest.mark.parametrize('name,world', [('fixedK_2d_P4', 4), ('fixedK_2d_P3', 3), ('alltodone_2d_P4', 4),
                                        ('fixedK_2d_P4_tail', 4)])
def test_lockstep_runs_with_two_hop_exchange(name, world, relay):
    """3 and 4 ranks whose iteration counts are equal by construction (fixed K, all_to_done): the end values are
    forwarded over two hops by all ranks together (controller_dist.exchange_two_hop; relay='0': direct
    messages).  Same golden serial runs, same bits either way."""
    case = load_cases('runs_relay.npz')[name]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, d, 'runs_relay.npz', relay), nprocs=world, join=True)
        r = [np.load(os.path.join(d, f'r{k}.npz')) for k in range(world)]
    times = np.concatenate([x['t'] for x in r])
    niter = np.concatenate([x['n'] for x in r])
    order = np.argsort(times)
    assert list(niter[order]) == list(case['niter'])
    np.testing.assert_allclose(times[order], case['niter_t'], rtol=0, atol=1e-14)
    for k in range(world):
        assert rel_err(r[k]['uend'], case['uend']) < 1e-13
        assert (int(r[k]['two_hop']) > 0) == (relay == '1')
        assert (int(r[k]['bcast']) > 0) == (relay == '1')     # the end value of a block: scatter + all-gather
    assert all(np.array_equal(r[0]['uend'], x['uend']) for x in r[1:])


@pytest.mark.parametrize('name', ['fixedK_2d_P8', 'alltodone_2d_P8'])
def test_eight_ranks_lockstep(name):
    """eight processes (one node's worth of time ranks): two-hop hand-over with 8 pieces per message, the mesh broadcast
    with 7, a second block with four active ranks - against the reference's serial run with num_procs=8."""
    case = load_cases('runs_relay8.npz')[name]
    world = 8
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, d, 'runs_relay8.npz', '1'), nprocs=world, join=True)
        r = [np.load(os.path.join(d, f'r{k}.npz')) for k in range(world)]
    times = np.concatenate([x['t'] for x in r])
    niter = np.concatenate([x['n'] for x in r])
    order = np.argsort(times)
    assert list(niter[order]) == list(case['niter'])
    np.testing.assert_allclose(times[order], case['niter_t'], rtol=0, atol=1e-14)
    for k in range(world):
        assert rel_err(r[k]['uend'], case['uend']) < 1e-13
        assert int(r[k]['two_hop']) > 0 and int(r[k]['bcast']) > 0


@pytest.mark.parametrize('name,world', [('fixedK_2d_P3_nsweeps2', 3), ('fixedK_2d_P2_nsweeps2', 2)])
def test_two_sweeps_per_iteration(name, world):
    """level_params nsweeps=2 in a lock-step run: it_fine exchanges between the two sweeps of an iteration as well
    (controller_MPI.py:736-768); golden serial runs of the reference."""
    case = load_cases('runs_nsweeps2.npz')[name]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, d, 'runs_nsweeps2.npz', '1'), nprocs=world, join=True)
        r = [np.load(os.path.join(d, f'r{k}.npz')) for k in range(world)]
    times = np.concatenate([x['t'] for x in r])
    niter = np.concatenate([x['n'] for x in r])
    order = np.argsort(times)
    assert list(niter[order]) == list(case['niter'])
    for k in range(world):
        assert rel_err(r[k]['uend'], case['uend']) < 1e-13
