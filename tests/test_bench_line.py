"""The ONE line bench.py prints is what the driver parses, out of the last 8 KB of standard output it keeps: the line must
stay far below that, be strict JSON (no NaN / Infinity tokens) and carry the contract's keys plus `roofline` and
`cpu_baseline` - with every sub-record present, with eight ranks reporting, and when numbers are not finite."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (standard library only at import time: no GPU, no torch)

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config')


def _record(title=None, world=1, steps=20, bad=False):
    x = float('nan') if bad else 15.2345678901234
    kern = {f'kernel_number_{i}[5]': {'ms_per_launch': x, 'launches': 80, 'gbs': 5000.123456789} for i in range(40)}
    roof = {'kernel': 'spec_z_res_v0[5]', 'bound': 'hbm', 'achieved': 5655.123456789 if not bad else float('inf'), 'peak': 8000.0,
            'unit': 'GB/s', 'frac': 0.7068904320987, 'traffic': 86105972736.0, 'traffic_source': 'p' * 150,
            'algorithmic_bytes_per_launch': 86067118080.0,
            'ms_per_launch': x, 'stream_reference_gbs': {'fill': 5600.0, 'amax': 5500.0, 'copy': 4800.0}}
    rs = {'bound': 'hbm', 'peak': 8000.0, 'unit': 'GB/s', 'ms_per_sweep': 37.4123456789, 'bytes_moved_per_sweep': 180741996544.0,
          'achieved': 4831.123456789, 'frac': 0.60389, 'floor_bytes_per_sweep': 137438953472.0, 'achieved_on_floor': 3673.6,
          'frac_on_floor': 0.4592, 'launches_ms_per_sweep': {f'k{i}': 1.0 for i in range(8)}}
    out = {'metric': 'time-steps/s (HeatND 3-D FD, M=5, implicit SDC sweeps)', 'value': 6.123456789012345, 'unit': 'time-steps/s',
           'n_gpus': world, 'steps': steps, 'warmup': 5, 'ms_per_step': 163.30612345678, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': 'w' * 900, 'time_parallel': 't' * 300},
           'sdc_iters_per_s': 24.4938271, 'niter': [4] * steps * world, 'work_counters': {'rhs': 144}, 'sweep_kernels_ms': 37.4,
           'sweep_floor_gbs': 3673.6, 'roofline': roof, 'roofline_sweep': rs, 'kernels': kern, 'finite': True,
           'device_bytes_per_gpu': 231928233984, 'params': {'M': 5, 'dt': 2.5e-4, 'n': 1024}, 'restol': -1.0}
    if title is not None:
        out['title'] = title
    if world > 1:
        out['per_rank'] = [{'rank': r, 'seconds': 12.3456789012, 'niter': [4] * steps, 'ms_per_iteration': 77.123456789,
                            'comm_ms_per_iteration': 33.123456789, 'kernel_ms_per_iteration': 71.123456789, 'two_hop_exchanges': 80,
                            'mesh_broadcasts': 20, 'wire': 'rccl', 'message_bytes': 8 * 1024**3} for r in range(world)]
    return out


def _cpu():
    return {'value': 1.2345678e-3, 'unit': 'time-steps/s', 'cores': 32, 'kind': 'port', 'sample': 's' * 200,
            'one_core_64_dof_scaled': 8.56e-5,
            'one_core': {'value': 4.9e-5, 'n': 256, 'seconds_per_sweep': 78.5455453, 'cg_iterations': 406},
            'same_algorithm': {'value': 2.3456789e-2, 'unit': 'time-steps/s', 'cores': 32, 'n': 128, 'seconds_per_sweep': 4.123456,
                               'note': 'n' * 180, 'one_core': {'value': 1.1e-3, 'n': 256, 'seconds_per_sweep': 35.123456}}}


def _strict(line):
    assert 'NaN' not in line and 'Infinity' not in line
    return json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))


def test_headline_line_with_every_sub_record_fits_and_parses():
    subs = [_record(title=t) for t, _ in bench.SUB_PLAN] + [{'title': 'x' * 100, 'error': 'e' * 1000}]
    for r in subs:
        if r['title'].startswith('cfg5'):
            r['ms_per_step_with_events'] = 18.123456789
    subs[2]['niter'] = [43, 32]
    subs[2]['restol'] = 1e-10
    line = json.dumps(bench.compact_line(_record(), subs, _cpu(), 'gpurun_out/bench_details.json', _record(steps=20)),
                      allow_nan=False)
    assert len(line) < 6000, len(line)
    rec = _strict(line)
    # the headline configuration re-timed at the end of the process
    assert rec['value_sustained'] and rec['sustained']['steps'] == 20 and 0 < rec['sustained']['roofline']['frac'] < 1
    assert rec['sustained']['sweep_frac'] and rec['sustained']['ms_per_step']
    for k in CONTRACT:
        assert k in rec, k
    assert set(rec['config']) == {'workload', 'time_parallel'} and 'model' not in rec['config']
    assert rec['roofline']['bound'] == 'hbm' and 0 < rec['roofline']['frac'] < 1
    assert rec['roofline']['traffic'] and rec['roofline']['algorithmic_bytes_per_launch'] and rec['roofline']['ms_per_launch']
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(rec['cpu_baseline']) and len(rec['cpu_baseline']['sample']) <= 200
    assert rec['cpu_baseline']['same_algorithm']['one_core']['n'] == 256 and rec['roofline']['traffic_source']
    sub5 = [x for x in rec['sub'] if x['title'].startswith('cfg5')][0]
    assert 'ms_per_step_with_events' in sub5     # (timed without HIP events, the kernel table from a second run with them)
    assert {'frac', 'frac_on_floor'} <= set(rec['roofline_sweep'])
    assert len(rec['sub']) == len(bench.SUB_PLAN) + 1 and all(len(s['title']) <= 40 for s in rec['sub'])
    assert rec['niter'] == 4 and rec['sub'][2]['niter'] == [43, 32]
    assert rec['value_eager_fields'] and rec['value_lazy_predictor_residual']
    assert all(len(t) <= 40 for t, _ in bench.SUB_PLAN)


def test_a_failed_sustained_run_does_not_take_the_line_down():
    rec = _strict(json.dumps(bench.compact_line(_record(), None, _cpu(), None, {'error': 'MemoryError()' * 40}), allow_nan=False))
    assert rec['value_sustained'] is None and len(rec['sustained']['error']) <= 120 and rec['value']


def test_eight_rank_line_fits_and_parses():
    line = json.dumps(bench.compact_line(_record(world=8)), allow_nan=False)
    assert len(line) < 6000, len(line)
    rec = _strict(line)
    assert rec['n_gpus'] == 8 and len(rec['per_rank']['seconds']) == 8 and rec['per_rank']['wire'] == 'rccl'


def test_numbers_that_are_not_finite_become_null():
    line = json.dumps(bench.compact_line(_record(bad=True), [_record(title='t', bad=True)], _cpu()), allow_nan=False)
    rec = _strict(line)
    assert rec['roofline']['achieved'] is None and rec['roofline']['ms_per_launch'] is None
    assert math.isfinite(rec['value'])
