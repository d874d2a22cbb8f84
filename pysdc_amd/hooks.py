"""Hook interface (pySDC/core/hooks.py:22-230) and the always-on default hooks
(pySDC/implementations/hooks/default_hook.py:10-98 residual / niter statistics;
hooks/log_timings.py:10-342 timings, here with wall clock after a device synchronisation)."""
import time

from pysdc_amd.stats import Entry


class Hooks:
    def __init__(self):
        self.__stats = {}

    def add_to_stats(self, value, **kwargs):
        meta = dict(process=None, process_sweeper=None, time=None, level=None, iter=None, sweep=None, type=None,
                    num_restarts=None)
        meta.update(kwargs)
        self.__stats[Entry(**meta)] = value

    def increment_stats(self, value, initialize=None, **kwargs):
        meta = dict(process=None, process_sweeper=None, time=None, level=None, iter=None, sweep=None, type=None,
                    num_restarts=None)
        meta.update(kwargs)
        key = Entry(**meta)
        if key in self.__stats:
            self.__stats[key] += value
        else:
            self.__stats[key] = value if initialize is None else initialize

    def return_stats(self):
        for k, v in self.__stats.items():   # residuals that were on their way when they were logged are numbers by now
            if getattr(v, 'queued', False):
                self.__stats[k] = v.result()
        return self.__stats

    def reset_stats(self):
        self.__stats = {}

    def pre_setup(self, step, level_number): pass
    def pre_run(self, step, level_number): pass
    def pre_predict(self, step, level_number): pass
    def pre_step(self, step, level_number): pass
    def pre_iteration(self, step, level_number): pass
    def pre_sweep(self, step, level_number): pass
    def pre_comm(self, step, level_number): pass
    def post_comm(self, step, level_number, add_to_stats=False): pass
    def post_sweep(self, step, level_number): pass
    def post_predict(self, step, level_number): pass
    def post_iteration(self, step, level_number): pass
    def post_step(self, step, level_number): pass
    def post_setup(self, step, level_number): pass
    def post_run(self, step, level_number): pass


def _meta(step, L, **kw):
    d = dict(process=step.status.slot, process_sweeper=L.sweep.rank, time=L.time, level=L.level_index,
             iter=step.status.iter, sweep=L.status.sweep)
    d.update(kw)
    return d


def _residual_now_or_later(L):
    """the residual for the statistics: the number - or, when the device is still working on it, the ResidualFuture that
    return_stats() turns into the number (logging must not make the host wait for the device)"""
    peek = getattr(L.status, 'peek_residual', None)
    return peek() if peek is not None else L.status.residual


class DefaultHooks(Hooks):
    """default_hook.py:10-98."""

    def post_sweep(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(_residual_now_or_later(L), **_meta(step, L, type='residual_post_sweep'))

    def post_iteration(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(_residual_now_or_later(L), **_meta(step, L, type='residual_post_iteration'))

    def post_step(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(step.status.iter, **_meta(step, L, type='niter'))
        self.add_to_stats(_residual_now_or_later(L), **_meta(step, L, type='residual_post_step'))


class Timings(Hooks):
    """log_timings.py:10-342: timing_run / timing_step / timing_iteration / timing_sweep / timing_comm, keyed
    like the reference (value in seconds).  ``_get_event`` / ``_compute_time_elapsed`` select the clock."""

    prefix = ''

    def __init__(self):
        super().__init__()
        self._t = {}

    def _get_event(self):
        return time.perf_counter()

    def _compute_time_elapsed(self, event_after, event_before):
        return event_after - event_before

    def _start(self, key, step):
        self._t[(key, id(step))] = self._get_event()

    def _stop(self, key, step, level_number, add=True):
        t0 = self._t.pop((key, id(step)), None)
        if t0 is None or not add:
            return
        L = step.levels[level_number]
        self.add_to_stats(self._compute_time_elapsed(self._get_event(), t0),
                          **_meta(step, L, type=f'{self.prefix}timing_{key}'))

    def pre_run(self, step, level_number):
        self._start('run', step)

    def post_run(self, step, level_number):
        self._stop('run', step, level_number)

    def pre_step(self, step, level_number):
        self._start('step', step)

    def post_step(self, step, level_number):
        self._stop('step', step, level_number)

    def pre_iteration(self, step, level_number):
        self._start('iteration', step)

    def post_iteration(self, step, level_number):
        self._stop('iteration', step, level_number)

    def pre_sweep(self, step, level_number):
        self._start('sweep', step)

    def post_sweep(self, step, level_number):
        self._stop('sweep', step, level_number)

    def pre_comm(self, step, level_number):
        self._start('comm', step)

    def post_comm(self, step, level_number, add_to_stats=False):
        self._stop('comm', step, level_number, add=add_to_stats)


class CPUTimings(Timings):
    """always installed by the controllers (pySDC/core/controller.py:51)."""


class GPUTimings(Timings):
    """log_timings.py:328-342 with HIP events (through torch's stream events, recorded on the stream the
    engine launches on): device time between the hook calls, keys prefixed ``GPU_``."""

    prefix = 'GPU_'

    def _get_event(self):
        import torch

        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _compute_time_elapsed(self, event_after, event_before):
        event_after.synchronize()
        return event_before.elapsed_time(event_after) / 1e3


class LogWork(Hooks):
    """hooks/log_work.py:4-56: increments of all work counters of the problem between pre_step and post_step as
    ``work_<key>`` statistics (e.g. ``work_newton``, ``work_rhs``)."""

    def __init__(self):
        super().__init__()
        self._last = {}

    def pre_step(self, step, level_number):
        if level_number == 0:
            self._last[step.status.slot] = [
                {k: L.prob.work_counters[k].niter for k in L.prob.work_counters.keys()} for L in step.levels
            ]

    def post_step(self, step, level_number):
        L = step.levels[level_number]
        for key, before in self._last[step.status.slot][level_number].items():
            self.add_to_stats(L.prob.work_counters[key].niter - before, process=step.status.slot,
                              process_sweeper=L.sweep.rank, time=L.time + L.dt, level=L.level_index,
                              iter=step.status.iter, sweep=L.status.sweep, type=f'work_{key}')


class LogSolution(Hooks):
    """hooks/log_solution.py:9-38: the end value of every step as statistic ``u``.  ``L.uend`` is a view into the
    level's slab (it changes with the next step), so an owning device copy is stored."""

    def post_step(self, step, level_number):
        L = step.levels[level_number]
        L.sweep.compute_end_point()
        self.add_to_stats(L.prob.dtype_u(L.uend), process=step.status.slot, time=L.time + L.dt, level=L.level_index,
                          iter=step.status.iter, sweep=L.status.sweep, type='u')


class LogSolutionAfterIteration(Hooks):
    """hooks/log_solution.py:41-70: the same after every iteration."""

    def post_iteration(self, step, level_number):
        L = step.levels[level_number]
        L.sweep.compute_end_point()
        self.add_to_stats(L.prob.dtype_u(L.uend), process=step.status.slot, time=L.time + L.dt, level=L.level_index,
                          iter=step.status.iter, sweep=L.status.sweep, type='u')
