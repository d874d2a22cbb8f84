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


class DefaultHooks(Hooks):
    """default_hook.py:10-98."""

    def post_sweep(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(L.status.residual, **_meta(step, L, type='residual_post_sweep'))

    def post_iteration(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(L.status.residual, **_meta(step, L, type='residual_post_iteration'))

    def post_step(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(step.status.iter, **_meta(step, L, type='niter'))
        self.add_to_stats(L.status.residual, **_meta(step, L, type='residual_post_step'))
        self.add_to_stats(step.status.get('restart'), **_meta(step, L, iter=0, sweep=0, type='_recomputed')) \
            if False else None


class Timings(Hooks):
    """log_timings.py: timing_run / timing_step / timing_iteration / timing_sweep / timing_comm."""

    def __init__(self):
        super().__init__()
        self._t = {}

    @staticmethod
    def _now(step=None):
        return time.perf_counter()

    def pre_run(self, step, level_number):
        self._t[('run', id(step))] = self._now()

    def post_run(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(self._now() - self._t.pop(('run', id(step))), **_meta(step, L, type='timing_run'))

    def pre_step(self, step, level_number):
        self._t[('step', id(step))] = self._now()

    def post_step(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(self._now() - self._t.pop(('step', id(step))), **_meta(step, L, type='timing_step'))

    def pre_sweep(self, step, level_number):
        self._t[('sweep', id(step))] = self._now()

    def post_sweep(self, step, level_number):
        L = step.levels[level_number]
        self.add_to_stats(self._now() - self._t.pop(('sweep', id(step))), **_meta(step, L, type='timing_sweep'))
