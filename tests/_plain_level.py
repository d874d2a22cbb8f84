"""TEST-ONLY: a Level / Step pair with the semantics of the reference's containers (pySDC/core/level.py:42-131,
core/step.py:47-271) and none of pysdc_amd.level's device plumbing: plain Python lists for u / f / tau that
``reset_level`` REPLACES, a frozen attribute set, ``init_step`` assigning a copy-constructed datatype object.  On the GPU
box (where the reference cannot be imported) it stands in for the reference's Level so that the sweeper's
ForeignLevelState path runs on the real device (tests/test_gpu_plugin.py::test_foreign_level_*)."""
from pysdc_amd.level import LevelParams, LevelStatus, StepParams, StepStatus


class _Frozen:
    _frozen = False

    def __setattr__(self, key, value):
        if self._frozen and not hasattr(self, key):
            raise TypeError(f'{type(self).__name__} is frozen: no new attribute {key!r}')
        object.__setattr__(self, key, value)


class PlainLevel(_Frozen):
    def __init__(self, problem_class, problem_params, sweeper_class, sweeper_params, level_params, level_index):
        self.params = LevelParams(level_params)
        self.status = LevelStatus()
        self._sweep = sweeper_class(sweeper_params, self)
        self._prob = problem_class(**problem_params)
        self.level_index = level_index
        M = self._sweep.coll.num_nodes
        self.uend = None
        self.u = [None] * (M + 1)
        self.uold = [None] * (M + 1)
        self.u_avg = [None] * M
        self.residual = [None] * M
        self.increment = [None] * M
        self.f = [None] * (M + 1)
        self.fold = [None] * (M + 1)
        self.tau = [None] * M
        self.tag = None
        self._frozen = True

    def reset_level(self, reset_status=True):
        if reset_status:
            self.status = LevelStatus()
        M = self._sweep.coll.num_nodes
        self.uend = None
        self.u = [None] * (M + 1)
        self.uold = [None] * (M + 1)
        self.f = [None] * (M + 1)
        self.fold = [None] * (M + 1)
        self.tau = [None] * M

    sweep = property(lambda self: self._sweep)
    prob = property(lambda self: self._prob)
    time = property(lambda self: self.status.time)
    dt = property(lambda self: self.params.dt)


from pysdc_amd.level import Step as _ProductStep


class PlainStep(_ProductStep):
    """the product's Step logic (level hierarchy from list-valued parameters, transfer operators between neighbours) over
    PlainLevel containers; init_step assigns a copy-constructed datatype object like core/step.py:256-271"""

    level_class = PlainLevel

    def init_step(self, u0):
        P = self.levels[0].prob
        self.levels[0].u[0] = P.dtype_u(u0)
