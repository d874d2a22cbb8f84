"""Constructor-argument checks of the finite-difference problems: same error type and message as the reference
(pySDC/implementations/problem_classes/generic_ND_FD.py:99-132) for the same arguments - the expected texts below were
recorded from the reference in the build container."""
import pytest

from pysdc_amd.errors import ProblemError
from pysdc_amd.problems import _grid_spec

BAD = [
    (dict(nvars=[64, 64], freq=2, bc='periodic'), 'nvars should be either tuple or int'),
    (dict(nvars=64, freq=2.0, bc='periodic'), 'freq should be either tuple or int'),
    (dict(nvars=(8, 8, 8, 8), freq=2, bc='periodic'), 'can work with up to three dimensions, got 4'),
    (dict(nvars=(8, 8), freq=(2, 2, 2), bc='periodic'), 'len(freq)=3, different to ndim=2'),
    (dict(nvars=(8, 8), freq=(2, 3), bc='periodic'), 'need even number of frequencies due to periodic BCs'),
    (dict(nvars=(8, 8), freq=(-1, 2), bc='periodic'), 'need even number of frequencies due to periodic BCs'),
    (dict(nvars=63, freq=2, bc='periodic'), 'the setup requires nvars = 2^p per dimension'),
    (dict(nvars=64, freq=1, bc='dirichlet-zero'), 'setup requires nvars = 2^p - 1'),
    (dict(nvars=(8, 16), freq=2, bc='periodic'), 'need a square domain, got (8, 16)'),
]


@pytest.mark.parametrize('kw,msg', BAD)
def test_rejected_like_the_reference(kw, msg):
    with pytest.raises(ProblemError) as e:
        _grid_spec(**kw)
    assert msg in str(e.value)


def test_accepted_and_normalised():
    assert _grid_spec(64, 2, 'periodic') == ((64,), (2,), 'periodic')
    assert _grid_spec((16, 16, 16), 4, 'periodic') == ((16, 16, 16), (4, 4, 4), 'periodic')
    assert _grid_spec(63, 3, 'dirichlet-zero') == ((63,), (3,), 'dirichlet-zero')     # odd frequencies are fine there
    assert _grid_spec(64, -1, 'dirichlet-zero') == ((64,), (-1,), 'periodic')          # 1-D Gaussian start value


def test_level_status_holds_a_put_off_residual():
    """pysdc_amd.level.LevelStatus.residual: a callable handed to it is evaluated once, when the attribute is read;
    readers (comparisons, get, copies) only ever see the number"""
    import copy
    import pickle

    from pysdc_amd.level import LevelStatus

    calls = []

    def thunk():
        calls.append(1)
        return 0.25

    st = LevelStatus()
    assert st.residual is None and not st.residual_is_deferred()
    st.residual = thunk
    assert st.residual_is_deferred() and not calls
    assert st.residual <= 0.3 and st.residual == 0.25 and st.get('residual') == 0.25
    assert calls == [1] and not st.residual_is_deferred()
    st.residual = thunk
    st.drop_deferred_residual()               # the state it belonged to is gone: nobody asked
    assert st.residual is None and calls == [1]
    st.residual = thunk
    clone = pickle.loads(pickle.dumps(st))    # copies carry the number
    assert clone.residual == 0.25 and calls == [1, 1]
    st.residual = 1.5
    assert copy.deepcopy(st).residual == 1.5 and st.get('unlocked') is False


def test_bounded_grids_of_any_mix_are_accepted_like_the_reference():
    """bc as one string for both ends or a pair (generic_ND_FD.py:50-54); an end is Dirichlet / Neumann if its string
    contains the word (helpers/problem_helper.py:157); point counts are only checked for the exact string 'dirichlet-zero'
    (generic_ND_FD.py:130); bcParams changes nothing, as in the reference (generic_ND_FD.py:140-148)."""
    import numpy as np

    from pysdc_amd.problems import heatNd_unforced, GenericNDimFinDiff

    P = heatNd_unforced(nvars=40, nu=0.1, freq=1, order=2, bc=('dirichlet', 'neumann'))
    assert P.bc == ('dirichlet', 'neumann') and P.banded and P.fused and P.engine_nvars == (40,)
    assert abs(P.dx - 1 / 41) < 1e-15
    assert heatNd_unforced(nvars=48, bc='dirichlet').banded                       # (an even grid: no rule for this string)
    assert heatNd_unforced(nvars=63, bc='dirichlet-zero').view_offset == 1        # order 2: the odd extension stays
    assert heatNd_unforced(nvars=(16, 16), bc='neumann', freq=(1, 2)).banded
    a = GenericNDimFinDiff(nvars=17, coeff=0.5, derivative=2, order=4, bc='neumann')
    b = GenericNDimFinDiff(nvars=17, coeff=0.5, derivative=2, order=4, bc='neumann', bcParams={'val': 3.0, 'reduce': True})
    assert np.array_equal(a._rows[0], b._rows[0]) and np.array_equal(a._rows[1], b._rows[1])
    with pytest.raises(NotImplementedError):
        heatNd_unforced(nvars=40, bc='robin')
    with pytest.raises(NotImplementedError):       # get_1d_grid looks for the exact words in a pair (problem_helper.py:263)
        heatNd_unforced(nvars=40, bc=('dirichlet-zero', 'neumann-zero'))
