"""GPU: the datatype as the ndarray the reference's `mesh` is (datatype_classes/mesh.py:12-60) - integer and slice indexing
through the box gather / scatter of the C-ABI (include/sdcmi.h: sdc_vec_box), index arrays and masks, reductions, reshape - on
owning fields and on views into a level's slabs, against NumPy on the host copy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS_3D = [(3,), (-1,), (slice(1, -1),), (Ellipsis, 2), (slice(None), 4), (2, slice(None, None, 2), slice(5, 1, -1)),
           (slice(None, None, -1),), (Ellipsis, slice(0, 0)), (1, 2, 3), (-2, -3, -1), (slice(2, 9, 3), Ellipsis, slice(1, None, 4)),
           (slice(None), slice(None), slice(None))]


def _mesh(shape, seed=0):
    from pysdc_amd.hip_mesh import hip_mesh

    h = np.random.default_rng(seed).standard_normal(shape)
    m = hip_mesh((shape, None, np.dtype('float64')))
    m[:] = h
    return m, h


@pytest.mark.parametrize('shape', [(40,), (12, 10), (9, 8, 10)])
def test_basic_indexing_equals_numpy(shape):
    m, h = _mesh(shape)
    keys = KEYS_3D if len(shape) == 3 else [k for k in KEYS_3D if len([x for x in k if x is not Ellipsis]) <= len(shape)]
    for key in keys:
        key = key[0] if len(key) == 1 else key
        want = h[key]
        got = m[key]
        if np.ndim(want) == 0:
            assert isinstance(got, float) and got == want, key
        else:
            assert got.shape == want.shape and np.array_equal(got.get(), want), key
            # one contiguous piece of memory (leading integers / unit-step slices, whole trailing axes): a view like ndarray's;
            # anything strided: a copy that refuses to be written through
            base = np.shares_memory(want, h) and want.flags['C_CONTIGUOUS'] and want.size > 0
            assert (got.ptr == m.ptr + (want.__array_interface__['data'][0] - h.__array_interface__['data'][0])) == bool(base), key
            if not base and want.size:
                with pytest.raises(Exception):
                    got[:] = 0.0
    assert len(m) == shape[0] and [np.array_equal(np.asarray(r), h[i]) for i, r in enumerate(m)] == [True] * shape[0]
    with pytest.raises(IndexError):
        m[(shape[0],) + (0,) * (len(shape) - 1)]
    with pytest.raises(IndexError):
        m[(0,) * (len(shape) + 1)]


@pytest.mark.parametrize('shape', [(40,), (12, 10), (9, 8, 10)])
def test_assignment_through_an_index_equals_numpy(shape):
    from pysdc_amd.hip_mesh import hip_mesh

    m, h = _mesh(shape, 1)
    rng = np.random.default_rng(2)
    for key in ((slice(1, -1),), (Ellipsis, 2), (-1,), (slice(None, None, 2),), (0,) * len(shape)):
        key = key[0] if len(key) == 1 else key
        # a scalar, a host array, a device field
        h[key] = 0.25
        m[key] = 0.25
        assert np.array_equal(m.get(), h), key
        val = rng.standard_normal(np.shape(h[key]))
        h[key] = val
        m[key] = val
        assert np.array_equal(m.get(), h), key
        if np.ndim(val):
            dev = hip_mesh((val.shape, None, np.dtype('float64')))
            dev[:] = 2.0 * val
            h[key] = 2.0 * val
            m[key] = dev
            assert np.array_equal(m.get(), h), key


def test_contiguous_indices_are_views_like_ndarray():
    """`v = u[1:-1]; v[:] = x` and `u[0][...] = x` write the field itself (datatype_classes/mesh.py: the reference's mesh IS an
    ndarray), also through a view into a level's slab - where the write reaches the engine"""
    m, h = _mesh((9, 8, 10), 4)
    v = m[1:-1]
    v[:] = 2.5
    h[1:-1] = 2.5
    assert np.array_equal(m.get(), h)
    m[0][...] = -1.0
    h[0][...] = -1.0
    m[3, 2][4:7] = 9.0
    h[3, 2][4:7] = 9.0
    row = m[5, 1:3]
    row += row
    h[5, 1:3] += h[5, 1:3]
    assert np.array_equal(m.get(), h)
    assert np.array_equal(np.asarray(m), h) and np.array(m).shape == h.shape    # one bulk copy (__array__), not one call per element


def test_index_arrays_masks_reductions_and_reshape():
    m, h = _mesh((12, 10), 3)
    idx = np.array([0, 3, 3, 11])
    assert np.array_equal(m[idx].get(), h[idx]) and np.array_equal(m[idx, 2].get(), h[idx, 2])
    mask = h > 0.5
    assert np.array_equal(m[mask].get(), h[mask])
    m[mask] = -1.0
    h[mask] = -1.0
    assert np.array_equal(m.get(), h)
    assert m.max() == h.max() and m.min() == h.min() and abs(m.sum() - h.sum()) < 1e-12 and abs(m.mean() - h.mean()) < 1e-14
    r = m.reshape(10, -1)
    assert r.shape == (10, 12) and np.array_equal(r.get(), h.reshape(10, 12))
    r[0] = 5.0                      # a reshape is a view of the same memory
    assert np.array_equal(m.get().reshape(-1)[:12], np.full(12, 5.0))
    assert np.array_equal(np.asarray(m.flatten()), m.get().reshape(-1)) and m.ravel().shape == (120,)
    assert np.linalg.norm(m) == np.linalg.norm(m.get())


def test_a_hook_that_indexes_level_fields_during_a_run():
    """views into the level's slabs answer an index like any other field - also while the node values they stand for live
    in Fourier space only (the access brings them back) - and writing through an index tells the engine (the next sweep
    starts from the changed value)"""
    from pysdc_amd.controller import controller_nonMPI
    from pysdc_amd.hooks import Hooks
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.sweepers import generic_implicit

    seen = []

    class Probe(Hooks):
        def post_iteration(self, step, level_number):
            L = step.levels[level_number]
            L.sweep.compute_end_point()
            u, last = L.uend, L.u[-1]
            seen.append((u[5, 6, 7], float(np.max(np.abs(np.asarray(u[2]) - np.asarray(last[2])))), last[..., 0].shape))

    n = 64
    desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=2),
                sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='IE'),
                level_params=dict(dt=1e-2, restol=-1), step_params=dict(maxiter=3))
    C = controller_nonMPI(1, dict(logger_level=40, hook_class=[Probe]), desc)
    P = C.MS[0].levels[0].prob
    uend, _ = C.run(P.u_exact(0.0), 0.0, 2e-2)
    assert len(seen) == 6 and all(d == 0.0 and s == (n, n) for _, d, s in seen)   # right end point is a node: uend IS u[-1]
    assert seen[-1][0] == uend[5, 6, 7] == uend.get()[5, 6, 7]
    # writing through an index of u[0] reaches the engine
    L = C.MS[0].levels[0]
    C2 = controller_nonMPI(1, dict(logger_level=40), desc)
    a = P.u_exact(0.0)
    a[0, 0, :4] = 3.0
    ref, _ = C2.run(a, 0.0, 1e-2)
    L2 = C2.MS[0].levels[0]
    assert L2.u[0][0, 0, 2] == 3.0 and abs(ref[0, 0, 2] - uend[0, 0, 2]) > 1e-3
