"""Round-4 entry points that batch what the reference does in a loop (include/sdcmi.h): sdc_eval_f_batch (eval_f of several
fields in one pass of the launches), sdc_transfer_apply_batch_acc (the last pass of a space transfer adds to its target; coarsening
in 3-D as one launch), sdc_residual_post_integrals (a residual that brings the quadrature sums along) - each against the calls it
replaces."""
import numpy as np
import pytest

from pysdc_amd import lib as L
from tests import _gpu as G

pytestmark = pytest.mark.gpu


def _coeffs(e, M, imex):
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    QI = np.zeros_like(c.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS['LU'](qGen=c.generator, tLeft=0).genCoeffs()
    QE = np.zeros_like(c.Qmat)
    if imex:
        QE[1:, 1:], QE[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    e.set_coeffs(c.Qmat, QI, QE if imex else None, c.nodes, c.weights)


@pytest.mark.parametrize('prob,nvars', [('heat_unforced', (32, 32, 32)), ('advdiff', (32, 32, 32)), ('heat_forced', (16, 16)),
                                        ('advection', (64,))])
def test_eval_f_batch_equals_field_by_field(prob, nvars):
    from pysdc_amd.hip_mesh import hip_mesh

    M = 3
    e = G.engine_for(prob, dict(nvars=nvars, nu=0.1, c=1.0, freq=2), M)
    imex = e.ncomp == 2
    _coeffs(e, M, imex)
    rng = np.random.default_rng(7)
    init = (nvars if len(nvars) > 1 else nvars[0], None, np.dtype('float64'))
    us = [hip_mesh(init) for _ in range(M)]
    for u in us:
        u[:] = rng.standard_normal(nvars)
    g = [0.3, -1.2, 2.5]
    one = [(hip_mesh(init), hip_mesh(init)) for _ in range(M)]
    many = [(hip_mesh(init), hip_mesh(init)) for _ in range(M)]
    for k in range(M):
        e.eval_f(us[k].ptr, g[k], one[k][0].ptr, one[k][1].ptr if imex else None)
    e.eval_f_many([u.ptr for u in us], [m[0].ptr for m in many], [m[1].ptr for m in many] if imex else None, g_ts=g)
    for k in range(M):
        assert np.array_equal(many[k][0].get(), one[k][0].get()), (prob, k)
        if imex:
            assert np.array_equal(many[k][1].get(), one[k][1].get()), (prob, k)
    e.close()


@pytest.mark.parametrize('ndim,nf,nc,order', [(3, 32, 16, 2), (3, 32, 16, 6), (2, 64, 32, 4), (1, 128, 64, 2)])
def test_transfer_accumulate_and_fused_coarsening(ndim, nf, nc, order):
    """prolongation that ADDS its result to the target equals prolongation + addition; restriction (in 3-D one launch for all
    axes) equals the Kronecker product of the 1-D rows applied by NumPy"""
    from pysdc_amd.hip_mesh import hip_mesh
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.transfer import mesh_to_mesh

    fine = heatNd_unforced(nvars=(nf,) * ndim, nu=0.1, freq=2)
    coarse = heatNd_unforced(nvars=(nc,) * ndim, nu=0.1, freq=2)
    T = mesh_to_mesh(fine, coarse, dict(iorder=order, rorder=2, periodic=True))
    rng = np.random.default_rng(11)
    fields = 3
    lib = L.load()
    shape_f, shape_c = (fields,) + (nf,) * ndim, (fields,) + (nc,) * ndim
    uf, uc = rng.standard_normal(shape_f), rng.standard_normal(shape_c)
    F = hip_mesh(((int(np.prod(shape_f)),), None, np.dtype('float64')))
    Cc = hip_mesh(((int(np.prod(shape_c)),), None, np.dtype('float64')))
    F[:] = uf.reshape(-1)
    Cc[:] = uc.reshape(-1)

    def apply(key, src, dst, acc):
        idx, w, width, (n_out, n_in) = T._tab[key]
        L.check(lib.sdc_transfer_apply_batch_acc(None, fields, ndim, n_out, n_in, width, idx.ptr, w.ptr, src.ptr, dst.ptr, int(acc)), None)

    # prolongation: out += P c  against  out + (P c)
    plain = hip_mesh(((F.size,), None, np.dtype('float64')))
    apply('P', Cc, plain, False)
    target = hip_mesh(F)
    apply('P', Cc, target, True)
    assert np.array_equal(target.get(), (F.get() + plain.get()))
    # restriction against the dense rows
    from pysdc_amd.transfer import interpolation_matrix_1d

    fine_grid = np.array([j * fine.dx for j in range(nf)])
    coarse_grid = np.array([j * coarse.dx for j in range(nc)])
    R1 = 0.5 * interpolation_matrix_1d(fine_grid, coarse_grid, k=2).T      # (what mesh_to_mesh builds its 'R' table from)
    want = uf
    for ax in range(1, ndim + 1):
        want = np.moveaxis(np.tensordot(R1, want, axes=([1], [ax])), 0, ax)
    got = hip_mesh(((Cc.size,), None, np.dtype('float64')))
    apply('R', F, got, False)
    assert np.max(np.abs(got.get().reshape(shape_c) - want)) <= 1e-14 * np.max(np.abs(want))


def test_residual_with_integrals_equals_the_two_calls():
    """node-by-node level (Allen-Cahn): the residual's pass over F also stores dt Q F - the same bits sdc_integrate writes, the
    same residual sdc_residual returns; a state whose residual comes from a cache writes nothing and says so"""
    from pysdc_amd.level import Step
    from pysdc_amd.problems import allencahn_imex
    from pysdc_amd.sweepers import imex_1st_order

    S = Step(dict(problem_class=allencahn_imex, problem_params=dict(nvars=(32, 32, 32), eps=0.08, radius=0.25),
                  sweeper_class=imex_1st_order, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU', QE='EE'),
                  level_params=dict(dt=1e-3), step_params=dict(maxiter=2)))
    Lv = S.levels[0]
    Lv.status.time = 0.0
    Lv.u[0] = Lv.prob.u_exact(0.0)
    Lv.sweep.predict()
    Lv.sweep.update_nodes()
    e = Lv.engine
    ref_res, ref_norms = e.residual(Lv.dt, 'full_abs')
    want = [x.get() for x in Lv.sweep.integrate()]
    me = Lv.sweep._integral_fields()
    fut = e.residual_post(Lv.dt, 'full_abs', integrals=[x.ptr for x in me])
    assert e.integrals_written
    assert fut.result() == ref_res and np.array_equal(fut.norms, ref_norms)
    for a, b in zip(me, want):
        assert np.array_equal(a.get(), b)
    # through the sweeper: compute_residual on a level whose restriction wants the sums, then integrate() hands them out once
    Lv.integrals_wanted = True
    Lv._res_cache = None
    Lv.sweep.compute_residual(stage='IT_FINE')
    cached = Lv._res_cache[2]
    assert cached is not None
    got = Lv.sweep.integrate()
    assert got is cached and Lv._res_cache[2] is None
    for a, b in zip(got, want):
        assert np.array_equal(a.get(), b)
    assert Lv.status.residual == ref_res
    Lv.sweep.update_nodes()                      # any change of the state drops what was cached
    assert Lv._res_cache is None
