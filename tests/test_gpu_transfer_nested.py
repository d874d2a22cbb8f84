"""GPU: the one-launch space transfers between nested periodic 3-D grids (sdc_transfer_apply_nested; round 6) against
vectors of the reference's mesh_to_mesh (tests/golden/transfer3d.npz, gen_golden.py: transfer3d_main) and against the
separable passes they replace, with the differences FAS forms around a transfer riding along."""
import ctypes as C

import numpy as np
import pytest

from tests._cases import load_cases, rel_err

pytestmark = pytest.mark.gpu


def _pair(m):
    from pysdc_amd.problems import heatNd_unforced
    from pysdc_amd.transfer import mesh_to_mesh

    pf = heatNd_unforced(nvars=tuple(m['nf']), nu=0.1, freq=2)
    pc = heatNd_unforced(nvars=tuple(m['nc']), nu=0.1, freq=2)
    return pf, pc, mesh_to_mesh(pf, pc, dict(iorder=m['iorder'], rorder=m['rorder'], periodic=True))


@pytest.mark.parametrize('name', list(load_cases('transfer3d.npz')))
def test_nested_transfer_vs_reference(name):
    c = load_cases('transfer3d.npz')[name]
    m = c['meta']
    pf, pc, T = _pair(m)
    # 64^3 <-> 32^3 take the fused launches, 48^3 <-> 24^3 (tiles do not fit) the separable passes: same entry point
    assert T._nested['R'] and T._nested['P']
    F, G = pf.u_init, pc.u_init
    F[:] = c['fine'].astype(float)
    G[:] = c['coarse'].astype(float)
    assert rel_err(T.restrict(F).get(), c['restricted']) < 1e-14
    assert rel_err(T.prolong(G).get(), c['prolonged']) < 1e-14


@pytest.mark.parametrize('n,io', [(64, 6), (64, 2), (128, 6), (128, 4), (96, 8), (48, 6)])
def test_nested_transfer_vs_separable_passes(n, io):
    """random fields, three at once, with u_G - uold_G on the way in / - Q_G f_G on the way out / += on the fine side: the
    fused launches and the passes + axpby launches they replace agree to rounding"""
    from pysdc_amd import lib as Lb
    from pysdc_amd.hip_mesh import hip_mesh

    pf, pc, T = _pair(dict(nf=(n,) * 3, nc=(n // 2,) * 3, iorder=io, rorder=2))
    lib = Lb.load()
    rng = np.random.default_rng(n + io)
    nf3, nc3, K = n**3, (n // 2) ** 3, 3

    def dev(a):
        x = hip_mesh(((a.size,), None, np.dtype('float64')), val=None)
        x[:] = a.reshape(-1)
        return x

    fine = rng.standard_normal((K, nf3))
    coarse, cold, qc = (rng.standard_normal((K, nc3)) for _ in range(3))
    # restriction with a difference on the way out
    idx, w, width, (n_out, n_in) = T._tab['R']
    dF, dQ = dev(fine), dev(qc)
    got, want = dev(np.zeros((K, nc3))), dev(np.zeros((K, nc3)))
    Lb.check(lib.sdc_transfer_apply_nested(None, K, 3, n_out, n_in, width, idx.ptr, w.ptr, dF.ptr, None, got.ptr, dQ.ptr, 0), None)
    Lb.check(lib.sdc_transfer_apply_batch_acc(None, K, 3, n_out, n_in, width, idx.ptr, w.ptr, dF.ptr, want.ptr, 0), None)
    ref = want.get().reshape(K, nc3) - qc
    assert rel_err(got.get().reshape(K, nc3), ref) < 1e-14
    # prolongation of a difference, added to what is there
    idx, w, width, (n_out, n_in) = T._tab['P']
    dC, dO = dev(coarse), dev(cold)
    got, want, diff = dev(fine.copy()), dev(fine.copy()), dev(coarse - cold)
    Lb.check(lib.sdc_transfer_apply_nested(None, K, 3, n_out, n_in, width, idx.ptr, w.ptr, dC.ptr, dO.ptr, got.ptr, None, 1), None)
    Lb.check(lib.sdc_transfer_apply_batch_acc(None, K, 3, n_out, n_in, width, idx.ptr, w.ptr, diff.ptr, want.ptr, 1), None)
    assert rel_err(got.get(), want.get()) < 1e-14
    # plain prolongation (stored, not added)
    Lb.check(lib.sdc_transfer_apply_nested(None, K, 3, n_out, n_in, width, idx.ptr, w.ptr, dC.ptr, None, got.ptr, None, 0), None)
    Lb.check(lib.sdc_transfer_apply_batch_acc(None, K, 3, n_out, n_in, width, idx.ptr, w.ptr, dC.ptr, want.ptr, 0), None)
    assert rel_err(got.get(), want.get()) < 1e-14
