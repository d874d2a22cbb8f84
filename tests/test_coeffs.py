"""Host-side coefficient module (pysdc_amd/coeffs.py) against the property tests the reference applies at
the qmat boundary: tests/test_collocation.py:15-120, tests/test_sweepers/test_preconditioners.py:15-207,
tests/test_Q_transfer.py, and closed-form Radau-IIA tableaux / the literal node values in
tutorial/step_7/D_pySDC_with_PyTorch.py:46.  Literal tableaux from the literature pin the values independently of this repository."""
import numpy as np
import pytest

from pysdc_amd.coeffs import (CollBase, Collocation, QDELTA_GENERATORS, LagrangeApproximation, NODE_TYPES,
                              QUAD_TYPES)
from tests._cases import load_cases

GRID = [(nt, qt, M) for nt in NODE_TYPES for qt in QUAD_TYPES for M in (2, 3, 4, 5)]


def qd(coll, name, k=None):
    return QDELTA_GENERATORS[name](qGen=coll.generator, tLeft=coll.tleft).genCoeffs(k=k)


def test_radau_literals():
    c = CollBase(3, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    # literal printed by the reference's own stack (qmat) - itself ~7e-16 off the closed form
    np.testing.assert_allclose(c.nodes, [0.15505102572168285, 0.6449489742783183, 1.0], rtol=0, atol=1e-15)
    s6 = np.sqrt(6)
    np.testing.assert_allclose(c.nodes, [(4 - s6) / 10, (4 + s6) / 10, 1.0], rtol=0, atol=2e-16)
    A = np.array([[(88 - 7 * s6) / 360, (296 - 169 * s6) / 1800, (-2 + 3 * s6) / 225],
                  [(296 + 169 * s6) / 1800, (88 + 7 * s6) / 360, (-2 - 3 * s6) / 225],
                  [(16 - s6) / 36, (16 + s6) / 36, 1 / 9]])
    np.testing.assert_allclose(c.Qmat[1:, 1:], A, rtol=0, atol=5e-16)
    np.testing.assert_allclose(c.weights, A[-1], rtol=0, atol=5e-16)
    assert c.order == 5 and c.right_is_node and not c.left_is_node
    assert np.all(c.Qmat[0] == 0) and np.all(c.Qmat[:, 0] == 0)


@pytest.mark.parametrize('nt,qt', [(nt, qt) for nt in NODE_TYPES for qt in QUAD_TYPES])
def test_quadrature_exactness(nt, qt):
    """tests/test_collocation.py: weights / Q integrate polynomials exactly; S is the row difference of Q."""
    for M in range(2, 13):
        if qt == 'LOBATTO' and M < 2:
            continue
        t0, t1 = 0.25, 1.5
        c = CollBase(M, t0, t1, nt, qt)
        deg = M - 1 if nt != 'LEGENDRE' else c.order - 1
        for p in range(deg + 1):
            exact = (t1 ** (p + 1) - t0 ** (p + 1)) / (p + 1)
            assert abs(c.weights @ c.nodes**p - exact) < 1e-13 * max(1.0, abs(exact)), (M, p)
        for p in range(M):
            exact = (c.nodes ** (p + 1) - t0 ** (p + 1)) / (p + 1)
            assert np.max(np.abs(c.Qmat[1:, 1:] @ c.nodes**p - exact)) < 1e-13 * max(1.0, np.max(np.abs(exact))), (M, p)
        assert np.max(np.abs(np.cumsum(c.Smat[1:, 1:], axis=0) - c.Qmat[1:, 1:])) < 1e-15
        assert np.allclose(np.cumsum(c.delta_m) + t0, c.nodes, atol=1e-15)
        assert np.all(np.diff(c.nodes) > 0)


@pytest.mark.parametrize('nt,qt,M', GRID)
def test_min_sr(nt, qt, M):
    c = CollBase(M, 0, 1, nt, qt)
    Q = c.Qmat[1:, 1:]
    QD = qd(c, 'MIN-SR-NS')
    assert np.all(np.diag(np.diag(QD)) == QD)
    assert np.linalg.norm(np.linalg.matrix_power(Q - QD, M), ord=np.inf) < 1e-10
    QD = qd(c, 'MIN-SR-S')
    assert np.all(np.diag(np.diag(QD)) == QD)
    if qt in ('LOBATTO', 'RADAU-LEFT'):
        D, Qr = np.diag(1 / np.diag(QD[1:, 1:])), Q[1:, 1:]
    else:
        D, Qr = np.diag(1 / np.diag(QD)), Q
    K = np.eye(Qr.shape[0]) - D @ Qr
    assert np.linalg.norm(np.linalg.matrix_power(K, M), ord=np.inf) < 1e-10


@pytest.mark.parametrize('nt,qt,M', GRID)
def test_min_sr_flex(nt, qt, M):
    c = CollBase(M, 0, 1, nt, qt)
    start = 1 if c.nodes[0] == 0 else 0
    Q = c.Qmat[1 + start:, 1 + start:]
    I = np.eye(M - start)
    K = np.eye(M - start)
    for k in range(1, M + 1):
        QD = qd(c, 'MIN-SR-FLEX', k=k)[start:, start:]
        assert np.all(np.diag(np.diag(QD)) == QD)
        K = (I - np.linalg.inv(QD) @ Q) @ K
    assert np.linalg.norm(K, ord=np.inf) < 1e-10
    assert QDELTA_GENERATORS['MIN-SR-FLEX'](qGen=c.generator, tLeft=0).isKDependent()
    assert not QDELTA_GENERATORS['LU'](qGen=c.generator, tLeft=0).isKDependent()


@pytest.mark.parametrize('nt,qt,M', GRID)
def test_lu_ie_structure(nt, qt, M):
    c = CollBase(M, 0, 1, nt, qt)
    Q = c.Qmat[1:, 1:]
    if not (M > 3 and nt == 'EQUID' and qt in ('GAUSS', 'RADAU-RIGHT')):
        QD = qd(c, 'LU')
        assert np.all(np.triu(QD, 1) == 0)
        Qr, QDr = (Q[1:, 1:], QD[1:, 1:]) if qt in ('LOBATTO', 'RADAU-LEFT') else (Q, QD)
        K = np.eye(Qr.shape[0]) - np.linalg.solve(QDr, Qr)
        assert np.linalg.norm(np.linalg.matrix_power(K, M), ord=np.inf) < 1e-14 * 50
    IE = qd(c, 'IE')
    for i in range(M):
        assert np.all(IE[i, : i + 1] == IE[-1, : i + 1])
    assert np.allclose(np.cumsum(IE[-1]), c.nodes, atol=1e-15)
    assert np.all(np.diag(qd(c, 'IEpar')) == c.nodes) and np.all(qd(c, 'PIC') == 0)
    assert np.all(np.diag(qd(c, 'Qpar')) == np.diag(Q))
    EE, dtau = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    assert np.all(np.triu(EE) == 0) and np.all(dtau == c.nodes[0])
    for i in range(1, M):
        assert np.allclose(EE[i, :i], c.delta_m[1: i + 1])


def test_interpolation_matrix_order():
    """tests/test_Q_transfer.py: interpolation between node sets is exact for polynomials."""
    for Mc, Mf in ((2, 3), (3, 5), (3, 3)):
        fine = CollBase(Mf, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
        coarse = CollBase(Mc, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
        P = LagrangeApproximation(coarse.nodes).getInterpolationMatrix(fine.nodes)
        R = LagrangeApproximation(fine.nodes).getInterpolationMatrix(coarse.nodes)
        for p in range(Mc):
            assert np.max(np.abs(P @ coarse.nodes**p - fine.nodes**p)) < 5e-15
        for p in range(Mf):
            assert np.max(np.abs(R @ fine.nodes**p - coarse.nodes**p)) < 5e-15


def test_literal_tableaux():
    """closed-form collocation tableaux from the literature (Hairer & Wanner, Solving ODEs II, IV.5: Gauss = Kuntzmann-
    Butcher, Radau IIA, Lobatto IIIA) - values no part of this repository computed."""
    s3, s15 = np.sqrt(3), np.sqrt(15)
    tol = dict(rtol=0, atol=5e-16)
    c = CollBase(2, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    np.testing.assert_allclose(c.nodes, [1 / 3, 1], **tol)
    np.testing.assert_allclose(c.Qmat[1:, 1:], [[5 / 12, -1 / 12], [3 / 4, 1 / 4]], **tol)
    np.testing.assert_allclose(qd(c, 'LU'), [[5 / 12, 0], [3 / 4, 2 / 5]], **tol)  # U^T of Q^T = L U
    np.testing.assert_allclose(qd(c, 'IE'), [[1 / 3, 0], [1 / 3, 2 / 3]], **tol)
    np.testing.assert_allclose(qd(c, 'IEpar'), [[1 / 3, 0], [0, 1]], **tol)
    np.testing.assert_allclose(qd(c, 'MIN-SR-NS'), [[1 / 6, 0], [0, 1 / 2]], **tol)          # nodes / M
    np.testing.assert_allclose(qd(c, 'Qpar'), [[5 / 12, 0], [0, 1 / 4]], **tol)              # diagonal of Q
    np.testing.assert_allclose(qd(c, 'PIC'), [[0, 0], [0, 0]], **tol)
    EE, dtau = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    np.testing.assert_allclose(EE, [[0, 0], [2 / 3, 0]], **tol)                              # strictly lower: node spacings
    np.testing.assert_allclose(dtau, [1 / 3, 1 / 3], **tol)                                  # first node - left end
    # LU of the three-node Radau IIA matrix, by hand: Q^T = L U, QDelta = U^T (entries from the closed-form tableau)
    c3 = CollBase(3, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    A = c3.Qmat[1:, 1:]
    U = np.array(A.T)
    for i in range(3):
        for r in range(i + 1, 3):
            U[r] -= U[r, i] / U[i, i] * U[i]
    np.testing.assert_allclose(qd(c3, 'LU'), np.triu(U).T, rtol=0, atol=1e-15)
    c = CollBase(2, 0, 1, 'LEGENDRE', 'GAUSS')
    np.testing.assert_allclose(c.nodes, [1 / 2 - s3 / 6, 1 / 2 + s3 / 6], **tol)
    np.testing.assert_allclose(c.Qmat[1:, 1:], [[1 / 4, 1 / 4 - s3 / 6], [1 / 4 + s3 / 6, 1 / 4]], **tol)
    np.testing.assert_allclose(c.weights, [1 / 2, 1 / 2], **tol)
    assert c.order == 4 and not c.right_is_node and not c.left_is_node
    c = CollBase(3, 0, 1, 'LEGENDRE', 'GAUSS')
    np.testing.assert_allclose(c.nodes, [1 / 2 - s15 / 10, 1 / 2, 1 / 2 + s15 / 10], **tol)
    np.testing.assert_allclose(c.Qmat[1:, 1:], [[5 / 36, 2 / 9 - s15 / 15, 5 / 36 - s15 / 30],
                                                [5 / 36 + s15 / 24, 2 / 9, 5 / 36 - s15 / 24],
                                                [5 / 36 + s15 / 30, 2 / 9 + s15 / 15, 5 / 36]], **tol)
    np.testing.assert_allclose(c.weights, [5 / 18, 4 / 9, 5 / 18], **tol)
    c = CollBase(3, 0, 1, 'LEGENDRE', 'LOBATTO')
    np.testing.assert_allclose(c.nodes, [0, 1 / 2, 1], **tol)
    np.testing.assert_allclose(c.Qmat[1:, 1:], [[0, 0, 0], [5 / 24, 1 / 3, -1 / 24], [1 / 6, 2 / 3, 1 / 6]], **tol)
    assert c.order == 4 and c.right_is_node and c.left_is_node
    s5 = np.sqrt(5)
    c = CollBase(4, 0, 1, 'LEGENDRE', 'LOBATTO')
    np.testing.assert_allclose(c.nodes, [0, (5 - s5) / 10, (5 + s5) / 10, 1], **tol)
    np.testing.assert_allclose(c.Qmat[1:, 1:], [[0, 0, 0, 0],
                                                [(11 + s5) / 120, (25 - s5) / 120, (25 - 13 * s5) / 120, (-1 + s5) / 120],
                                                [(11 - s5) / 120, (25 + 13 * s5) / 120, (25 + s5) / 120, (-1 - s5) / 120],
                                                [1 / 12, 5 / 12, 5 / 12, 1 / 12]], **tol)
    c = CollBase(2, 0, 1, 'LEGENDRE', 'RADAU-LEFT')
    np.testing.assert_allclose(c.nodes, [0, 2 / 3], **tol)
    np.testing.assert_allclose(c.Qmat[1:, 1:], [[0, 0], [1 / 3, 1 / 3]], **tol)
    np.testing.assert_allclose(c.weights, [1 / 4, 3 / 4], **tol)
    # the five Radau IIA nodes of the headline configuration (RADAU5-family tables, ten digits)
    c = CollBase(5, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    np.testing.assert_allclose(c.nodes, [0.0571041961, 0.2768430136, 0.5835904324, 0.8602401357, 1.0], rtol=0,
                               atol=5e-11)
    np.testing.assert_allclose(c.weights, [0.1437135608, 0.2813560151, 0.3118265230, 0.2231039011, 0.04], rtol=0,
                               atol=5e-11)
    assert c.order == 9
    # equidistant (Newton-Cotes) nodes: Simpson's rule
    c = CollBase(3, 0, 1, 'EQUID', 'LOBATTO')
    np.testing.assert_allclose(c.weights, [1 / 6, 2 / 3, 1 / 6], **tol)


def test_golden_files_carry_the_coefficients_they_were_made_with():
    """not a pin of coeffs.py (the goldens' matrices came through it, as the qmat stand-in of the generating run) - a
    guard that a later change to coeffs.py does not silently drift from the matrices the stored reference outputs
    belong to."""
    for fname in ('sweeps_heat.npz', 'sweeps_imex.npz', 'runs.npz'):
        for name, case in load_cases(fname).items():
            sp = case['meta']['sweeper_params']
            c = CollBase(sp['num_nodes'], 0, 1, 'LEGENDRE', sp['quad_type'])
            assert np.array_equal(c.Qmat, case['coll_Qmat']) and np.array_equal(c.nodes, case['coll_nodes'])
            QI = np.zeros_like(c.Qmat)
            QI[1:, 1:] = qd(c, sp.get('QI', 'IE'))
            assert np.allclose(QI, case['coll_QI'], rtol=0, atol=1e-15) or 'FLEX' in sp.get('QI', '')
