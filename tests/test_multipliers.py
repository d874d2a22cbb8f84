"""CPU: the algebra behind the sweeps that do not store their iterate (include/sdcmi.h: sdc_set_virtual_sweeps, DESIGN.md
section 3) against the oracle's restatement of the reference's sweeps (generic_implicit.py:51-103, imex_1st_order.py:57-108,
core/sweeper.py:125-215).  After a 'spread' predictor every node value of a linear problem is, per Fourier mode,
u_m = g_m(lam, mu) * u0 and its collocation residual h_m * u0 with the node multipliers the HIP kernels iterate
(kernels_fft.hpp: virt_multipliers_real / virt_multipliers): here the same recurrence in NumPy, held against transforms of
what the oracle's sweeps produce from the same start value."""
import numpy as np
import pytest

from oracle import sdc_oracle as O
from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS


def multipliers(Q, QI, QE, dt, lam, mu, nsweeps):
    """g[m] after nsweeps sweeps from g = 1 (all nodes equal u0), and the residual multipliers h[m]; lam / mu: symbols of
    the implicit / explicit operator (arrays over the modes), Q, QI, QE: (M+1)x(M+1) in the reference's layout."""
    M = Q.shape[0] - 1
    g = [np.ones_like(lam, dtype=complex) for _ in range(M)]
    for _ in range(nsweeps):
        old = [x.copy() for x in g]
        for m in range(M):
            tI = sum(dt * (Q[m + 1, q + 1] - QI[m + 1, q + 1]) * old[q] for q in range(M))
            tI = tI + sum(dt * QI[m + 1, q + 1] * g[q] for q in range(m))
            acc = 1.0 + lam * tI
            if QE is not None:
                tE = sum(dt * (Q[m + 1, q + 1] - QE[m + 1, q + 1]) * old[q] for q in range(M))
                tE = tE + sum(dt * QE[m + 1, q + 1] * g[q] for q in range(m))
                acc = acc + mu * tE
            g[m] = acc / (1.0 - dt * QI[m + 1, m + 1] * lam)
    sym = lam + (mu if QE is not None else 0.0)
    h = [1.0 - g[m] + sym * sum(dt * Q[m + 1, j + 1] * g[j] for j in range(M)) for m in range(M)]
    return g, h


def coefficients(M, qd, imex):
    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    QI = np.zeros_like(c.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS[qd](qGen=c.generator, tLeft=0).genCoeffs()
    QE = None
    if imex:
        QE = np.zeros_like(c.Qmat)
        QE[1:, 1:], QE[1:, 0] = QDELTA_GENERATORS['EE'](qGen=c.generator, tLeft=0).genCoeffs(dTau=True)
    return c, QI, QE


def symbol(prob, n):
    lam1 = prob.nu * O.fd_symbol_1d(2, prob.order, prob.stencil_type, n, prob.dx)
    lam = 0
    for ax in range(prob.ndim):
        shape = [1] * prob.ndim
        shape[ax] = n
        lam = lam + lam1.reshape(shape)
    return lam


@pytest.mark.parametrize('M,qd', [(3, 'IE'), (5, 'IE'), (5, 'LU'), (4, 'MIN-SR-S')])
@pytest.mark.parametrize('ndim,n', [(1, 32), (2, 16)])
def test_heat_iterates_are_multiples_of_the_start_value(ndim, n, M, qd):
    c, QI, _ = coefficients(M, qd, False)
    P = O.HeatUnforced((n,) * ndim, 0.1, 2, order=2)
    dt = 0.05
    L = O.Level(P, O.Coll(c.nodes, c.weights, c.Qmat, QI), dt)
    L.time = 0.0
    u0 = np.random.default_rng(5).standard_normal((n,) * ndim)
    L.u[0] = u0.copy()
    O.predict(L, 'spread')
    lam = symbol(P, n)
    assert np.max(np.abs(lam.imag)) < 1e-9 * np.max(np.abs(lam.real))      # symmetric stencil: real symbol ...
    if ndim == 1:
        assert np.allclose(lam.real[1:], lam.real[1:][::-1], rtol=1e-13)     # ... and lam(k) = lam(n - k): mode pairs
    u0h = np.fft.fftn(u0)
    scale = np.max(np.abs(u0h))
    for k in range(1, 5):
        O.sweep_generic_implicit(L)
        O.compute_residual(L)
        g, h = multipliers(c.Qmat, QI, None, dt, lam.real, None, k)
        for m in range(M):
            assert np.max(np.abs(g[m].imag)) == 0.0                          # real symbol: real multipliers
            assert np.max(np.abs(np.fft.fftn(L.u[m + 1]) - g[m] * u0h)) < 1e-12 * scale, (k, m)
            assert np.max(np.abs(np.fft.fftn(L.residual[m]) - h[m] * u0h)) < 1e-12 * scale, (k, m)


@pytest.mark.parametrize('M,qd', [(3, 'IE'), (5, 'LU')])
def test_imex_iterates_are_multiples_of_the_start_value(M, qd):
    n, dt = 32, 0.02
    c, QI, QE = coefficients(M, qd, True)
    P = O.AdvectionDiffusionIMEX((n,), nu=0.05, c=0.7, freq=2, order=2)
    L = O.Level(P, O.Coll(c.nodes, c.weights, c.Qmat, QI, QE), dt)
    L.time = 0.0
    u0 = np.random.default_rng(6).standard_normal((n,))
    L.u[0] = u0.copy()
    O.predict(L, 'spread')
    lam = P.diff.nu * O.fd_symbol_1d(2, P.diff.order, P.diff.stencil_type, n, P.diff.dx)
    mu = -P.adv.c * O.fd_symbol_1d(1, P.adv.order, P.adv.stencil_type, n, P.adv.dx)
    u0h = np.fft.fft(u0)
    scale = np.max(np.abs(u0h))
    for k in range(1, 4):
        O.sweep_imex(L)
        O.compute_residual(L)
        g, h = multipliers(c.Qmat, QI, QE, dt, lam, mu, k)
        for m in range(M):
            assert np.max(np.abs(np.fft.fft(L.u[m + 1]) - g[m] * u0h)) < 1e-12 * scale, (k, m)
            assert np.max(np.abs(np.fft.fft(L.residual[m]) - h[m] * u0h)) < 1e-12 * scale, (k, m)
