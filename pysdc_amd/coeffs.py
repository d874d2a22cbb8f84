"""Collocation and Q-Delta coefficient generation (host side, NumPy).

The reference obtains every quadrature coefficient from the third-party module
``qmat`` (pin ``qmat>=0.1.19``, /root/reference/pyproject.toml:34), which is not
vendored and not installed here.  This module restates qmat's published
algorithms for the entry points pySDC binds:

* ``Q_GENERATORS["Collocation"]``  -> :class:`Collocation`
  (call site: pySDC/core/collocation.py:73-100)
* ``QDELTA_GENERATORS[...]``        -> :data:`QDELTA_GENERATORS`
  (call sites: pySDC/core/sweeper.py:14-16,98,104,117,271-276)
* ``LagrangeApproximation``         -> :class:`LagrangeApproximation`
  (call site: pySDC/core/base_transfer.py:90-91)

Values are pinned by the reference's property tests (quadrature exactness,
nilpotency, structure) and closed-form Radau-IIA tableaux, not bit-wise
("parity unpinned" at the coefficient level).
"""

import warnings

import numpy as np
import scipy.linalg as spl
import scipy.optimize as spo

NODE_TYPES = ['EQUID', 'LEGENDRE', 'CHEBY-1', 'CHEBY-2', 'CHEBY-3', 'CHEBY-4']
QUAD_TYPES = ['GAUSS', 'RADAU-LEFT', 'RADAU-RIGHT', 'LOBATTO']


# --------------------------------------------------------------------------------------
# nodes on [-1, 1]
# --------------------------------------------------------------------------------------
def _recurrence(node_type, n):
    """Three-term recurrence coefficients (alpha_k, beta_k), k < n, of the monic
    orthogonal polynomials for the Jacobi weight belonging to ``node_type``."""
    k = np.arange(n, dtype=float)
    alpha = np.zeros(n)
    beta = np.zeros(n)
    if node_type == 'LEGENDRE':
        beta[0] = 2.0
        beta[1:] = 1.0 / (4.0 - 1.0 / k[1:] ** 2)
    elif node_type == 'CHEBY-1':
        beta[0] = np.pi
        if n > 1:
            beta[1] = 0.5
        beta[2:] = 0.25
    elif node_type == 'CHEBY-2':
        beta[0] = np.pi / 2
        beta[1:] = 0.25
    elif node_type == 'CHEBY-3':
        alpha[0] = 0.5
        beta[0] = np.pi
        beta[1:] = 0.25
    elif node_type == 'CHEBY-4':
        alpha[0] = -0.5
        beta[0] = np.pi
        beta[1:] = 0.25
    else:
        raise ValueError(f'unknown node_type {node_type}')
    return alpha, beta


def _eval_monic(alpha, beta, n, x):
    """p_{n-1}(x), p_n(x) of the monic family."""
    pm1, p = 0.0, 1.0
    for k in range(n):
        pm1, p = p, (x - alpha[k]) * p - (beta[k] if k > 0 else 0.0) * pm1
    return pm1, p


def _jacobi_eig(alpha, beta):
    n = len(alpha)
    J = np.diag(alpha)
    if n > 1:
        off = np.sqrt(beta[1:n])
        J += np.diag(off, 1) + np.diag(off, -1)
    return np.sort(np.linalg.eigvalsh(J))


def _legendre_polish(nodes, M, quad_type):
    """Newton polish of Legendre-type nodes in extended precision."""
    from numpy.polynomial import legendre as L

    x = np.asarray(nodes, dtype=np.longdouble)
    cM = np.zeros(M + 1)
    cM[M] = 1.0
    if quad_type == 'GAUSS':
        c = cM
    elif quad_type in ('RADAU-RIGHT', 'RADAU-LEFT'):
        c = cM.copy()
        c[M - 1] = -1.0 if quad_type == 'RADAU-RIGHT' else 1.0  # P_M -/+ P_{M-1}
    else:  # LOBATTO: (1-x^2) P'_{M-1}; polish interior roots on P'_{M-1}
        cm = np.zeros(M)
        cm[M - 1] = 1.0
        c = L.legder(cm)
    dc = L.legder(c)
    c = np.asarray(c, dtype=np.longdouble)
    dc = np.asarray(dc, dtype=np.longdouble)
    fixed = np.zeros(len(x), dtype=bool)
    if quad_type in ('RADAU-RIGHT', 'LOBATTO'):
        fixed[-1] = True
    if quad_type in ('RADAU-LEFT', 'LOBATTO'):
        fixed[0] = True
    for _ in range(3):
        fx = L.legval(x, c)
        dfx = L.legval(x, dc)
        step = np.where(fixed, 0.0, fx / np.where(dfx == 0, 1.0, dfx))
        x = x - step
    x[fixed] = np.round(x[fixed])
    return np.asarray(x, dtype=float)


def gen_nodes(num_nodes, node_type='LEGENDRE', quad_type='RADAU-RIGHT'):
    """Nodes on [-1, 1] (Golub-Welsch, with the Radau / Lobatto modification of the
    Jacobi matrix).  Restates qmat.nodes.NodesGenerator.getNodes."""
    M = num_nodes
    if node_type not in NODE_TYPES:
        raise ValueError(f'node_type {node_type} not in {NODE_TYPES}')
    if quad_type not in QUAD_TYPES:
        raise ValueError(f'quad_type {quad_type} not in {QUAD_TYPES}')
    if node_type == 'EQUID':
        if quad_type == 'GAUSS':
            return np.linspace(-1, 1, M + 2)[1:-1]
        if quad_type == 'LOBATTO':
            if M < 2:
                raise ValueError('LOBATTO needs at least 2 nodes')
            return np.linspace(-1, 1, M)
        if quad_type == 'RADAU-RIGHT':
            return np.linspace(-1, 1, M + 1)[1:]
        return np.linspace(-1, 1, M + 1)[:-1]

    alpha, beta = _recurrence(node_type, M + 1)
    if quad_type == 'GAUSS':
        nodes = _jacobi_eig(alpha[:M], beta[:M])
    elif quad_type in ('RADAU-RIGHT', 'RADAU-LEFT'):
        x0 = 1.0 if quad_type == 'RADAU-RIGHT' else -1.0
        a = alpha[:M].copy()
        if M == 1:
            nodes = np.array([x0])
        else:
            pm1, p = _eval_monic(alpha, beta, M - 1, x0)
            a[M - 1] = x0 - beta[M - 1] * pm1 / p
            nodes = _jacobi_eig(a, beta[:M])
            nodes[-1 if x0 > 0 else 0] = x0
    else:  # LOBATTO
        if M < 2:
            raise ValueError('LOBATTO needs at least 2 nodes')
        a = alpha[:M].copy()
        b = beta[:M].copy()
        pm1L, pL = _eval_monic(alpha, beta, M - 1, -1.0)
        pm1R, pR = _eval_monic(alpha, beta, M - 1, 1.0)
        sol = np.linalg.solve(np.array([[pL, pm1L], [pR, pm1R]]), np.array([-pL, pR]))
        a[M - 1], b[M - 1] = sol
        nodes = _jacobi_eig(a, b)
        nodes[0], nodes[-1] = -1.0, 1.0
    if node_type == 'LEGENDRE':
        nodes = _legendre_polish(nodes, M, quad_type)
    return nodes


# --------------------------------------------------------------------------------------
# Lagrange basis: integration and interpolation matrices
# --------------------------------------------------------------------------------------
class LagrangeApproximation:
    """Barycentric Lagrange interpolant on ``points`` (restates
    qmat.lagrange.LagrangeApproximation as bound at pySDC/core/base_transfer.py:90-91)."""

    def __init__(self, points):
        pts = np.asarray(points, dtype=np.longdouble).ravel()
        n = pts.size
        diffs = pts[:, None] - pts[None, :]
        diffs[np.arange(n), np.arange(n)] = 1.0
        # scale to avoid over/underflow for many points
        scale = 4.0 / max(float(pts.max() - pts.min()), np.finfo(float).tiny) if n > 1 else 1.0
        w = 1.0 / np.prod(diffs * scale, axis=1)
        self._pts = pts
        self._w = w / np.max(np.abs(w))
        self.points = np.asarray(pts, dtype=float)
        self.weights = np.asarray(self._w, dtype=float)
        self.n = n

    def _basis(self, t):
        """Lagrange basis values, shape (len(t), n), extended precision."""
        t = np.asarray(t, dtype=np.longdouble).ravel()
        d = t[:, None] - self._pts[None, :]
        hit = d == 0
        d[hit] = 1.0
        terms = self._w[None, :] / d
        B = terms / np.sum(terms, axis=1)[:, None]
        rows = np.any(hit, axis=1)
        if np.any(rows):
            B[rows] = 0.0
            B[hit] = 1.0
        return B

    def getInterpolationMatrix(self, times):
        return np.asarray(self._basis(times), dtype=float)

    def getIntegrationMatrix(self, intervals):
        """rows = intervals (a, b), cols = basis functions: int_a^b l_j(t) dt.
        Gauss-Legendre with n//2+1 points is exact for the degree n-1 basis."""
        nq = self.n // 2 + 1
        xq, wq = np.polynomial.legendre.leggauss(nq)
        xq = np.asarray(xq, dtype=np.longdouble)
        wq = np.asarray(wq, dtype=np.longdouble)
        out = np.zeros((len(intervals), self.n), dtype=np.longdouble)
        for i, (a, b) in enumerate(intervals):
            a = np.longdouble(a)
            b = np.longdouble(b)
            if a == b:
                continue
            t = 0.5 * (b - a) * xq + 0.5 * (a + b)
            out[i] = 0.5 * (b - a) * (wq @ self._basis(t))
        return np.asarray(out, dtype=float)


class Collocation:
    """Collocation coefficients on [tLeft, tRight] (restates the attributes of
    qmat.qcoeff.collocation.Collocation that pySDC/core/collocation.py:73-100 reads:
    ``nodes, weights, Q, S (parent definition: row differences of Q), order``)."""

    def __init__(self, nNodes=None, nodeType='LEGENDRE', quadType='RADAU-RIGHT', tLeft=0.0, tRight=1.0):
        if nNodes is None or not nNodes > 0:
            raise ValueError(f'at least one quadrature node required, got {nNodes}')
        if not tLeft < tRight:
            raise ValueError(f'interval boundaries are corrupt, got {tLeft} and {tRight}')
        self.nNodes = nNodes
        self.nodeType = nodeType
        self.quadType = quadType
        self.tLeft = tLeft
        self.tRight = tRight
        ref = gen_nodes(nNodes, nodeType, quadType)
        a = (tRight - tLeft) / 2.0
        b = (tRight + tLeft) / 2.0
        self.nodes = a * ref + b
        if quadType in ('RADAU-LEFT', 'LOBATTO'):
            self.nodes[0] = tLeft
        if quadType in ('RADAU-RIGHT', 'LOBATTO'):
            self.nodes[-1] = tRight
        approx = LagrangeApproximation(self.nodes)
        self.weights = approx.getIntegrationMatrix([(tLeft, tRight)]).ravel()
        self.Q = approx.getIntegrationMatrix([(tLeft, tau) for tau in self.nodes])

    @property
    def S(self):
        """node-to-node matrix as row differences of Q (the definition pySDC uses,
        pySDC/core/collocation.py:98-105)."""
        S = self.Q.copy()
        S[1:] -= self.Q[:-1]
        return S

    @property
    def order(self):
        M = self.nNodes
        if self.nodeType != 'LEGENDRE':
            return M
        return {'GAUSS': 2 * M, 'RADAU-LEFT': 2 * M - 1, 'RADAU-RIGHT': 2 * M - 1, 'LOBATTO': 2 * M - 2}[self.quadType]

    @property
    def deltas(self):
        d = np.empty(self.nNodes)
        d[0] = self.nodes[0] - self.tLeft
        d[1:] = np.diff(self.nodes)
        return d


# --------------------------------------------------------------------------------------
# Q-Delta generators
# --------------------------------------------------------------------------------------
class QDeltaGenerator:
    """Base class mirroring what pySDC/core/sweeper.py:98-123,262-276 calls:
    ``Generator(qGen=coll, tLeft=...)``, ``genCoeffs(k=None, dTau=False)``,
    ``isKDependent()``."""

    aliases = ()
    k_dependent = False

    def __init__(self, qGen=None, tLeft=0.0, **kwargs):
        self.coll = qGen
        self.Q = np.asarray(qGen.Q, dtype=float)
        self.nodes = np.asarray(qGen.nodes, dtype=float)
        self.nNodes = len(self.nodes)
        self.tLeft = tLeft
        self.quadType = getattr(qGen, 'quadType', 'RADAU-RIGHT')
        self.nodeType = getattr(qGen, 'nodeType', 'LEGENDRE')

    def isKDependent(self):
        return self.k_dependent

    def computeQDelta(self, k=None):
        raise NotImplementedError

    @property
    def dTau(self):
        return np.zeros(self.nNodes)

    def genCoeffs(self, k=None, dTau=False):
        QD = np.array(self.computeQDelta(k), dtype=float)
        if dTau:
            return QD, self.dTau.copy()
        return QD

    @property
    def _deltas(self):
        d = np.empty(self.nNodes)
        d[0] = self.nodes[0] - self.tLeft
        d[1:] = np.diff(self.nodes)
        return d


class BE(QDeltaGenerator):
    """implicit Euler: lower triangular, column j carries delta_j."""

    aliases = ('BE', 'IE')

    def computeQDelta(self, k=None):
        M = self.nNodes
        QD = np.zeros((M, M))
        d = self._deltas
        for i in range(M):
            QD[i, : i + 1] = d[: i + 1]
        return QD


class FE(QDeltaGenerator):
    """explicit Euler: strictly lower triangular, column j carries delta_{j+1};
    the distance tLeft -> first node goes to ``dTau`` (pySDC stores it in column 0,
    pySDC/core/sweeper.py:117)."""

    aliases = ('FE', 'EE')

    def computeQDelta(self, k=None):
        M = self.nNodes
        QD = np.zeros((M, M))
        d = self._deltas
        for i in range(1, M):
            QD[i, :i] = d[1 : i + 1]
        return QD

    @property
    def dTau(self):
        return np.full(self.nNodes, self.nodes[0] - self.tLeft)


class TRAP(QDeltaGenerator):
    aliases = ('TRAP', 'CN')

    def computeQDelta(self, k=None):
        M = self.nNodes
        d = self._deltas
        QD = np.zeros((M, M))
        for i in range(M):
            QD[i, : i + 1] += 0.5 * d[: i + 1]
            QD[i, :i] += 0.5 * d[1 : i + 1]
        return QD

    @property
    def dTau(self):
        return np.full(self.nNodes, 0.5 * (self.nodes[0] - self.tLeft))


class LU(QDeltaGenerator):
    """U^T of the LU decomposition of Q^T (Weiser 2015)."""

    aliases = ('LU',)

    def computeQDelta(self, k=None):
        _, _, U = spl.lu(self.Q.T)
        return U.T


class LU2(LU):
    aliases = ('LU2',)

    def computeQDelta(self, k=None):
        return 2.0 * super().computeQDelta(k)


class PIC(QDeltaGenerator):
    aliases = ('PIC',)

    def computeQDelta(self, k=None):
        return np.zeros((self.nNodes, self.nNodes))


class GS(QDeltaGenerator):
    aliases = ('GS',)

    def computeQDelta(self, k=None):
        return np.tril(self.Q)


class BEpar(QDeltaGenerator):
    aliases = ('BEpar', 'IEpar')

    def computeQDelta(self, k=None):
        return np.diag(self.nodes - self.tLeft)


class Qpar(QDeltaGenerator):
    aliases = ('Qpar', 'Jacobi')

    def computeQDelta(self, k=None):
        return np.diag(np.diag(self.Q))


class MIN_SR_NS(QDeltaGenerator):
    aliases = ('MIN-SR-NS', 'MIN_SR_NS')

    def computeQDelta(self, k=None):
        return np.diag(self.nodes - self.tLeft) / self.nNodes


class MIN_SR_S(QDeltaGenerator):
    """Diagonal coefficients making I - D^{-1} Q nilpotent (Caklovic, Lunet, Goetschel,
    Ruprecht 2024): det((1-z) I + z D^{-1} Q) = 1 at z = nodes, solved incrementally in
    the number of nodes with a power-law extrapolated initial guess."""

    aliases = ('MIN-SR-S', 'MIN_SR_S')

    def _coeffs_for(self, m, a=None, b=None):
        coll = Collocation(m, self.nodeType, self.quadType, self.tLeft, self.tLeft + 1.0)
        # work on the unit interval, scale afterwards (coefficients scale linearly)
        Qm, nodes = coll.Q, coll.nodes - self.tLeft
        zero_first = self.quadType in ('LOBATTO', 'RADAU-LEFT')
        if zero_first:
            Qm, nodes = Qm[1:, 1:], nodes[1:]
        n = len(nodes)
        if n == 1:
            coeffs = np.diag(Qm).copy()
        else:

            def nilpotency(c):
                c = np.asarray(c)
                DinvQ = Qm / c[:, None]
                return np.array([np.linalg.det((1 - z) * np.eye(n) + z * DinvQ) - 1.0 for z in nodes])

            c0 = nodes / m if a is None else a * nodes**b / m
            with warnings.catch_warnings():
                warnings.simplefilter('ignore', RuntimeWarning)
                coeffs = spo.fsolve(nilpotency, c0, xtol=1e-15)
        if zero_first:
            coeffs = np.concatenate([[0.0], coeffs])
            nodes = np.concatenate([[0.0], nodes])
        return coeffs, nodes

    @staticmethod
    def _fit(coeffs, nodes):
        sel = nodes > 0

        def law(ab):
            return np.linalg.norm(ab[0] * nodes[sel] ** ab[1] - coeffs[sel])

        return spo.minimize(law, [1.0, 1.0], method='nelder-mead').x

    def computeQDelta(self, k=None):
        a = b = None
        m0 = 2 if self.quadType in ('LOBATTO', 'RADAU-LEFT') else 1
        coeffs = None
        for m in range(m0, self.nNodes + 1):
            coeffs, nodes = self._coeffs_for(m, a, b)
            if m > 1:
                a, b = self._fit(coeffs * m, nodes)
        scale = self.coll.tRight - self.coll.tLeft if hasattr(self.coll, 'tRight') else 1.0
        return np.diag(coeffs) * scale


class FLEX(MIN_SR_S):
    """k-dependent diagonal: nodes / k for sweeps k = 1..M, MIN-SR-S afterwards."""

    aliases = ('MIN-SR-FLEX', 'FLEX')
    k_dependent = True

    def computeQDelta(self, k=None):
        if k is None:
            k = 1
        if k < 1:
            raise ValueError(f'k must be >= 1, got {k}')
        if k <= self.nNodes:
            return np.diag(self.nodes - self.tLeft) / k
        return super().computeQDelta()


QDELTA_GENERATORS = {}
for _cls in (BE, FE, TRAP, LU, LU2, PIC, GS, BEpar, Qpar, MIN_SR_NS, MIN_SR_S, FLEX):
    QDELTA_GENERATORS[_cls.__name__] = _cls
    for _al in _cls.aliases:
        QDELTA_GENERATORS[_al] = _cls

Q_GENERATORS = {'Collocation': Collocation, 'coll': Collocation}


# --------------------------------------------------------------------------------------
# pySDC-shaped collocation object (pySDC/core/collocation.py:8-141)
# --------------------------------------------------------------------------------------
class CollBase:
    """Same attributes as the reference's ``CollBase`` (pySDC/core/collocation.py:48-108):
    ``num_nodes, tleft, tright, node_type, quad_type, left_is_node, right_is_node, order,
    nodes, weights, Qmat, Smat, delta_m`` with (M+1)x(M+1) zero-padded Qmat/Smat."""

    def __init__(self, num_nodes=None, tleft=0, tright=1, node_type='LEGENDRE', quad_type=None, **kwargs):
        from pysdc_amd.errors import CollocationError

        if num_nodes is None or not num_nodes > 0:
            raise CollocationError('at least one quadrature node required, got %s' % num_nodes)
        if not tleft < tright:
            raise CollocationError('interval boundaries are corrupt, got %s and %s' % (tleft, tright))
        try:
            self.generator = Collocation(
                nNodes=num_nodes, nodeType=node_type, quadType=quad_type, tLeft=tleft, tRight=tright
            )
        except Exception as e:
            raise CollocationError(f'could not instantiate collocation generator, got error: {e}') from e
        self.num_nodes = num_nodes
        self.tleft = tleft
        self.tright = tright
        self.node_type = node_type
        self.quad_type = quad_type
        self.left_is_node = quad_type in ['LOBATTO', 'RADAU-LEFT']
        self.right_is_node = quad_type in ['LOBATTO', 'RADAU-RIGHT']
        self.order = self.generator.order
        self.nodes = self.generator.nodes.copy()
        self.weights = self.generator.weights.copy()
        Q = np.zeros([num_nodes + 1, num_nodes + 1])
        Q[1:, 1:] = self.generator.Q
        self.Qmat = Q
        S = np.zeros([num_nodes + 1, num_nodes + 1])
        S[1:, 1:] = self.generator.S
        self.Smat = S
        self.delta_m = self.generator.deltas

    @staticmethod
    def evaluate(weights, data):
        from pysdc_amd.errors import CollocationError

        if not np.size(weights) == np.size(data):
            raise CollocationError('Input size does not match number of weights, but is %s' % np.size(data))
        return np.dot(weights, data)
