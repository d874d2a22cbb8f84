"""TEST INFRASTRUCTURE ONLY - Runge-Kutta tableaux are out of scope; empty registry so that
``from qmat.qcoeff.butcher import RK_SCHEMES`` in the reference does not fail on import."""
RK_SCHEMES = {}
