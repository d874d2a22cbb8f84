"""TEST INFRASTRUCTURE ONLY - see oracle/qmat_shim/qmat/__init__.py."""
from pysdc_amd.coeffs import LagrangeApproximation  # noqa: F401
