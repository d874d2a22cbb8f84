"""TEST INFRASTRUCTURE ONLY - stand-in for the third-party module ``qmat`` (absent from this
container, pin qmat>=0.1.19 in /root/reference/pyproject.toml:34) so that the reference can be
imported *in the build container* to generate golden vectors (tests/golden/gen_golden.py).
It forwards to pysdc_amd.coeffs, which restates qmat's published algorithms.  Never imported
by the product path and never needed on the GPU box."""
from pysdc_amd import coeffs as _c


class _QGenerator:
    @property
    def S(self):
        M = self.Q.copy()
        M[1:] -= self.Q[:-1]
        return M


class Collocation(_QGenerator, _c.Collocation):
    pass


Q_GENERATORS = {'Collocation': Collocation, 'coll': Collocation}
