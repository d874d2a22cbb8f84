"""GPU: lines of 3 * 2^p and 5 * 2^p modes (generic_ND_FD.py:125-133 accepts any even n) take the headline data flow since
round 6 - iterates recomputed from the transform of u[0] through node multipliers, only the residual lines written
(k_spec_z MODE 3; MODE 4 with mode pairs at n = 768).  Against the stored-iterate flow of the same engine (whose sweeps are
pinned to the reference at these line lengths by tests/golden/sweeps_radix3.npz / sweeps_radix5.npz): residuals after every
sweep, node values and right-hand sides once they are asked for, end values over two steps."""
import ctypes as C

import numpy as np
import pytest

from pysdc_amd import lib as L
from tests import _gpu as G
from tests._cases import rel_err

pytestmark = pytest.mark.gpu


def _coeffs(M, QI):
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS

    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    qi = np.zeros_like(c.Qmat)
    qi[1:, 1:] = QDELTA_GENERATORS[QI](qGen=c.generator, tLeft=0).genCoeffs()
    return c, qi


@pytest.mark.parametrize('nvars,M,QI', [((96, 96, 96), 5, 'IE'), ((80, 80, 80), 3, 'LU'), ((192, 192), 5, 'IE'), ((160, 160), 5, 'LU'),
                                        ((384, 384), 3, 'IE'), ((320, 320), 5, 'IE'), ((768, 768), 5, 'IE'), ((640, 640), 5, 'LU'),
                                        ((768, 768), 2, 'LU')])
def test_recomputed_iterates_equal_stored_ones_on_odd_factor_lines(nvars, M, QI):
    n = nvars[0]
    dt = 1e-3 * (512.0 / n) ** 2
    c, qi = _coeffs(M, QI)
    engines = []
    for virtual in (16, 0):
        e = G.engine_for('heat_unforced', dict(nvars=nvars, nu=0.1), M)
        e.set_coeffs(c.Qmat, qi, None, c.nodes, c.weights)
        e.set_virtual_sweeps(virtual)
        freq = (C.c_int * 3)(2, 4, 2)
        L.check(e.lib.sdc_init_field(e.ctx, e.ptr(L.SLOT_U, 0), freq, 0.3, 5), e.ctx)
        e.invalidate_spectra(1)
        e.profile_enable(True)
        engines.append(e)
    a, b = engines
    for step in range(2):
        for e in engines:
            e.predict(0.0, dt)
        for k in range(4):
            for e in engines:
                e.sweep(0.0, dt)
            ra, rb = a.residual(dt), b.residual(dt)
            assert abs(ra[0] - rb[0]) <= 1e-10 * abs(rb[0]) + 1e-14, (step, k, ra[0], rb[0])
            np.testing.assert_allclose(ra[1], rb[1], rtol=1e-10, atol=1e-14)
        if step == 0:
            for e in engines:
                e.end_point(dt, False)
                e.advance()
    assert rel_err(a.download_u(), b.download_u()) < 1e-12      # (asking stores the iterate that was never stored)
    assert rel_err(a.download_f(), b.download_f()) < 1e-10
    for e in engines:
        e.end_point(dt, False)
    assert rel_err(a.download(L.SLOT_UEND), b.download(L.SLOT_UEND)) < 1e-12
    pa, pb = a.profile_read(), b.profile_read()
    # the recomputing launches ran on the one engine and not on the other; with mode pairs (and the end spectrum on the way) at 768
    assert any(k.startswith('spec_z_res_v') for k in pa) and not any(k.startswith('spec_z_res_v') for k in pb), (sorted(pa), sorted(pb))
    assert any(k.startswith('spec_point') for k in pb)
    for e in engines:
        e.close()
