"""Time slabs on awkward shapes: ny, nx not multiples of the tile sizes, nt not divisible by the slab count, up to six
slabs of unequal length -- all three loops against their single-slab runs."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle.examples import get_example_2d

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")
SOLVERS = {"inPALM": D.solver_socp_inPALM, "PALM": D.solver_socp_PALM, "acc-ADMM": D.solver_socp_accADMM}


def _relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("method", ["inPALM", "PALM", "acc-ADMM"])
@pytest.mark.parametrize("ny,nx,nt,nslabs", [(70, 50, 37, 5), (65, 130, 53, 6), (129, 31, 26, 2)])
def test_unequal_slabs(ny, nx, nt, nslabs, method):
    rho0, rho1 = get_example_2d("example1", ny, nx)
    res = []
    for ns in (1, nslabs):
        var, model = D.initialize(rho0, rho1, nt)
        oo = OD.default_opts(dict(tol=0.0, maxit=35), method, False)
        D.InitialScaling(var, model, oo["scaling"], None, dim=2)
        hist, sigma = SOLVERS[method](var, oo, model, nslabs=ns)
        res.append((var, hist, sigma))
    (ref, h1, s1), (got, hn, sn) = res
    np.testing.assert_array_equal(hn["iter"], h1["iter"])
    np.testing.assert_allclose(hn["kkt"], h1["kkt"], rtol=1e-7, atol=1e-10)
    assert abs(sn - s1) <= 1e-12 * s1
    errs = {f: _relerr(getattr(got, f), getattr(ref, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-10, errs
