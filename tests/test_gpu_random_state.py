"""PALM and accelerated ADMM from a RANDOM state: phi, q, z, alpha, beta are generic -- no term
of a q-step, of a multiplier update or of a KKT sum vanishes as it does after the all-zero start of initialize.m -- and a
few iterations with a KKT check in each are compared with the CPU restatement of the loop field by field, on one slab and on
time slabs.  (The inPALM twin is tests/test_gpu_solver.py::test_iterations_from_a_random_state.)  z and beta are zero where
a cone row has no edge (the slots mexBFd leaves unwritten at the domain boundary): the reference never makes those
entries non-zero, and its KKT block reads what an earlier projection left in such slots of a shared temporary
(solver_socp_PALM.m:251,253, solver_socp_accADMM.m:269,271: mexProjSoc into z2, then mexBFd into the same z2)."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle import mexops
from oracle.examples import get_example_2d

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")


def _relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("split", ["nslabs=1", "nslabs=2", "nslabs=3", "ngpu=2", "ngpu=3"])
@pytest.mark.parametrize("K", [1, 3])
@pytest.mark.parametrize("method", ["PALM", "acc-ADMM", "inPALM"])
def test_iterations_from_a_random_state(method, K, split):
    """split: nslabs = time slabs on one device sharing a stream pair; ngpu = dotsocp_create_multi (a stream pair per slab,
    event-ordered messages; the devices wrap around on a one-GPU box)"""
    kind, n = split.split("=")
    if method == "inPALM" and kind == "nslabs":
        pytest.skip("covered by tests/test_gpu_solver.py::test_iterations_from_a_random_state")
    rng = np.random.default_rng(23)
    ny, nx, nt = 20, 28, 9
    rho0, rho1 = get_example_2d("example1", ny, nx)
    opts = dict(tol=0.0, maxit=K, ifCheckStepByStep=True, sigma=0.7)
    ovar, omodel, oo = OD.make_level(rho0, rho1, nt, opts, method)
    gvar, gmodel = D.initialize(rho0, rho1, nt)
    D.InitialScaling(gvar, gmodel, oo["scaling"], None, dim=2)
    start = {"phi": rng.standard_normal(ovar.phi.shape), "q": 0.3 * rng.standard_normal(ovar.q.shape),
             "alpha": 0.5 * rng.standard_normal(ovar.alpha.shape),
             "z": np.asfortranarray(0.6 * rng.standard_normal(ovar.z.shape)),
             "beta": np.asfortranarray(0.4 * rng.standard_normal(ovar.beta.shape))}
    probe = np.full(ovar.beta.shape, np.nan, order="F")
    mexops.mexBFd(probe, start["q"], nt, nx, ny)
    hole = np.isnan(probe)
    assert 0 < hole.sum() < probe.size // 4
    start["z"][hole] = 0.0
    start["beta"][hole] = 0.0
    for v in (ovar, gvar):
        for f, a in start.items():
            setattr(v, f, a.copy(order="F"))
    st = OD.make_state(ovar, oo, omodel, method)
    st.run()
    o_hist, o_sigma = st.finish()
    ctx = D.InPALMContext(gvar, oo, gmodel, method=method, **{kind: int(n)})
    try:
        ctx.run(-1)
        g_hist, g_sigma = ctx.finish()
    finally:
        ctx.close()
    assert g_hist["len"] == o_hist["len"] == K
    assert abs(g_sigma - o_sigma) <= 1e-12 * abs(o_sigma)
    np.testing.assert_allclose(g_hist["kkt"], o_hist["kkt"], rtol=1e-8, atol=1e-12)
    errs = {f: _relerr(getattr(gvar, f), getattr(ovar, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-11, errs
    assert _relerr(gvar.q, start["q"]) > 1e-2 and _relerr(gvar.beta, start["beta"]) > 1e-2
