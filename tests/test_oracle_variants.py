"""The CPU restatements of the PALM and acc-ADMM loops (oracle/palm.py, oracle/accadmm.py; SURVEY.md 8f rows
1 and 4).  PARITY UNPINNED -- the reference has no fixture for them -- so they are checked through what the
problem itself pins: every loop must reach the KKT tolerance, conserve mass and arrive at the transport the
(golden-fixture-checked) inPALM oracle finds."""
import numpy as np
import pytest

from oracle import driver as OD
from oracle.accadmm import AccADMMState
from oracle.examples import get_example_2d


@pytest.fixture(scope="module")
def inpalm_solution():
    rho0, rho1 = get_example_2d("example1", 16, 16)
    var, model, hist, sigma = OD.solve_single_level(rho0, rho1, 8, dict(tol=1e-4), "inPALM")
    return rho0, rho1, OD.recover_RhoE(var, model)[0]


@pytest.mark.parametrize("method,extra", [("PALM", {}), ("acc-ADMM", {}), ("acc-ADMM", dict(theta=3.0, restart=20)),
                                           ("acc-ADMM", dict(rho=1.6, restart=7)), ("ALG2", {})])
def test_variant_reaches_the_same_transport(inpalm_solution, method, extra):
    rho0, rho1, rho_ref = inpalm_solution
    var, model, hist, sigma = OD.solve_single_level(rho0, rho1, 8, dict(tol=1e-4, **extra), method)
    assert np.max(hist["kkt"][-1][[0, 2, 5, 6]]) < 1e-4
    assert hist["iter"][-1] < 3000
    rho = OD.recover_RhoE(var, model)[0]
    assert OD.check_massConservation(rho, 1e-2)[0]
    assert np.max(np.abs(rho - rho_ref)) < 0.05 * np.max(rho_ref)


def test_accadmm_with_rho1_restart1_is_plain_admm():
    """rho = 1, restart = 1: x = (x0 + x^+)/2 with the anchor reset to x after every step -- a damped ADMM;
    the Halpern weights are c1 = c2 = 1/2 in every iteration (solver_socp_accADMM.m:373-388)."""
    rho0, rho1 = get_example_2d("example1", 16, 16)
    var, model, o = OD.make_level(rho0, rho1, 8, dict(tol=0.0, maxit=5, restart=1, rho=1.0), "acc-ADMM")
    st = AccADMMState(var, o, model)
    for _ in range(5):
        old = st._copy()
        assert st.k == 0
        st.step()
        for f in ("phi", "q"):
            assert np.all(np.isfinite(getattr(st, f)))
        # anchors were re-set to the new state
        np.testing.assert_array_equal(st.anchor["q"], st.q)
        assert not np.array_equal(old["q"], st.q)
