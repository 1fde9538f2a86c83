"""the proximal ALM loop (solver_socp_PALM.m) on the device against its CPU
restatement (oracle/palm.py, PARITY UNPINNED: no fixture in the reference).  Same bar as inPALM."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle.examples import get_example_2d
from oracle.palm import PALMState

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")


def _relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _compare(rho0, rho1, nt, opts, K, tol=1e-9, nslabs=1):
    opts = dict(opts, maxit=K)
    ovar, omodel, oo = OD.make_level(rho0, rho1, nt, opts, "PALM")
    st = PALMState(ovar, oo, omodel)
    st.run()
    o_hist, o_sigma = st.finish()
    gvar, gmodel = D.initialize(rho0, rho1, nt)
    D.InitialScaling(gvar, gmodel, oo["scaling"], None, dim=2)
    g_hist, g_sigma = D.solver_socp_PALM(gvar, oo, gmodel, nslabs=nslabs)
    assert g_hist["len"] == o_hist["len"]
    np.testing.assert_array_equal(g_hist["iter"], o_hist["iter"])
    assert abs(g_sigma - o_sigma) <= 1e-12 * abs(o_sigma)
    np.testing.assert_allclose(g_hist["kkt"], o_hist["kkt"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(g_hist["pdGap"], o_hist["pdGap"], rtol=1e-6, atol=1e-14)
    errs = {f: _relerr(getattr(gvar, f), getattr(ovar, f)) for f in FIELDS}
    assert max(errs.values()) <= tol, errs
    assert abs(gvar.cScale - ovar.cScale) <= 1e-12 * ovar.cScale
    assert abs(gvar.dScale - ovar.dScale) <= 1e-12 * ovar.dScale
    assert gvar.name == 'Proximal ALM'
    assert list(gvar.time) == list(ovar.time)
    return errs


@pytest.mark.parametrize("n,nt,K", [(16, 8, 1), (16, 8, 2), (16, 8, 5), (32, 16, 60), (64, 32, 30), (33, 17, 25)])
def test_trajectory(n, nt, K):
    rho0, rho1 = get_example_2d("example1", n, n)
    _compare(rho0, rho1, nt, dict(tol=0.0), K)


@pytest.mark.parametrize("n,nt,K,nslabs", [(32, 16, 60, 2), (33, 49, 40, 3), (40, 36, 30, 4)])
def test_trajectory_on_time_slabs(n, nt, K, nslabs):
    """The one-pass dataflow on time slabs (round 4: the second gather's tails travel with the first's, q~ halo and u0 tail
    from q3) against the oracle, over KKT checks, sigma updates and rescale blocks; slabs with and without chunked passes."""
    rho0, rho1 = get_example_2d("example1", n, n)
    _compare(rho0, rho1, nt, dict(tol=0.0), K, nslabs=nslabs)


@pytest.mark.parametrize("n,nt,nslabs", [(33, 49, 3), (32, 16, 2)])
def test_one_pass_dataflow_equals_two_pass_on_time_slabs(n, nt, nslabs, monkeypatch):
    rho0, rho1 = get_example_2d("example1", n, n)
    res = []
    for fast in ("1", "0"):
        monkeypatch.setenv("DOTSOCP_PALM_FAST", fast)
        var, model = D.initialize(rho0, rho1, nt)
        oo = OD.default_opts(dict(tol=0.0, maxit=40), "PALM", False)
        D.InitialScaling(var, model, oo["scaling"], None, dim=2)
        hist, sigma = D.solver_socp_PALM(var, oo, model, nslabs=nslabs)
        res.append((var, hist, sigma))
    (a, ha, sa), (b, hb, sb) = res
    assert abs(sa - sb) <= 1e-13 * abs(sb)
    np.testing.assert_array_equal(ha["iter"], hb["iter"])
    errs = {f: _relerr(getattr(a, f), getattr(b, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-11, errs


def test_trajectory_rectangular_checkstep():
    rho0, rho1 = get_example_2d("example1", 24, 40)
    _compare(rho0, rho1, 12, dict(tol=0.0, scaling=False, sigma=0.1, ifCheckStepByStep=True), 12)


def test_free_running_solve():
    rho0, rho1 = get_example_2d("example1", 32, 32)
    ovar, omodel, o_hist, o_sigma = OD.solve_single_level(rho0, rho1, 16, dict(tol=1e-4), "PALM")
    out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, 16, 1, dict(tol=1e-4), "PALM")
    assert hist["iter"][-1] == o_hist["iter"][-1]
    assert np.max(hist["kkt"][-1][[0, 2, 5, 6]]) < 1e-4
    np.testing.assert_allclose(hist["kkt"][-1], o_hist["kkt"][-1], rtol=1e-6, atol=1e-14)
    rho_o, Ex_o, Ey_o = OD.recover_RhoE(ovar, omodel)
    np.testing.assert_allclose(out["rho"], rho_o, atol=1e-8)
    assert D.check_massConservation(out["rho"], 1e-2)
    assert hist["method"] == "PALM for DOT-SOCP"


@pytest.mark.parametrize("tsolve", ["tridiag", "dct"])
@pytest.mark.parametrize("n,nt", [(32, 16), (33, 49)])
@pytest.mark.parametrize("nslabs", [2, 3, 4])
def test_time_slabs_match_single_slab(n, nt, nslabs, tsolve, monkeypatch):
    """PALM in time-slab mode (all slabs on the one GPU of the test box): four neighbour exchanges per iteration
    (adjoint tails before each q-step, q~ halo + u0 tail after the first, q halo after the second) around the
    shared Poisson solve; 40 iterations incl. KKT blocks, sigma updates and the rescale blocks."""
    monkeypatch.setenv("DOTSOCP_TSOLVE", tsolve)
    rho0, rho1 = get_example_2d("example1", n, n)
    res = []
    for ns in (1, nslabs):
        var, model = D.initialize(rho0, rho1, nt)
        oo = OD.default_opts(dict(tol=0.0, maxit=40), "PALM", False)
        D.InitialScaling(var, model, oo["scaling"], None, dim=2)
        hist, sigma = D.solver_socp_PALM(var, oo, model, nslabs=ns)
        res.append((var, hist, sigma))
    (ref, h1, s1), (got, hn, sn) = res
    np.testing.assert_array_equal(hn["iter"], h1["iter"])
    np.testing.assert_allclose(hn["kkt"], h1["kkt"], rtol=1e-7, atol=1e-10)
    assert abs(sn - s1) <= 1e-12 * s1
    errs = {f: _relerr(getattr(got, f), getattr(ref, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-10, errs


def test_rejected_configurations():
    from oracle.examples import get_example_1d
    rho0, rho1 = get_example_1d("gaussian", 64)
    with pytest.raises(ValueError):
        D.solver_dotsocp1d(rho0, rho1, 16, 1, dict(tol=1e-3), "PALM")


@pytest.mark.parametrize("ny,nx,nt,K", [(32, 32, 16, 70), (40, 24, 9, 45), (65, 33, 17, 30), (130, 70, 20, 24)])
def test_one_pass_dataflow_equals_two_pass(ny, nx, nt, K, monkeypatch):
    """One slab: one pass over beta per iteration -- the first q-step's gather F*B*(z^k + beta^k) is formed from the second
    gather of the previous cone pass, F*B*((1 + tau) z^k + beta^{k-1}), minus tau F*B*(BF q^k + d) taken entry by entry
    (k_cone_fused modes 5 / 6, k_qstep_rhs VAR 3 with qk) -- against the two-pass dataflow (DOTSOCP_PALM_FAST=0).  The
    same algebra in another order of summation: 1e-11, over KKT checks, sigma updates and rescale blocks; tiles that are
    cut by the grid (130 x 70), odd sizes."""
    rho0, rho1 = get_example_2d("example1", ny, nx)
    res = []
    for fast in ("1", "0"):
        monkeypatch.setenv("DOTSOCP_PALM_FAST", fast)
        var, model = D.initialize(rho0, rho1, nt)
        oo = OD.default_opts(dict(tol=0.0, maxit=K), "PALM", False)
        D.InitialScaling(var, model, oo["scaling"], None, dim=2)
        hist, sigma = D.solver_socp_PALM(var, oo, model)
        res.append((var, hist, sigma))
    (a, ha, sa), (b, hb, sb) = res
    assert abs(sa - sb) <= 1e-13 * abs(sb)
    np.testing.assert_array_equal(ha["iter"], hb["iter"])
    np.testing.assert_allclose(ha["kkt"], hb["kkt"], rtol=1e-8, atol=1e-12)
    errs = {f: _relerr(getattr(a, f), getattr(b, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-11, errs
