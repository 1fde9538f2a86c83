"""the accelerated ADMM loop (solver_socp_accADMM.m / solver_wsocp_accADMM.m)
on the device against its CPU restatement (oracle/accadmm.py, PARITY UNPINNED like the inPALM oracle:
the reference holds no fixture for it).  Same bar as the inPALM loop: every state array <= 1e-9 relative
after K iterations with the live sigma / rescale / restart schedule, identical KKT history."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle.accadmm import AccADMMState
from oracle.examples import (ensure_barrier_validity, gene_barrier_of_circle_pillar, get_example_2d,
                             get_weight_by_barrier)

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")


def _relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _compare(rho0, rho1, nt, opts, K, weight=None, tol=1e-9):
    opts = dict(opts, maxit=K)
    weighted = weight is not None
    ovar, omodel, oo = OD.make_level(rho0, rho1, nt, opts, "acc-ADMM", weight)
    st = AccADMMState(ovar, oo, omodel, weighted=weighted)
    st.run()
    o_hist, o_sigma = st.finish()
    gvar, gmodel = D.initialize(rho0, rho1, nt)
    if weighted:
        gmodel.weight = np.asarray(weight, dtype=np.float64)
    D.InitialScaling(gvar, gmodel, oo["scaling"], None, dim=2, weighted=weighted)
    solve = D.solver_wsocp_accADMM if weighted else D.solver_socp_accADMM
    g_hist, g_sigma = solve(gvar, oo, gmodel)
    assert g_hist["len"] == o_hist["len"]
    np.testing.assert_array_equal(g_hist["iter"], o_hist["iter"])
    assert abs(g_sigma - o_sigma) <= 1e-12 * abs(o_sigma)
    np.testing.assert_allclose(g_hist["kkt"], o_hist["kkt"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(g_hist["pdGap"], o_hist["pdGap"], rtol=1e-6, atol=1e-14)
    errs = {f: _relerr(getattr(gvar, f), getattr(ovar, f)) for f in FIELDS}
    assert max(errs.values()) <= tol, errs
    assert abs(gvar.cScale - ovar.cScale) <= 1e-12 * ovar.cScale
    assert abs(gvar.dScale - ovar.dScale) <= 1e-12 * ovar.dScale
    assert gvar.name == 'Accelerated ADMM'
    assert list(gvar.time) == list(ovar.time)
    return errs


@pytest.mark.parametrize("n,nt,K", [(16, 8, 1), (16, 8, 2), (16, 8, 7), (32, 16, 60), (64, 32, 30), (33, 17, 25)])
def test_trajectory_halpern(n, nt, K):
    rho0, rho1 = get_example_2d("example1", n, n)
    _compare(rho0, rho1, nt, dict(tol=0.0), K)


def test_trajectory_halpern_short_restart():
    """restart = 5: anchors are re-set every five extrapolations, also between KKT checks."""
    rho0, rho1 = get_example_2d("example1", 24, 40)
    _compare(rho0, rho1, 12, dict(tol=0.0, restart=5, rho=1.7), 45)


@pytest.mark.parametrize("theta,K", [(3.0, 30), (5.0, 12)])
def test_trajectory_nesterov_type(theta, K):
    """theta != 2: the non-Halpern branch (solver_socp_accADMM.m:389-422) with its HatOld carry."""
    rho0, rho1 = get_example_2d("example1", 32, 32)
    _compare(rho0, rho1, 16, dict(tol=0.0, theta=theta, restart=8), K)


def test_trajectory_checkstep_no_scaling():
    rho0, rho1 = get_example_2d("example1", 16, 16)
    _compare(rho0, rho1, 8, dict(tol=0.0, scaling=False, sigma=0.1, ifCheckStepByStep=True), 12)


@pytest.mark.parametrize("n,nt,K", [(32, 16, 40), (33, 9, 20)])
def test_trajectory_weighted(n, nt, K):
    rho0, rho1 = get_example_2d("example1", n, n)
    barrier = gene_barrier_of_circle_pillar()
    weight = get_weight_by_barrier(n, n, nt, barrier)
    rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    _compare(rho0, rho1, nt, dict(tol=0.0), K, weight=weight, tol=1e-8)


def test_free_running_solve():
    """Full solve through the driver with method = "acc-ADMM": same stop iteration as the oracle, KKT < tol,
    mass conservation, same transport as the oracle."""
    rho0, rho1 = get_example_2d("example1", 32, 32)
    ovar, omodel, o_hist, o_sigma = OD.solve_single_level(rho0, rho1, 16, dict(tol=1e-4), "acc-ADMM")
    out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, 16, 1, dict(tol=1e-4), "acc-ADMM")
    assert hist["iter"][-1] == o_hist["iter"][-1]
    assert np.max(hist["kkt"][-1][[0, 2, 5, 6]]) < 1e-4
    np.testing.assert_allclose(hist["kkt"][-1], o_hist["kkt"][-1], rtol=1e-6, atol=1e-14)
    rho_o, Ex_o, Ey_o = OD.recover_RhoE(ovar, omodel)
    np.testing.assert_allclose(out["rho"], rho_o, atol=1e-8)
    assert D.check_massConservation(out["rho"], 1e-2)
    assert hist["method"] == "acc-ADMM for DOT-SOCP"


@pytest.mark.parametrize("tsolve", ["tridiag", "dct"])
@pytest.mark.parametrize("case", ["halpern_32x16", "halpern_33x49", "theta3_32x16", "weighted_32x48", "checkstep_24x12"])
@pytest.mark.parametrize("nslabs", [2, 3, 4])
def test_time_slabs_match_single_slab(case, nslabs, tsolve, monkeypatch):
    """acc-ADMM in time-slab mode (all slabs on the one GPU of the test box): adjoint tails + phi head in front of the
    q-step, raw q^+ halo + u0 tail behind it, phi^+ head in front of a KKT block; folded (Halpern, no KKT check) and
    unfolded iterations, sigma updates, rescale blocks and restarts included."""
    monkeypatch.setenv("DOTSOCP_TSOLVE", tsolve)
    kind, dims = case.split("_")
    n, nt = [int(v) for v in dims.split("x")]
    rho0, rho1 = get_example_2d("example1", n, n)
    weight = None
    opts = dict(tol=0.0, maxit=90)
    if kind == "theta3":
        opts["theta"] = 3.0
    if kind == "checkstep":
        opts.update(ifCheckStepByStep=True, maxit=25)
    if kind == "weighted":
        barrier = gene_barrier_of_circle_pillar()
        weight = get_weight_by_barrier(n, n, nt, barrier)
        rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    res = []
    for ns in (1, nslabs):
        var, model = D.initialize(rho0, rho1, nt)
        if weight is not None:
            model.weight = np.asarray(weight, dtype=np.float64)
        oo = OD.default_opts(opts, "acc-ADMM", weight is not None)
        D.InitialScaling(var, model, oo["scaling"], None, dim=2, weighted=weight is not None)
        solve = D.solver_wsocp_accADMM if weight is not None else D.solver_socp_accADMM
        hist, sigma = solve(var, oo, model, nslabs=ns)
        res.append((var, hist, sigma))
    (ref, h1, s1), (got, hn, sn) = res
    np.testing.assert_array_equal(hn["iter"], h1["iter"])
    np.testing.assert_allclose(hn["kkt"], h1["kkt"], rtol=1e-7, atol=1e-10)
    assert abs(sn - s1) <= 1e-12 * s1
    errs = {f: _relerr(getattr(got, f), getattr(ref, f)) for f in FIELDS}
    assert max(errs.values()) <= (1e-8 if weight is not None else 1e-10), errs


@pytest.mark.parametrize("case", ["halpern", "restart5", "weighted", "slabs3"])
def test_post_kkt_cone_pass_changes_nothing(case, monkeypatch):
    """Halpern iterations that end with a KKT check: z and beta are extrapolated by ONE cone pass (k_acc_cone modes 1 / 3:
    x^+ recomputed from the untouched state, the sigma factor applied on the way, the anchors stored, the next gather
    emitted) instead of two scalings, two anchor copies, two extrapolation passes and a gather pass
    (DOTSOCP_ACC_POST=0).  Same arithmetic in the same order: bit-identical."""
    n, nt, K = 32, 16, 70
    rho0, rho1 = get_example_2d("example1", n, n)
    opts, weight, ns = dict(tol=0.0, maxit=K), None, 1
    if case == "restart5":
        opts.update(restart=5, rho=1.7)
    if case == "weighted":
        barrier = gene_barrier_of_circle_pillar()
        weight = get_weight_by_barrier(n, n, nt, barrier)
        rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    if case == "slabs3":
        ns = 3
    res = []
    for post in ("1", "0"):
        monkeypatch.setenv("DOTSOCP_ACC_POST", post)
        var, model = D.initialize(rho0, rho1, nt)
        if weight is not None:
            model.weight = np.asarray(weight, dtype=np.float64)
        oo = OD.default_opts(opts, "acc-ADMM", weight is not None)
        D.InitialScaling(var, model, oo["scaling"], None, dim=2, weighted=weight is not None)
        solve = D.solver_wsocp_accADMM if weight is not None else D.solver_socp_accADMM
        hist, sigma = solve(var, oo, model, nslabs=ns)
        res.append((var, hist, sigma))
    (a, ha, sa), (b, hb, sb) = res
    assert sa == sb
    np.testing.assert_array_equal(ha["kkt"], hb["kkt"])
    for f in FIELDS:
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)


@pytest.mark.parametrize("case", ["halpern", "theta3", "weighted", "checkstep", "cut_tiles"])
def test_folded_kkt_equals_unfolded(case, monkeypatch):
    """One slab: on an iteration that ends with a KKT check the cone pass runs after the phi-step and takes the cell sums
    and the F*B*beta^+ terms of every entry while z^+, beta^+ are in registers (k_acc_cone<RAW, ., KKT>, k_kkt_bnd for the
    tile borders), and what is left -- the sums made of phi^+, q^+, alpha^+, c -- comes from k_kkt's lean parts without a
    read of beta.  Against the unfolded block (DOTSOCP_KKT_FOLD=0: four launches that read z^+, beta^+ from memory):
    the same sums in another order of summation -- KKT histories to 1e-9 relative, iterates to 1e-11."""
    ny, nx, nt, K = 32, 32, 16, 70
    opts, weight = dict(tol=0.0, maxit=K), None
    if case == "theta3":
        opts.update(theta=3.0, restart=8, maxit=40)
    if case == "checkstep":
        opts.update(ifCheckStepByStep=True, maxit=20)
    if case == "cut_tiles":
        ny, nx, nt = 70, 41, 11
    rho0, rho1 = get_example_2d("example1", ny, nx)
    if case == "weighted":
        barrier = gene_barrier_of_circle_pillar()
        weight = get_weight_by_barrier(ny, nx, nt, barrier)
        rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    res = []
    for fold in ("1", "0"):
        monkeypatch.setenv("DOTSOCP_KKT_FOLD", fold)
        var, model = D.initialize(rho0, rho1, nt)
        if weight is not None:
            model.weight = np.asarray(weight, dtype=np.float64)
        oo = OD.default_opts(opts, "acc-ADMM", weight is not None)
        D.InitialScaling(var, model, oo["scaling"], None, dim=2, weighted=weight is not None)
        solve = D.solver_wsocp_accADMM if weight is not None else D.solver_socp_accADMM
        hist, sigma = solve(var, oo, model)
        res.append((var, hist, sigma))
    (a, ha, sa), (b, hb, sb) = res
    assert abs(sa - sb) <= 1e-13 * abs(sb)
    np.testing.assert_array_equal(ha["iter"], hb["iter"])
    np.testing.assert_allclose(ha["kkt"], hb["kkt"], rtol=1e-9, atol=1e-14)
    errs = {f: _relerr(getattr(a, f), getattr(b, f)) for f in FIELDS}
    assert max(errs.values()) <= (1e-9 if weight is not None else 1e-11), errs
