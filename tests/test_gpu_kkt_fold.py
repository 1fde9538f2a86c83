"""The folded KKT path (one slab, fused dataflow): on an iteration that ends with a KKT check the q-step kernel
accumulates every sum that needs only phi, q^+, alpha^+, A phi and c (k_qstep_rhs<., 0, true>), the cell pass adds the
F*B*beta terms of the edges inside its tiles (k_kkt_cells<., true>), a small launch finishes the edges on tile borders
(k_kkt_bnd), and a sigma update corrects the stored right-hand side instead of recomputing it (k_rhs_sigma_fix, alpha
divided on load by the next q-step).  Checked against the unfolded path of the same library (DOTSOCP_KKT_FOLD=0:
separate node / edge launches, eager scalings, new rhs pass) -- both are also compared with the oracle by the
trajectory tests -- on shapes with partial tiles, several chunks per tile, weights, 1-D, step-by-step checks."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle.examples import (ensure_barrier_validity, gene_barrier_of_circle_pillar, get_example_1d,
                             get_example_2d, get_weight_by_barrier)

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")


def _run(rho0, rho1, nt, opts, weight=None):
    dim = 2 if np.ndim(rho0) == 2 else 1
    var, model = D.initialize(rho0, rho1, nt)
    if weight is not None:
        model.weight = np.asarray(weight, dtype=np.float64)
    o = OD.default_opts(opts, "inPALM", weight is not None)
    D.InitialScaling(var, model, o["scaling"], None, dim=dim, weighted=weight is not None)
    solve = D.solver_wsocp_inPALM if weight is not None else D.solver_socp_inPALM
    hist, sigma = solve(var, o, model)
    return var, hist, sigma


def _both(monkeypatch, rho0, rho1, nt, opts, weight=None, tol=1e-12):
    monkeypatch.setenv("DOTSOCP_KKT_FOLD", "0")
    ref, h0, s0 = _run(rho0, rho1, nt, opts, weight)
    monkeypatch.setenv("DOTSOCP_KKT_FOLD", "1")
    got, h1, s1 = _run(rho0, rho1, nt, opts, weight)
    np.testing.assert_array_equal(h1["iter"], h0["iter"])
    assert abs(s1 - s0) <= 1e-13 * abs(s0)
    # columns: every residual but the exact-cancellation one (5) to 1e-9 relative; that one is rounding noise
    np.testing.assert_allclose(h1["kkt"], h0["kkt"], rtol=1e-9, atol=1e-10 if weight is None else 1e-9)
    np.testing.assert_allclose(h1["pdGap"], h0["pdGap"], rtol=1e-9, atol=1e-15)
    for f in FIELDS:
        a, b = getattr(got, f), getattr(ref, f)
        err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        assert err <= tol, (f, err)


@pytest.mark.parametrize("ny,nx,nt,K", [(32, 32, 16, 40), (100, 70, 20, 30), (65, 129, 33, 25), (63, 5, 7, 20), (5, 3, 4, 12),
                                         (2, 2, 2, 6), (129, 3, 5, 15), (256, 256, 64, 30)])
def test_folded_equals_unfolded(ny, nx, nt, K, monkeypatch):
    if ny * nx <= 6:
        rho0 = np.ones((ny, nx))
        rho1 = np.ones((ny, nx))
        rho1.flat[0] = 1.5
        rho1 /= rho1.mean()
    else:
        rho0, rho1 = get_example_2d("example1", ny, nx)
    _both(monkeypatch, rho0, rho1, nt, dict(tol=0.0, maxit=K))


def test_folded_step_by_step_alg2_and_no_scaling(monkeypatch):
    rho0, rho1 = get_example_2d("example1", 24, 40)
    _both(monkeypatch, rho0, rho1, 12, dict(tol=0.0, maxit=15, ifCheckStepByStep=True))
    _both(monkeypatch, rho0, rho1, 12, dict(tol=0.0, maxit=20, scaling=False, sigma=0.1))


def test_folded_weighted_and_1d(monkeypatch):
    rho0, rho1 = get_example_2d("example1", 33, 47)
    barrier = gene_barrier_of_circle_pillar()
    weight = get_weight_by_barrier(47, 33, 13, barrier)
    rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    _both(monkeypatch, rho0, rho1, 13, dict(tol=0.0, maxit=25), weight=weight, tol=1e-9)
    r0, r1 = get_example_1d("gaussian", 129)
    _both(monkeypatch, r0, r1, 33, dict(tol=0.0, maxit=60))


def test_folded_free_running_stops_like_unfolded(monkeypatch):
    rho0, rho1 = get_example_2d("example1", 32, 32)
    monkeypatch.setenv("DOTSOCP_KKT_FOLD", "0")
    ref, h0, s0 = _run(rho0, rho1, 16, dict(tol=1e-4))
    monkeypatch.setenv("DOTSOCP_KKT_FOLD", "1")
    got, h1, s1 = _run(rho0, rho1, 16, dict(tol=1e-4))
    assert h1["iter"][-1] == h0["iter"][-1] and h1["len"] == h0["len"]
    np.testing.assert_allclose(h1["kkt"], h0["kkt"], rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("ny,nx,nt,K,sigma0", [(32, 32, 16, 230, 1.0), (33, 17, 9, 130, 1.0), (16, 16, 8, 320, 0.01)])
def test_rescale_checks_every_100_iterations(ny, nx, nt, K, sigma0, monkeypatch):
    """Past the second rescale the loop checks the norm ratio every 100 iterations (solver_socp_inPALM.m:139-149).  No KKT
    check precedes those iterations, so the five norms come from the light pass (cell kernel in norms-only mode + three
    sums of squares, nothing materialised) -- against the oracle, and against the path that materialises z and runs
    the full KKT sums (DOTSOCP_NORM_CACHE=0).  On the small grids (the coarse levels of a multilevel run) the check does
    fire a rescale: the pending multiplier step then has to be executed under the OLD scaling first.  The sigma0 = 0.01
    case is one where a check past iteration 100 does rescale (the oracle ends with rescale = 4)."""
    from oracle.inpalm import InPALMState
    rho0, rho1 = get_example_2d("example1", ny, nx)
    opts = dict(tol=0.0, maxit=K, sigma=sigma0)
    ovar, omodel, oo = OD.make_level(rho0, rho1, nt, opts, "inPALM", None)
    st = InPALMState(ovar, oo, omodel)
    st.run()
    assert st.rescale == (4 if sigma0 != 1.0 else 3)
    o_hist, o_sigma = st.finish()
    monkeypatch.setenv("DOTSOCP_NORM_CACHE", "1")
    got, h1, s1 = _run(rho0, rho1, nt, opts)
    np.testing.assert_array_equal(h1["iter"], o_hist["iter"])
    assert abs(s1 - o_sigma) <= 1e-12 * abs(o_sigma)
    assert abs(got.cScale - ovar.cScale) <= 1e-12 * ovar.cScale and abs(got.dScale - ovar.dScale) <= 1e-12 * ovar.dScale
    for f in FIELDS:
        a, b = getattr(got, f), getattr(ovar, f)
        assert np.max(np.abs(a - b)) / np.max(np.abs(b)) <= 1e-9, f
    monkeypatch.setenv("DOTSOCP_NORM_CACHE", "0")
    ref, h0, s0 = _run(rho0, rho1, nt, opts)
    assert abs(s1 - s0) <= 1e-13 * abs(s0)
    for f in FIELDS:
        a, b = getattr(got, f), getattr(ref, f)
        assert np.max(np.abs(a - b)) / np.max(np.abs(b)) <= 1e-11, f


@pytest.mark.parametrize("ny,nx,nt,K", [(100, 70, 20, 30), (65, 129, 33, 25), (63, 5, 7, 20), (64, 8, 9, 12), (130, 9, 5, 15),
                                         (256, 256, 64, 30)])
def test_wide_qstep_tile_changes_nothing(ny, nx, nt, K, monkeypatch):
    """The plain inPALM q-step runs on 64 x 8 tiles on large grids and on 64 x 4 tiles otherwise (DOTSOCP_QTX forces
    either): every entry goes through the same arithmetic in both, so the trajectories must agree to the last bit."""
    rho0, rho1 = get_example_2d("example1", ny, nx)
    monkeypatch.setenv("DOTSOCP_QTX", "4")
    ref, h0, s0 = _run(rho0, rho1, nt, dict(tol=0.0, maxit=K))
    monkeypatch.setenv("DOTSOCP_QTX", "8")
    got, h1, s1 = _run(rho0, rho1, nt, dict(tol=0.0, maxit=K))
    assert s1 == s0
    np.testing.assert_array_equal(h1["kkt"], h0["kkt"])
    for f in FIELDS:
        np.testing.assert_array_equal(getattr(got, f), getattr(ref, f))


@pytest.mark.parametrize("ny,nx,nt,K,ngpu", [(32, 32, 16, 40, None), (65, 129, 17, 25, None), (100, 70, 20, 30, None),
                                              (64, 48, 24, 30, 3), (33, 33, 17, 25, 2)])
def test_palm_folded_equals_unfolded(ny, nx, nt, K, ngpu, monkeypatch):
    """PALM (solver_socp_PALM.m:231-232 calls the same KKT block): on an iteration that ends with a check its second
    q-step runs in the KKT flavour of k_qstep_rhs and the block takes the folded cell pass -- against the unfolded block
    of the same library (DOTSOCP_KKT_FOLD=0), on one slab (pitched and unpitched rows) and on in-process time slabs."""
    rho0, rho1 = get_example_2d("example1", ny, nx)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DOTSOCP_KKT_FOLD", flag)
        var, model = D.initialize(rho0, rho1, nt)
        o = OD.default_opts(dict(tol=0.0, maxit=K), "PALM", False)
        D.InitialScaling(var, model, o["scaling"], None, dim=2)
        ctx = D.InPALMContext(var, o, model, method="PALM", ngpu=ngpu)
        ctx.run(-1)
        hist, sigma = ctx.finish(download=True)
        ctx.close()
        res[flag] = (var, hist, sigma)
    (ref, h0, s0), (got, h1, s1) = res["0"], res["1"]
    np.testing.assert_array_equal(h1["iter"], h0["iter"])
    assert len(h0["iter"]) >= 4 and abs(s1 - s0) <= 1e-13 * abs(s0)
    np.testing.assert_allclose(h1["kkt"], h0["kkt"], rtol=1e-9, atol=1e-10)
    for f in FIELDS:
        a, b = getattr(got, f), getattr(ref, f)
        err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
        assert err <= 1e-12, (f, err)


@pytest.mark.parametrize("method", ["inPALM", "PALM", "acc-ADMM"])
def test_nontemporal_streams_change_nothing(method, monkeypatch):
    """The cone multipliers travel with the non-temporal cache policy (device_utils.h: ld_stream / st_stream): a hint to
    the caches, no arithmetic -- every iterate and the KKT history are bit-identical with DOTSOCP_NT=0."""
    rho0, rho1 = get_example_2d("example1", 70, 40)
    res = []
    for nt_on in ("1", "0"):
        monkeypatch.setenv("DOTSOCP_NT", nt_on)
        var, model = D.initialize(rho0, rho1, 17)
        D.InitialScaling(var, model, True, None, dim=2)
        o = OD.default_opts(dict(tol=0.0, maxit=40), method, False)
        ctx = D.InPALMContext(var, o, model, method=method)
        ctx.run(40)
        hist, sigma = ctx.finish(download=True)
        ctx.close()
        res.append((var, hist, sigma))
    (a, ha, sa), (b, hb, sb) = res
    assert sa == sb
    np.testing.assert_array_equal(ha["kkt"], hb["kkt"])
    for f in ("phi", "q", "z", "alpha", "beta"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)


@pytest.mark.parametrize("ny,nx,nt,method", [(512, 64, 16, "inPALM"), (512, 40, 9, "inPALM"), (768, 32, 8, "inPALM"),
                                              (512, 64, 16, "PALM"), (512, 64, 16, "acc-ADMM")])
def test_longer_rows_for_power_of_two_heights_change_nothing(ny, nx, nt, method, monkeypatch):
    """Single-slab contexts store rows of 512, 768, 1024, ... doubles 16 doubles longer (Solver::row_pitch: rows a multiple of
    2 KB apart keep the x lines of the Poisson solve on the same DRAM banks).  Only addresses change: the same lines share
    a complex transform, every kernel sums in the same order -- bit-identical with DOTSOCP_PITCH2=0.  Covers the pipelined
    DCT kernels on pitched rows (y: line distance as an argument, x / t: LineMap with es != nin), the generic ones for
    the short axes, PALM's and acc-ADMM's kernels."""
    rho0, rho1 = get_example_2d("example1", ny, nx)
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("DOTSOCP_PITCH2", on)
        var, model = D.initialize(rho0, rho1, nt)
        D.InitialScaling(var, model, True, None, dim=2)
        o = OD.default_opts(dict(tol=0.0, maxit=12), method, False)
        ctx = D.InPALMContext(var, o, model, method=method)
        ctx.run(12)
        hist, sigma = ctx.finish(download=True)
        ctx.close()
        res.append((var, hist, sigma))
    (a, ha, sa), (b, hb, sb) = res
    assert sa == sb
    np.testing.assert_array_equal(ha["kkt"], hb["kkt"])
    for f in ("phi", "q", "z", "alpha", "beta"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)
