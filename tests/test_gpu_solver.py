"""Solver-level parity (boundary B1): the device-resident inPALM loop against the CPU oracle on
identical rho0/rho1 inputs -- fixed-length trajectories (every iterate compared) and free-running
solves (same stop iteration, KKT history, mass conservation).  Tolerance: the loop is fp64 and
differs from the oracle only in the summation order of the DCT and of the global norms, so
trajectories agree to ~1e-12 relative after tens of iterations; the bar asserted here is 1e-9
(SURVEY.md section 8d "parity gate")."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle.examples import (ensure_barrier_validity, gene_barrier_of_circle_pillar, get_example_1d,
                             get_example_2d, get_weight_by_barrier)
from oracle.inpalm import InPALMState

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")


@pytest.fixture(params=["fused", "unfused"], autouse=True)
def _dataflow(request, monkeypatch):
    """Every solver test runs on both device dataflows: the fused cone kernel (deferred multiplier
    update, z never stored -- the production path) and the unfused one that stores z and mirrors the
    reference's three cone passes (DOTSOCP_FUSED=0)."""
    monkeypatch.setenv("DOTSOCP_FUSED", "1" if request.param == "fused" else "0")


def _gpu_level(rho0, rho1, nt, opts, method="inPALM", weight=None):
    dim = 2 if np.ndim(rho0) == 2 else 1
    var, model = D.initialize(rho0, rho1, nt)
    if weight is not None:
        model.weight = np.asarray(weight, dtype=np.float64)
    o = OD.default_opts(opts, method, weight is not None)
    D.InitialScaling(var, model, o["scaling"], None, dim=dim, weighted=weight is not None)
    return var, model, o


def _relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _compare_run(rho0, rho1, nt, opts, K, weight=None, method="inPALM", tol=1e-9):
    opts = dict(opts, maxit=K)
    ovar, omodel, oo = OD.make_level(rho0, rho1, nt, opts, method, weight)
    st = InPALMState(ovar, oo, omodel, weighted=weight is not None)
    st.run()
    o_hist, o_sigma = st.finish()
    gvar, gmodel, go = _gpu_level(rho0, rho1, nt, opts, method, weight)
    # host-side set-up must agree exactly (same formulas)
    assert gvar.D == ovar.D and gvar.E == ovar.E
    solve = D.solver_wsocp_inPALM if weight is not None else D.solver_socp_inPALM
    g_hist, g_sigma = solve(gvar, go, gmodel)
    assert g_hist["len"] == o_hist["len"]
    np.testing.assert_array_equal(g_hist["iter"], o_hist["iter"])
    assert abs(g_sigma - o_sigma) <= 1e-12 * abs(o_sigma)
    # column 5 (||F*B*beta + D_w alpha||) is an exact-cancellation residual: pure rounding noise
    # (1e-16 unweighted, 1e-11 with 1e6 barrier weights), hence the absolute floor
    np.testing.assert_allclose(g_hist["kkt"], o_hist["kkt"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(g_hist["pdGap"], o_hist["pdGap"], rtol=1e-6, atol=1e-14)
    errs = {f: _relerr(getattr(gvar, f), getattr(ovar, f)) for f in FIELDS}
    assert max(errs.values()) <= tol, errs
    assert abs(gvar.cScale - ovar.cScale) <= 1e-12 * ovar.cScale
    assert abs(gvar.dScale - ovar.dScale) <= 1e-12 * ovar.dScale
    return errs


@pytest.mark.parametrize("n,nt,K", [(16, 8, 1), (16, 8, 5), (32, 16, 60), (64, 32, 30), (33, 17, 25)])
def test_trajectory_dot2d(n, nt, K):
    rho0, rho1 = get_example_2d("example1", n, n)
    _compare_run(rho0, rho1, nt, dict(tol=0.0), K)


@pytest.mark.parametrize("n,nt,K", [(256, 64, 50), (257, 65, 50)])
def test_parity_gate_config2(n, nt, K, request):
    """SURVEY.md section 8d parity gate: BASELINE config 2 (256x256x64) and its multilevel-compatible
    2^k+1 twin, K iterations with the live sigma / rescale schedule, all five state arrays <= 1e-9.
    (The oracle needs about a second per iteration at this size, hence one dataflow only.)"""
    if "unfused" in request.node.name:
        pytest.skip("production dataflow only at this size")
    rho0, rho1 = get_example_2d("example1", n, n)
    _compare_run(rho0, rho1, nt, dict(tol=0.0), K)


def test_parity_gate_config5_wdot2d(request):
    """BASELINE configs[4] at its full size: wdot2d 512 x 512 x 128 with the circle-pillar obstacle (weight 1e6 on the
    barrier edges), K iterations against the oracle (about ten seconds per oracle iteration, hence K = 4 and the
    production dataflow only); tolerance as in the small weighted cases."""
    if "unfused" in request.node.name:
        pytest.skip("production dataflow only at this size")
    n, nt, K = 512, 128, 4
    rho0, rho1 = get_example_2d("example1", n, n)
    barrier = gene_barrier_of_circle_pillar()
    weight = get_weight_by_barrier(n, n, nt, barrier)
    rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    _compare_run(rho0, rho1, nt, dict(tol=0.0), K, weight=weight, tol=1e-8)


@pytest.mark.parametrize("ny,nx,nt,K", [(100, 70, 20, 15), (50, 130, 9, 12), (65, 129, 33, 10)])
def test_trajectory_dot2d_odd_shapes(ny, nx, nt, K):
    """Lengths that are neither powers of two nor multiples of the tile sizes: partial tiles in y and x, the DCT as a
    matrix product on the matrix cores (even lengths 100 / 70 / 50 / 130, odd 65 / 129) or the scalar dense kernel
    (n < 48)."""
    rho0, rho1 = get_example_2d("example1", ny, nx)
    assert rho0.shape == (ny, nx)
    _compare_run(rho0, rho1, nt, dict(tol=0.0), K)


def test_trajectory_dot2d_rectangular_alg2():
    rho0, rho1 = get_example_2d("example1", 24, 40)     # generator returns (nx, ny) arrays: ny = 24, nx = 40
    _compare_run(rho0, rho1, 12, dict(tol=0.0), 20, method="ALG2")


def test_trajectory_no_scaling_checkstep():
    rho0, rho1 = get_example_2d("example1", 16, 16)
    _compare_run(rho0, rho1, 8, dict(tol=0.0, scaling=False, sigma=0.1, ifCheckStepByStep=True), 12)


@pytest.mark.parametrize("nx,nt,K", [(64, 16, 40), (129, 33, 40), (128, 32, 80)])
def test_trajectory_dot1d(nx, nt, K):
    rho0, rho1 = get_example_1d("gaussian", nx)
    _compare_run(rho0, rho1, nt, dict(tol=0.0), K)


@pytest.mark.parametrize("n,nt,K", [(32, 16, 40), (33, 9, 20)])
def test_trajectory_wdot2d(n, nt, K):
    rho0, rho1 = get_example_2d("example1", n, n)
    barrier = gene_barrier_of_circle_pillar()
    weight = get_weight_by_barrier(n, n, nt, barrier)
    rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    assert (weight == 1e6).sum() > 0
    _compare_run(rho0, rho1, nt, dict(tol=0.0), K, weight=weight, tol=1e-8)


@pytest.mark.parametrize("case,K,nslabs", [("dot2d", 1, 1), ("dot2d", 3, 1), ("dot1d", 1, 1), ("dot1d", 3, 1), ("wdot2d", 2, 1),
                                           ("dot2d", 3, 2), ("dot2d", 3, 3), ("dot1d", 3, 2), ("wdot2d", 2, 3)])
def test_iterations_from_a_random_state(case, K, nslabs, request):
    """The steps of ONE iteration in isolation from any trajectory: phi, q, alpha and beta are
    random -- every term of the q-step's right-hand side, of its diagonal, of the multiplier updates and of the KKT sums
    is generic, none vanishes as in the all-zero start of initialize.m -- and one to three iterations with a KKT check in
    each are compared with the oracle field by field (z is random too: solver_socp_inPALM.m:199 overwrites it before its
    first use, so it must not matter).  nslabs > 1: the same on time slabs, whose very first halo exchanges then carry generic layers.
    beta is zero where the cone row has no edge (the slots mexBFd leaves unwritten at the domain boundary): the reference
    never makes them non-zero, and compute_kkt_dot_complement reads what the projection of :240 left in those slots of
    the shared temporary z2 (:242 does not overwrite them) -- stale data the device, which builds z2 in registers, does
    not reproduce."""
    if nslabs > 1 and "unfused" in request.node.name:
        pytest.skip("time slabs run the fused dataflow")
    rng = np.random.default_rng(11)
    weight = None
    if case == "dot1d":
        rho0, rho1 = get_example_1d("gaussian", 48)
        nt = 10
    else:
        rho0, rho1 = get_example_2d("example1", 20, 28)
        nt = 9
        if case == "wdot2d":
            barrier = gene_barrier_of_circle_pillar()
            weight = get_weight_by_barrier(20, 28, nt, barrier)
            rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    opts = dict(tol=0.0, maxit=K, ifCheckStepByStep=True, sigma=0.7)
    ovar, omodel, oo = OD.make_level(rho0, rho1, nt, opts, "inPALM", weight)
    gvar, gmodel, go = _gpu_level(rho0, rho1, nt, opts, "inPALM", weight)
    start = {"phi": rng.standard_normal(ovar.phi.shape), "q": 0.3 * rng.standard_normal(ovar.q.shape),
             "alpha": 0.5 * rng.standard_normal(ovar.alpha.shape),
             "z": np.asfortranarray(0.6 * rng.standard_normal(ovar.z.shape)),      # read by nobody (:199 overwrites it first)
             "beta": np.asfortranarray(0.4 * rng.standard_normal(ovar.beta.shape))}
    from oracle import mexops
    probe = np.full(ovar.beta.shape, np.nan, order="F")
    if case == "dot1d":
        mexops.mexBFd1d(probe, start["q"], nt, rho0.size)
    else:
        mexops.mexBFd(probe, start["q"], nt, rho0.shape[1], rho0.shape[0])
    assert 0 < np.isnan(probe).sum() < probe.size // 4
    start["beta"][np.isnan(probe)] = 0.0
    start["z"][np.isnan(probe)] = 0.0
    for v in (ovar, gvar):
        for f, a in start.items():
            setattr(v, f, a.copy(order="F"))
    st = InPALMState(ovar, oo, omodel, weighted=weight is not None)
    st.run()
    o_hist, o_sigma = st.finish()
    solve = D.solver_wsocp_inPALM if weight is not None else D.solver_socp_inPALM
    g_hist, g_sigma = solve(gvar, go, gmodel, nslabs=nslabs)
    assert g_hist["len"] == o_hist["len"] == K
    assert abs(g_sigma - o_sigma) <= 1e-12 * abs(o_sigma)
    np.testing.assert_allclose(g_hist["kkt"], o_hist["kkt"], rtol=1e-8, atol=1e-12)
    errs = {f: _relerr(getattr(gvar, f), getattr(ovar, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-11, errs
    assert _relerr(gvar.q, start["q"]) > 1e-2 and _relerr(gvar.beta, start["beta"]) > 1e-2     # the state did move


def test_free_running_solve_dot2d():
    """Full solve to tolerance through the driver: same stop iteration as the oracle, KKT < tol,
    per-layer mass conservation (solver_dotsocp2d.m:283-287)."""
    rho0, rho1 = get_example_2d("example1", 32, 32)
    ovar, omodel, o_hist, o_sigma = OD.solve_single_level(rho0, rho1, 16, dict(tol=1e-4))
    out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, 16, 1, dict(tol=1e-4), "inPALM")
    assert hist["iter"][-1] == o_hist["iter"][-1]
    assert np.max(hist["kkt"][-1][[0, 2, 5, 6]]) < 1e-4
    np.testing.assert_allclose(hist["kkt"][-1], o_hist["kkt"][-1], rtol=1e-6, atol=1e-14)
    rho_o, Ex_o, Ey_o = OD.recover_RhoE(ovar, omodel)
    np.testing.assert_allclose(out["rho"], rho_o, atol=1e-8)
    assert D.check_massConservation(out["rho"], 1e-2)
    assert timeML[0]["Iters"] == hist["iter"][-1]


def test_free_running_solve_dot1d():
    rho0, rho1 = get_example_1d("gaussian", 129)
    ovar, omodel, o_hist, o_sigma = OD.solve_single_level(rho0, rho1, 33, dict(tol=1e-4))
    out, timeML, histML, hist = D.solver_dotsocp1d(rho0, rho1, 33, 1, dict(tol=1e-4), "inPALM")
    assert hist["iter"][-1] == o_hist["iter"][-1] == 364        # SURVEY.md 8c(iii) known answer
    # The survey's probe also recorded its final KKT vector and sigma (0.592).  Its exact configuration is not
    # recoverable (it is described as "rescale block disabled", but the restated loop stops at 339 that way and at 364
    # only with the block enabled), so the digits can only be held loosely: every KKT entry of the
    # 364-iteration run lies within 10 % of the recorded one, sigma (0.686 here) within 20 %.
    survey_kkt = np.array([6.8e-5, 6.7e-5, 9.0e-5, 1.7e-5, 1.9e-16, 2.6e-5, 8.2e-5])
    np.testing.assert_allclose(hist["kkt"][-1], survey_kkt, rtol=0.10)
    np.testing.assert_allclose(o_hist["kkt"][-1], survey_kkt, rtol=0.10)
    assert abs(o_sigma - 0.592) <= 0.2 * 0.592
    assert D.check_massConservation(out["rho"], 1e-2)
    rho_o, Ex_o = OD.recover_RhoE_1d(ovar, omodel)
    np.testing.assert_allclose(out["rho"], rho_o, atol=1e-7)


def test_run_in_pieces_is_identical():
    """dotsocp_run(n) called repeatedly gives the same trajectory as one call (bench warm-up + timed region)."""
    rho0, rho1 = get_example_2d("example1", 32, 32)
    res = []
    for pieces in ((40,), (7, 13, 20)):
        var, model, o = _gpu_level(rho0, rho1, 16, dict(tol=0.0, maxit=40))
        ctx = D.InPALMContext(var, o, model)
        for k in pieces:
            assert ctx.run(k) == k
        hist, sigma = ctx.finish()
        ctx.close()
        res.append((var.phi.copy(), var.beta.copy(), sigma, hist["kkt"].copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and np.array_equal(res[0][3], res[1][3])


# --------------------------------------------------------------------------------------------------
# time slabs: the multi-GPU algorithm (halo exchange + slab<->pencil transposes of the Poisson solve)
# executed with all slabs on the one GPU of the test box must reproduce the single-slab run
# --------------------------------------------------------------------------------------------------
def _run_slabs(rho0, rho1, nt, opts, nslabs, weight=None):
    var, model, o = _gpu_level(rho0, rho1, nt, opts, "inPALM", weight)
    solve = D.solver_wsocp_inPALM if weight is not None else D.solver_socp_inPALM
    hist, sigma = solve(var, o, model, nslabs=nslabs)
    return var, hist, sigma


@pytest.mark.parametrize("tsolve", ["tridiag", "dct"])
# the x48 / x49 / 1-D cases have >= 12 time layers per slab: the cone pass and the q-step then run in chunks around the
# halo exchanges (Solver::step, split branches); the shorter ones take the unsplit branches
@pytest.mark.parametrize("case", ["dot2d_32x32x16", "dot2d_24x40x12", "dot2d_33x33x17", "dot1d_128x32", "wdot2d_32x32x16",
                                  "dot2d_33x33x49", "wdot2d_32x32x48"])
@pytest.mark.parametrize("nslabs", [2, 3, 4])
def test_time_slabs_match_single_slab(case, nslabs, tsolve, request, monkeypatch):
    """Both ways of solving along t across slabs: partitioned tridiagonal systems (default: 4 numbers per mode over
    the links) and slab <-> pencil transposes around the t-axis DCT (DOTSOCP_TSOLVE=dct)."""
    if "unfused" in request.node.name:
        pytest.skip("time slabs exist on the fused dataflow only")
    monkeypatch.setenv("DOTSOCP_TSOLVE", tsolve)
    weight = None
    if case == "dot1d_128x32":
        rho0, rho1 = get_example_1d("gaussian", 128)
        nt = 32
    else:
        a, b, nt = [int(v) for v in case.split("_")[1].split("x")]
        rho0, rho1 = get_example_2d("example1", a, b)
        if case.startswith("wdot2d"):
            barrier = gene_barrier_of_circle_pillar()
            weight = get_weight_by_barrier(b, a, nt, barrier)
            rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    opts = dict(tol=0.0, maxit=30)
    ref, h1, s1 = _run_slabs(rho0, rho1, nt, opts, 1, weight)
    got, hn, sn = _run_slabs(rho0, rho1, nt, opts, nslabs, weight)
    np.testing.assert_array_equal(hn["iter"], h1["iter"])
    np.testing.assert_allclose(hn["kkt"], h1["kkt"], rtol=1e-7, atol=1e-10)
    assert abs(sn - s1) <= 1e-12 * s1
    # only summation orders differ (tile-boundary partial sums, per-slab norm partials)
    errs = {f: _relerr(getattr(got, f), getattr(ref, f)) for f in FIELDS}
    assert max(errs.values()) <= (1e-8 if weight is not None else 1e-10), errs


@pytest.mark.parametrize("case", ["dot2d_33x33x49", "dot2d_32x32x16", "dot2d_40x24x36"])
def test_messages_on_second_streams_change_nothing(case, request, monkeypatch):
    """Time slabs: by default every message between slabs (halos, tails, the interface exchanges of the partitioned t-solve)
    and the two small kernels between those exchanges travel on the slabs' second streams while the main streams run the
    kernels that do not need them (Solver::step, comm_z); DOTSOCP_OVERLAP=0 issues everything on the main streams, one
    after the other.  Same kernels on the same data in both orders: the iterates must be bit-identical -- in-process slabs
    with their own stream pairs (ngpu, the multi-device placement) and sharing the device's pair (nslabs)."""
    if "unfused" in request.node.name:
        pytest.skip("time slabs exist on the fused dataflow only")
    a, b, nt = [int(v) for v in case.split("_")[1].split("x")]
    rho0, rho1 = get_example_2d("example1", a, b)
    opts = dict(tol=0.0, maxit=25)
    for kw in (dict(nslabs=3), dict(ngpu=3)):
        runs = []
        for ov in ("1", "0"):
            monkeypatch.setenv("DOTSOCP_OVERLAP", ov)
            var, model, o = _gpu_level(rho0, rho1, nt, opts, "inPALM", None)
            ctx = D.InPALMContext(var, o, model, weighted=False, **kw)
            try:
                ctx.run(-1)
                hist, sigma = ctx.finish()
            finally:
                ctx.close()
            runs.append((var, hist, sigma))
        (v1, h1, s1), (v0, h0, s0) = runs
        np.testing.assert_array_equal(h1["kkt"], h0["kkt"])
        assert s1 == s0
        for f in FIELDS:
            np.testing.assert_array_equal(getattr(v1, f), getattr(v0, f))


def test_time_slabs_free_running_against_oracle(request):
    if "unfused" in request.node.name:
        pytest.skip("time slabs exist on the fused dataflow only")
    rho0, rho1 = get_example_2d("example1", 32, 32)
    ovar, omodel, o_hist, o_sigma = OD.solve_single_level(rho0, rho1, 16, dict(tol=1e-4))
    var, hist, sigma = _run_slabs(rho0, rho1, 16, dict(tol=1e-4), 4)
    assert hist["iter"][-1] == o_hist["iter"][-1]
    D.recoverOrgVar(var)
    errs = {f: _relerr(getattr(var, f), getattr(ovar, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-7, errs


def test_rccl_communicator_world1(request):
    """One-process-per-GPU mode with a world of one rank: exercises the RCCL binding (dlopen,
    ncclCommInitRank, the all-reduce of the KKT sums) on the single GPU of the test box; the
    neighbour exchanges of larger worlds are the ones the in-process slab tests validate."""
    if "unfused" in request.node.name:
        pytest.skip("time slabs exist on the fused dataflow only")
    rho0, rho1 = get_example_2d("example1", 32, 32)
    ref, h1, s1 = _run_slabs(rho0, rho1, 16, dict(tol=0.0, maxit=25), 1)
    var, model = D.initialize_slab(rho0, rho1, 16, 0, 16)
    o = OD.default_opts(dict(tol=0.0, maxit=25), "inPALM", False)
    D.InitialScaling(var, model, True, None, dim=2)
    # the slab initialisation (||c|| from its two non-zero layers, h from the global node count) must give
    # exactly the scaling constants of the full-grid one
    full, fmodel = D.initialize(rho0, rho1, 16)
    D.InitialScaling(full, fmodel, True, None, dim=2)
    assert (var.D, var.E, var.cScale, var.dScale) == (full.D, full.E, full.cScale, full.dScale)
    assert var.D == ref.D
    ctx = D.InPALMContext(var, o, model, rccl=(D.capi.rccl_unique_id(), 0, 1))
    ctx.run(-1)
    hist, sigma = ctx.finish(download=False)
    phi = ctx.download(D.capi.F_PHI, var.phi)
    ctx.close()
    np.testing.assert_array_equal(hist["iter"], h1["iter"])
    np.testing.assert_allclose(hist["kkt"], h1["kkt"], rtol=1e-9, atol=1e-12)
    assert _relerr(phi, ref.phi) <= 1e-11


def test_pitched_rows_change_nothing():
    """Single-slab contexts store rows whose length is no multiple of 16 doubles padded to the next 128-byte boundary
    (common.h: Grid::py; the 2^k+1 grids of the multilevel driver).  Only addresses change -- tiles, summation orders and
    the arithmetic of every kernel are the same; the one difference is WHICH two lines share a complex transform in the
    t pass of the Poisson solve (reference layout: the ny * nx columns of a layer are paired across row ends, pitched:
    within rows), which moves results by rounding.  So every iterate, the KKT history and the device-side outputs agree
    with the reference layout (DOTSOCP_PITCH=0; read once per process, hence the subprocesses) to <= 1e-11 of the
    array's largest entry, with identical KKT schedules.  Covers 2-D (ny = 65 -> 80, nx a power of two and not),
    weighted, 1-D (129 -> 144), PALM and acc-ADMM."""
    import os
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, numpy as np, dotsocp_amd as D\n"
        "from oracle import driver as OD\n"
        "from oracle.examples import get_example_2d, get_example_1d, gene_barrier_of_circle_pillar, get_weight_by_barrier, ensure_barrier_validity\n"
        "out = {}\n"
        "def run(tag, rho0, rho1, nt, method='inPALM', weight=None, K=40):\n"
        "    dim = 2 if np.ndim(rho0) == 2 else 1\n"
        "    var, model = D.initialize(rho0, rho1, nt)\n"
        "    if weight is not None: model.weight = weight\n"
        "    D.InitialScaling(var, model, True, None, dim=dim, weighted=weight is not None)\n"
        "    o = OD.default_opts(dict(tol=0.0, maxit=K), method, weight is not None)\n"
        "    ctx = D.InPALMContext(var, o, model, weighted=weight is not None, method=method)\n"
        "    ctx.run(K); hist, sigma = ctx.finish(download=True); ctx.close()\n"
        "    for f in ('phi', 'q', 'z', 'alpha', 'beta'): out[tag + f] = np.asarray(getattr(var, f)).copy()\n"
        "    out[tag + 'kkt'] = np.asarray(hist['kkt']); out[tag + 'sigma'] = np.array([sigma])\n"
        "r0, r1 = get_example_2d('example1', 65, 33)\n"
        "run('a', r0, r1, 17)\n"
        "r0, r1 = get_example_2d('example1', 49, 64)\n"
        "run('b', r0, r1, 16)\n"
        "import os\n"
        "if os.environ.get('DOTSOCP_FUSED') != '0':      # PALM and acc-ADMM exist on the fused dataflow only\n"
        "    run('p', r0, r1, 16, method='PALM', K=15)\n"
        "    run('h', r0, r1, 16, method='acc-ADMM', K=15)\n"
        "r0, r1 = get_example_2d('example1', 33, 33)\n"
        "bar = gene_barrier_of_circle_pillar(); w = get_weight_by_barrier(33, 33, 9, bar); r0, r1, _ = ensure_barrier_validity(r0, r1, bar)\n"
        "run('w', r0, r1, 9, weight=w, K=20)\n"
        "r0, r1 = get_example_1d('gaussian', 129)\n"
        "run('d', r0, r1, 33)\n"
        "o2, tML, hML, h = D.solver_dotsocp2d(*get_example_2d('example1', 33, 33), 17, 2, dict(tol=1e-3), 'inPALM')\n"
        "for f in ('rho', 'Ex', 'Ey', 'q0', 'bx', 'by'): out['m' + f] = np.asarray(o2[f])\n"
        "out['miter'] = np.asarray(h['iter'])\n"
        "np.savez(sys.argv[1], **out)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for flag in ("0", "1"):
            path = os.path.join(tmp, f"pitch{flag}.npz")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, DOTSOCP_PITCH=flag), cwd=root,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
            with np.load(path) as z:
                res[flag] = {k: z[k].copy() for k in z.files}
    assert set(res["0"]) == set(res["1"]) and len(res["0"]) > 30
    for k in res["0"]:
        a, b = res["1"][k], res["0"][k]
        if k.endswith("iter"):
            np.testing.assert_array_equal(a, b, err_msg=k)
        elif k.endswith("kkt") or k.endswith("sigma"):
            np.testing.assert_allclose(a, b, rtol=1e-8, atol=1e-10, err_msg=k)     # entries that are rounding noise themselves (weighted dual feasibility ~1e-11)
        else:
            scale = max(np.max(np.abs(b)), 1e-300)
            assert np.max(np.abs(a - b)) <= 1e-11 * scale, (k, np.max(np.abs(a - b)) / scale)
