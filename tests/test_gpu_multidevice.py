"""Single-process multi-device mode (dotsocp_create_multi / opts.ngpu of the MEX gateway): the one MATLAB process of
solver_dotsocp2d.m:208 driving several GPUs.  Every time slab has its own device, its own pair of streams and
event-ordered peer copies to its neighbours.  The test box has ONE GPU, so slab r's device is (0 + r) mod 1 = 0 for
every slab: what is exercised here is everything but the physical link -- per-slab streams running concurrently,
the event ordering of every cross-slab copy (a missing dependency shows up as a wrong or irreproducible result),
the host-side reduction of the per-slab KKT sums, global upload / download / outputs.  Checked against the
single-slab run of the same problem, for all three loops."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle.examples import (ensure_barrier_validity, gene_barrier_of_circle_pillar, get_example_1d,
                             get_example_2d, get_weight_by_barrier)

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")


def _relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _run(rho0, rho1, nt, opts, method, weight=None, outputs=False, **kw):
    dim = 2 if np.ndim(rho0) == 2 else 1
    var, model = D.initialize(rho0, rho1, nt)
    if weight is not None:
        model.weight = np.asarray(weight, dtype=np.float64)
    o = OD.default_opts(opts, method, weight is not None)
    D.InitialScaling(var, model, o["scaling"], None, dim=dim, weighted=weight is not None)
    ctx = D.InPALMContext(var, o, model, weighted=weight is not None, method=method, **kw)
    ctx.run(-1)
    hist, sigma = ctx.finish(download=True)
    outs = ctx.outputs() if outputs else None
    ctx.close()
    return var, model, hist, sigma, outs


@pytest.mark.parametrize("method,ngpu,ny,nx,nt,K", [
    ("inPALM", 2, 32, 24, 16, 30), ("inPALM", 3, 40, 36, 48, 30), ("inPALM", 4, 64, 64, 64, 25),
    ("inPALM", 8, 32, 32, 128, 20), ("ALG2", 2, 33, 17, 9, 20), ("PALM", 2, 32, 24, 16, 25), ("PALM", 3, 24, 40, 48, 20),
    ("acc-ADMM", 2, 32, 24, 16, 25), ("acc-ADMM", 4, 32, 32, 64, 20),
    # slabs of 33 .. 64 time nodes (register-resident tridiagonal kernels of width 64) and longer ones (memory-resident)
    ("inPALM", 2, 24, 20, 100, 20), ("PALM", 2, 24, 20, 70, 14), ("inPALM", 2, 16, 12, 150, 12)])
def test_multi_device_matches_single_slab(method, ngpu, ny, nx, nt, K):
    rho0, rho1 = get_example_2d("example1", ny, nx)
    opts = dict(tol=0.0, maxit=K)
    ref, _, h1, s1, _ = _run(rho0, rho1, nt, opts, method)
    got, _, hn, sn, _ = _run(rho0, rho1, nt, opts, method, ngpu=ngpu)
    np.testing.assert_array_equal(hn["iter"], h1["iter"])
    np.testing.assert_allclose(hn["kkt"], h1["kkt"], rtol=1e-7, atol=1e-10)
    assert abs(sn - s1) <= 1e-12 * s1
    errs = {f: _relerr(getattr(got, f), getattr(ref, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-10, errs
    # several streams run concurrently: a race would make two runs differ
    again, _, ha, sa, _ = _run(rho0, rho1, nt, opts, method, ngpu=ngpu)
    assert sa == sn and np.array_equal(ha["kkt"], hn["kkt"])
    for f in FIELDS:
        assert np.array_equal(getattr(again, f), getattr(got, f)), f


@pytest.mark.parametrize("method,ngpu,ny,nx,nt,K", [("inPALM", 3, 40, 36, 48, 30), ("inPALM", 8, 32, 32, 128, 20),
                                                     ("acc-ADMM", 4, 32, 32, 64, 20), ("PALM", 3, 24, 40, 48, 20)])
def test_interface_messages_by_gather_launch_or_by_copies(method, ngpu, ny, nx, nt, K, monkeypatch):
    """Between the slabs of one process the interface messages of the partitioned tridiagonal solve and the neighbour
    messages of a group (halos, tails) are pulled by one launch per receiving slab (default) or travel as event-ordered
    copies, one per message (DOTSOCP_TRI_GATHER=0, DOTSOCP_MSG_BATCH=0): the same doubles end up in the same places, so
    the two runs must agree to the last bit."""
    rho0, rho1 = get_example_2d("example1", ny, nx)
    opts = dict(tol=0.0, maxit=K)
    monkeypatch.setenv("DOTSOCP_TRI_GATHER", "0")
    monkeypatch.setenv("DOTSOCP_MSG_BATCH", "0")
    ref, _, h0, s0, _ = _run(rho0, rho1, nt, opts, method, ngpu=ngpu)
    monkeypatch.setenv("DOTSOCP_TRI_GATHER", "1")
    monkeypatch.setenv("DOTSOCP_MSG_BATCH", "1")
    got, _, h1, s1, _ = _run(rho0, rho1, nt, opts, method, ngpu=ngpu)
    assert s1 == s0 and np.array_equal(h1["kkt"], h0["kkt"])
    for f in FIELDS:
        assert np.array_equal(getattr(got, f), getattr(ref, f)), f


@pytest.mark.parametrize("tsolve", ["tridiag", "dct"])
def test_multi_device_both_t_solves_weighted_and_1d(tsolve, monkeypatch):
    monkeypatch.setenv("DOTSOCP_TSOLVE", tsolve)
    rho0, rho1 = get_example_2d("example1", 32, 32)
    barrier = gene_barrier_of_circle_pillar()
    weight = get_weight_by_barrier(32, 32, 16, barrier)
    rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    opts = dict(tol=0.0, maxit=25)
    ref, _, h1, s1, _ = _run(rho0, rho1, 16, opts, "inPALM", weight)
    got, _, hn, sn, _ = _run(rho0, rho1, 16, opts, "inPALM", weight, ngpu=3)
    np.testing.assert_array_equal(hn["iter"], h1["iter"])
    errs = {f: _relerr(getattr(got, f), getattr(ref, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-8, errs
    r0, r1 = get_example_1d("gaussian", 128)
    ref, _, h1, s1, _ = _run(r0, r1, 32, dict(tol=0.0, maxit=40), "inPALM")
    got, _, hn, sn, _ = _run(r0, r1, 32, dict(tol=0.0, maxit=40), "inPALM", ngpu=4)
    np.testing.assert_array_equal(hn["iter"], h1["iter"])
    errs = {f: _relerr(getattr(got, f), getattr(ref, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-10, errs


@pytest.mark.parametrize("case,ngpu", [("dot2d", 3), ("wdot2d", 2), ("dot1d", 4)])
def test_outputs_on_time_slabs_equal_host_recovery(case, ngpu):
    """dotsocp_recover_outputs on a multi-slab context (the density at a slab's first node averages over the left
    neighbour's last cell) against recoverOrgVar + recover_RhoE + recover_q of the iterates downloaded from the same
    context: same arithmetic in the same order -> identical."""
    weight = None
    if case == "dot1d":
        rho0, rho1 = get_example_1d("gaussian", 65)
        nt, dim = 17, 1
    else:
        rho0, rho1 = get_example_2d("example1", 24, 40)
        nt, dim = 12, 2
        if case == "wdot2d":
            barrier = gene_barrier_of_circle_pillar()
            weight = get_weight_by_barrier(40, 24, nt, barrier)
            rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    var, model, hist, sigma, dev = _run(rho0, rho1, nt, dict(tol=0.0, maxit=30), "inPALM", weight, outputs=True, ngpu=ngpu)
    D.recoverOrgVar(var)
    rE = D.recover_RhoE(var, model, weighted=weight is not None)
    rq = D.recover_q(var, model)
    names = ("rho", "Ex", "q0", "bx") if dim == 1 else ("rho", "Ex", "Ey", "q0", "bx", "by")
    host = dict(zip(names, (rE + rq) if dim == 2 else (rE[0], rE[1], rq[0], rq[1])))
    for k in names:
        assert dev[k].shape == host[k].shape, k
        np.testing.assert_array_equal(dev[k], host[k], err_msg=k)


def test_free_running_multi_device_against_oracle():
    rho0, rho1 = get_example_2d("example1", 32, 32)
    ovar, omodel, o_hist, o_sigma = OD.solve_single_level(rho0, rho1, 16, dict(tol=1e-4))
    var, _, hist, sigma, _ = _run(rho0, rho1, 16, dict(tol=1e-4), "inPALM", ngpu=4)
    assert hist["iter"][-1] == o_hist["iter"][-1]
    D.recoverOrgVar(var)
    errs = {f: _relerr(getattr(var, f), getattr(ovar, f)) for f in FIELDS}
    assert max(errs.values()) <= 1e-7, errs


@pytest.mark.parametrize("case,ngpu", [("dot2d", 4), ("wdot2d", 3), ("dot1d", 4)])
def test_multilevel_driver_on_time_slabs(case, ngpu):
    """The whole multilevel driver (solver_dotsocp2d.m:154-250 and twins) with every level cut into time slabs of this one
    process: dotsocp_jump_next_level between multi-slab contexts with DIFFERENT slab counts per level (a coarse level
    with few time nodes gets fewer slabs), outputs recovered on the slabs.  Per-level iteration counts must equal the
    one-slab run's, the outputs agree to the slab tolerance."""
    if case == "dot1d":
        rho0, rho1 = get_example_1d("gaussian", 129)
        run = lambda o: D.solver_dotsocp1d(rho0, rho1, 33, 3, o, "inPALM")            # noqa: E731
        opts = dict(tol=1e-4)
    elif case == "wdot2d":
        rho0, rho1 = get_example_2d("example1", 33, 33)
        barrier = gene_barrier_of_circle_pillar()
        weight = get_weight_by_barrier(33, 33, 17, barrier)
        rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
        run = lambda o: D.solver_wdotsocp2d(rho0, rho1, 17, 2, dict(o, weight=weight), "inPALM", barrier=barrier)   # noqa: E731
        opts = dict(tol=1e-3, maxit=400)
    else:
        rho0, rho1 = get_example_2d("example1", 33, 33)
        run = lambda o: D.solver_dotsocp2d(rho0, rho1, 17, 3, o, "inPALM")            # noqa: E731
        opts = dict(tol=1e-4)
    ref_out, ref_time, ref_ml, _ = run(dict(opts))
    got_out, got_time, got_ml, _ = run(dict(opts, ngpu=ngpu))
    assert [int(t["Iters"]) for t in got_time[:-1]] == [int(t["Iters"]) for t in ref_time[:-1]]
    np.testing.assert_array_equal(got_ml["iter"], ref_ml["iter"])
    np.testing.assert_allclose(got_ml["kkt"], ref_ml["kkt"], rtol=1e-5, atol=1e-9)
    for k in ref_out:
        err = np.max(np.abs(got_out[k] - ref_out[k])) / max(np.max(np.abs(ref_out[k])), 1e-300)
        assert err <= 1e-7, (k, err)


_STRESS_SCRIPT = """
import hashlib, sys
import numpy as np
import dotsocp_amd as D
from oracle import driver as OD
from oracle.examples import get_example_2d
h = hashlib.sha256()
for method, ngpu, ny, nx, nt, K in (("inPALM", 4, 40, 36, 64, 24), ("inPALM", 3, 33, 17, 48, 20), ("PALM", 3, 24, 40, 48, 14),
                                    ("acc-ADMM", 2, 32, 24, 16, 14)):
    rho0, rho1 = get_example_2d("example1", ny, nx)
    var, model = D.initialize(rho0, rho1, nt)
    o = OD.default_opts(dict(tol=0.0, maxit=K), method, False)
    D.InitialScaling(var, model, True, None, dim=2)
    ctx = D.InPALMContext(var, o, model, method=method, ngpu=ngpu)
    ctx.run(-1)
    hist, sigma = ctx.finish(download=True)
    out = ctx.outputs()
    ctx.close()
    for a in (var.phi, var.q, var.z, var.alpha, var.beta, hist["kkt"], out["rho"]):
        assert np.all(np.isfinite(a))
        h.update(np.ascontiguousarray(a).tobytes())
print("HASH", h.hexdigest())
"""


def test_random_stream_stalls_change_nothing():
    """Race detector (DOTSOCP_STRESS_STREAMS=1, csrc/guard.hip): stalls of random length in front of the work of every
    slab stream -- main and second stream of every slab -- so that no ordering between streams can come from kernels
    happening to take their usual time.  All three loops on 2 - 4 concurrent slabs must produce bit-identical iterates,
    KKT histories and outputs with and without the stalls."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for stress in ("0", "1", "1"):
        env = dict(os.environ, DOTSOCP_STRESS_STREAMS=stress, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        out = subprocess.run([sys.executable, "-c", _STRESS_SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
        hashes.append([ln for ln in out.stdout.splitlines() if ln.startswith("HASH")][-1])
    assert hashes[0] == hashes[1] == hashes[2], hashes


def test_slab_threads_change_nothing():
    """One host thread per slab issues that slab's launches, copies, event records and stream waits while run() is active
    (csrc/defer.h; opt-in: DOTSOCP_HOST_THREADS=1) -- against the same runs with everything issued by the caller's thread
    (DOTSOCP_HOST_THREADS=0, the default).  The device sees the same operations in the same per-stream order with the same event
    dependencies, so iterates, KKT histories and outputs of all three loops are bit-identical -- also under random
    stream stalls, which move the relative timing of the slab threads' work around, and with the pull launches
    replaced by event-ordered copies (many more cross-slab events per iteration)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for threads, stress, batch in (("0", "0", "1"), ("1", "0", "1"), ("1", "1", "1"), ("1", "1", "0")):
        env = dict(os.environ, DOTSOCP_HOST_THREADS=threads, DOTSOCP_STRESS_STREAMS=stress, DOTSOCP_MSG_BATCH=batch,
                   DOTSOCP_TRI_GATHER=batch, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        out = subprocess.run([sys.executable, "-c", _STRESS_SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
        hashes.append([ln for ln in out.stdout.splitlines() if ln.startswith("HASH")][-1])
    assert len(set(hashes)) == 1, hashes
