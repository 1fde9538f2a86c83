"""One process per slab -- the production multi-GPU structure -- on the single GPU of the test box.
RCCL refuses several ranks on one device, so the ranks talk through tests/fake_rccl (a shared-memory
stand-in loaded via DOTSOCP_RCCL_LIB, test infrastructure only).  What this exercises is libdotsocp's
own rank logic: dotsocp_attach_rccl, local uploads / downloads, shift() (who sends which layer to
whom), the slab<->pencil transposes with one send/recv per peer, and the all-reduced KKT sums."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE_SO = os.path.join(FAKE_DIR, "libfake_rccl.so")
pytestmark = pytest.mark.gpu

NY, NX, K = 32, 24, 25


def _build_fake():
    src = os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(FAKE_SO) or os.path.getmtime(FAKE_SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-Wno-unused-result", "-o", FAKE_SO, src])


def _random_state(NT):
    """a generic global state in the reference layout (seeded: every process builds the same one); z and beta are zero in
    the slots of cone rows without an edge (tests/test_gpu_random_state.py says why)"""
    from oracle import mexops
    rng = np.random.default_rng(31)
    nz = NY * NX * (NT - 1)
    nq = nz + NY * (NX - 1) * NT + (NY - 1) * NX * NT
    st = {"phi": rng.standard_normal(NY * NX * NT), "q": 0.3 * rng.standard_normal(nq), "alpha": 0.5 * rng.standard_normal(nq),
          "z": np.asfortranarray(0.6 * rng.standard_normal((nz, 10))), "beta": np.asfortranarray(0.4 * rng.standard_normal((nz, 10)))}
    probe = np.full((nz, 10), np.nan, order="F")
    mexops.mexBFd(probe, st["q"], NT, NX, NY)
    st["z"][np.isnan(probe)] = 0.0
    st["beta"][np.isnan(probe)] = 0.0
    return st


def _slab_of(st, NT, t0, t1):
    """the layers [t0, t1) of a global state: what a rank of the one-process-per-GPU mode holds and uploads"""
    ntl = t1 - t0
    ncl = ntl if t1 < NT else ntl - 1
    nz = NY * NX * (NT - 1)
    nbx = NY * (NX - 1) * NT
    out = {"phi": st["phi"].reshape((NY, NX, NT), order="F")[:, :, t0:t1].ravel(order="F")}
    for f in ("q", "alpha"):
        v = st[f]
        c0 = v[:nz].reshape((NY, NX, NT - 1), order="F")[:, :, t0:t0 + ncl]
        bx = v[nz:nz + nbx].reshape((NY, NX - 1, NT), order="F")[:, :, t0:t1]
        by = v[nz + nbx:].reshape((NY - 1, NX, NT), order="F")[:, :, t0:t1]
        out[f] = np.concatenate([c0.ravel(order="F"), bx.ravel(order="F"), by.ravel(order="F")])
    for f in ("z", "beta"):
        out[f] = np.asfortranarray(st[f].reshape((NY, NX, NT - 1, 10), order="F")[:, :, t0:t0 + ncl].reshape((NY * NX * ncl, 10), order="F"))
    return out


def _worker(rank, world, uid_hex, q, NT, method="inPALM", start="zeros"):
    os.environ["DOTSOCP_RCCL_LIB"] = FAKE_SO
    sys.path.insert(0, ROOT)
    try:
        import dotsocp_amd as D
        from oracle import driver as OD
        from oracle.examples import get_example_2d
        rho0, rho1 = get_example_2d("example1", NY, NX)
        t0, t1 = D.capi.slab_range(NT, world, rank)
        var, model = D.initialize_slab(rho0, rho1, NT, t0, t1)
        o = OD.default_opts(dict(tol=0.0, maxit=K), method, False)
        D.InitialScaling(var, model, True, None, dim=2)
        if start == "random":
            for f, a in _slab_of(_random_state(NT), NT, t0, t1).items():
                setattr(var, f, a)
        ctx = D.InPALMContext(var, o, model, rccl=(bytes.fromhex(uid_hex), rank, world), method=method)
        ctx.run(-1)
        hist, sigma = ctx.finish(download=False)
        ntl = t1 - t0
        ncl = ntl if t1 < NT else ntl - 1
        phi = ctx.download(D.capi.F_PHI, np.empty(NY * NX * ntl))
        nq = NY * NX * ncl + (NY * (NX - 1) + (NY - 1) * NX) * ntl
        qv = ctx.download(D.capi.F_Q, np.empty(nq))
        beta = ctx.download(D.capi.F_BETA, np.empty((NY * NX * ncl, 10)))
        ctx.close()
        q.put((rank, "ok", t0, t1, phi, qv, beta, hist["kkt"], hist["iter"], sigma))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))


# nt = 32 on two ranks / 64 on four: 15-16 cell layers per slab = two cone chunks, i.e. the path where the q-halo exchange runs beside
# the first chunk (Solver::step, split branch); nt = 16: one chunk per slab, exchange in front of the cone pass
@pytest.mark.parametrize("world,tsolve,NT", [(2, "tridiag", 16), (2, "dct", 16), (4, "tridiag", 16), (4, "dct", 16),
                                             (2, "tridiag", 32), (2, "dct", 32), (4, "tridiag", 64)])
def test_one_process_per_slab_matches_single_process(world, tsolve, NT, monkeypatch):
    _one_process_per_slab(world, tsolve, NT, "inPALM", monkeypatch)


@pytest.mark.parametrize("world,tsolve,NT", [(4, "tridiag", 64), (2, "dct", 32)])
def test_one_process_per_slab_under_random_stream_stalls(world, tsolve, NT, monkeypatch):
    """The rank processes run with DOTSOCP_STRESS_STREAMS=1 (random stalls in front of the work of both streams of every
    rank, csrc/guard.hip): the overlap of the cone chunks / middle q-step chunks with the neighbour exchanges must rest
    on events alone."""
    monkeypatch.setenv("DOTSOCP_STRESS_STREAMS", "1")
    _one_process_per_slab(world, tsolve, NT, "inPALM", monkeypatch)


@pytest.mark.parametrize("world,NT", [(2, 16), (3, 48)])
def test_palm_one_process_per_slab(world, NT, monkeypatch):
    """solver_socp_PALM.m's loop in time-slab mode, one process per slab"""
    _one_process_per_slab(world, "tridiag", NT, "PALM", monkeypatch)


@pytest.mark.parametrize("world,NT", [(2, 16), (3, 48)])
def test_accadmm_one_process_per_slab(world, NT, monkeypatch):
    """solver_socp_accADMM.m's loop in time-slab mode, one process per slab"""
    _one_process_per_slab(world, "tridiag", NT, "acc-ADMM", monkeypatch)


@pytest.mark.parametrize("world,NT,method", [(2, 16, "inPALM"), (3, 48, "inPALM"), (4, 64, "inPALM"), (3, 48, "PALM"), (2, 32, "acc-ADMM")])
def test_one_process_per_slab_from_a_random_state(world, NT, method, monkeypatch):
    """Every rank uploads ITS layers of one generic global state (phi, q, z, alpha, beta all non-zero): the slab-local
    field layout of upload(), the first halo / u0 / tail exchanges with generic layers, and the loop against the
    single-process run from the same state."""
    _one_process_per_slab(world, "tridiag", NT, method, monkeypatch, start="random")


def _one_process_per_slab(world, tsolve, NT, method, monkeypatch, start="zeros"):
    monkeypatch.setenv("DOTSOCP_TSOLVE", tsolve)        # inherited by the rank processes
    import multiprocessing as mp
    _build_fake()
    sys.path.insert(0, ROOT)
    import dotsocp_amd as D
    from oracle import driver as OD
    from oracle.examples import get_example_2d
    # single-process reference
    rho0, rho1 = get_example_2d("example1", NY, NX)
    var, model = D.initialize(rho0, rho1, NT)
    o = OD.default_opts(dict(tol=0.0, maxit=K), method, False)
    D.InitialScaling(var, model, True, None, dim=2)
    if start == "random":
        for f, a in _random_state(NT).items():
            setattr(var, f, a)
    solve1 = {"PALM": D.solver_socp_PALM, "acc-ADMM": D.solver_socp_accADMM}.get(method, D.solver_socp_inPALM)
    hist1, sigma1 = solve1(var, o, model)
    phi1 = var.phi.reshape((NY, NX, NT), order="F")
    beta1 = var.beta.reshape((NY, NX, NT - 1, 10), order="F")
    qi = var.qInd
    q0_1 = var.q[:qi.bx].reshape((NY, NX, NT - 1), order="F")
    bx_1 = var.q[qi.bx:qi.by].reshape((NY, NX - 1, NT), order="F")
    by_1 = var.q[qi.by:].reshape((NY - 1, NX, NT), order="F")

    os.environ["DOTSOCP_RCCL_LIB"] = FAKE_SO
    uid = D.capi.rccl_unique_id()
    ctx = mp.get_context("spawn")
    qu = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, uid.hex(), qu, NT, method, start)) for r in range(world)]
    for p in procs:
        p.start()
    results = [qu.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for res in results:
        assert res[1] == "ok", f"rank {res[0]}: {res[1]}"
    tol = 1e-10
    for rank, _, t0, t1, phi, qv, beta, kkt, iters, sigma in results:
        ntl = t1 - t0
        ncl = ntl if t1 < NT else ntl - 1
        np.testing.assert_array_equal(iters, hist1["iter"])
        np.testing.assert_allclose(kkt, hist1["kkt"], rtol=1e-7, atol=1e-10)
        assert abs(sigma - sigma1) <= 1e-12 * sigma1
        assert np.max(np.abs(phi.reshape((NY, NX, ntl), order="F") - phi1[:, :, t0:t1])) <= tol * np.abs(phi1).max()
        n0 = NY * NX * ncl
        nb = NY * (NX - 1) * ntl
        assert np.max(np.abs(qv[:n0].reshape((NY, NX, ncl), order="F") - q0_1[:, :, t0:t0 + ncl])) <= tol * np.abs(q0_1).max()
        assert np.max(np.abs(qv[n0:n0 + nb].reshape((NY, NX - 1, ntl), order="F") - bx_1[:, :, t0:t1])) <= tol * np.abs(bx_1).max()
        assert np.max(np.abs(qv[n0 + nb:].reshape((NY - 1, NX, ntl), order="F") - by_1[:, :, t0:t1])) <= tol * np.abs(by_1).max()
        # beta comes back multiplied by sigma (finish()), like the single-process run
        b = beta.reshape((NY, NX, ncl, 10), order="F")
        assert np.max(np.abs(b - beta1[:, :, t0:t0 + ncl])) <= 1e-9 * np.abs(beta1).max()


@pytest.mark.parametrize("world", [2, 4])
def test_bench_under_torchrun_rehearsal(world):
    """bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run ...), with the
    ranks sharing the one GPU of the test box: torch.distributed over gloo and the solver's communicator over
    the shared-memory stand-in.  Checks the N > 1 code path of bench.py end to end: rendezvous, unique-id
    broadcast, slab initialisation, timed region, max-over-ranks, ONE JSON line from rank 0 whose iterates agree
    with the N = 1 run of the same grid (same KKT-check count, finite positive throughput)."""
    import json
    import socket
    _build_fake()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DOTSOCP_RCCL_LIB=FAKE_SO, DOTSOCP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    grid = ["--grid", "64", "48", "16"]
    common = ["--steps", "12", "--warmup", "3", "--no-cpu-baseline"] + grid
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads(one.stdout.strip().splitlines()[-1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world)] + common
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["steps"] == 12 and rec["scaling"] == "strong"
    assert rec["metric"] == ref["metric"] and rec["config"]["grid"] == [64, 48, 16]
    assert rec["config"]["kkt_checks_in_timed_region"] == ref["config"]["kkt_checks_in_timed_region"]
    assert np.isfinite(rec["value"]) and rec["value"] > 0
    assert rec["roofline"]["launches"] > 0 and rec["kernel_ms"]["comm"] > 0


def test_bench_line_survives_a_second_pass_that_does_not_finish():
    """bench.py with several ranks times a pass without per-phase events and fills kernel_ms / roofline from a second,
    untimed pass on a communicator of its own.  Should that pass ever hang on real hardware, a watchdog prints the line with
    the timed pass's value and empty phase timers and every rank leaves: here the limit is zero, so it always fires."""
    import json
    import socket
    _build_fake()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DOTSOCP_RCCL_LIB=FAKE_SO, DOTSOCP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               DOTSOCP_BENCH_PASS2_LIMIT="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--no-cpu-baseline", "--grid", "64", "48", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 12 and np.isfinite(rec["value"]) and rec["value"] > 0
    assert "did not finish" in rec["config"]["timed_pass"] and rec["kernel_ms"]["comm"] == 0


@pytest.mark.parametrize("world", [2])
def test_bench_self_launch_rehearsal(world):
    """`python bench.py --gpus N` with NO outer launcher: bench.py starts its own rank processes (self_launch),
    relays rank 0's single JSON line and returns 0.  Ranks share the test box's GPU (gloo + the stand-in)."""
    import json
    _build_fake()
    env = dict(os.environ, DOTSOCP_RCCL_LIB=FAKE_SO, DOTSOCP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "12", "--warmup", "3",
           "--no-cpu-baseline", "--grid", "64", "48", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["steps"] == 12 and rec["config"]["grid"] == [64, 48, 16]
    assert np.isfinite(rec["value"]) and rec["value"] > 0


def test_bench_rank_share_line():
    """`bench.py --rank-share N` (one rank's slab of the N-way split through the code path of a real rank, its messages as
    local copies in tools/libloopback_rccl.so) on a small grid: the line carries the full grid's time of the SAME process and
    the ratio of the two (round-3 verdict: same-run pairs only), and says that its timed pass ran without per-phase events."""
    import json
    lb = os.path.join(ROOT, "tools", "libloopback_rccl.so")
    if not os.path.exists(lb):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "--offload-arch=gfx950", "-o", lb,
                               os.path.join(ROOT, "tools", "loopback_rccl.cpp")])
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DOTSOCP_RCCL_LIB"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rank-share", "4", "--steps", "12", "--warmup", "3",
           "--no-cpu-baseline", "--grid", "64", "48", "64"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    rs = rec["rank_share"]
    assert rs["n"] == 4 and rs["time_nodes"] == 16
    assert rs["full_grid_ms_per_step"] > 0 and abs(rs["ceiling"] - rs["full_grid_ms_per_step"] / rec["ms_per_step"]) < 1e-9
    assert rec["config"]["per_phase_hip_events_in_timed_region"] is False and "without per-phase" in rs["timed_pass"]
    assert rec["kernel_ms"]["poisson"] > 0 and rec["kernel_ms"]["comm"] > 0          # the second, instrumented pass
    assert rec["config"]["box_copy_gbs"] is None or rec["config"]["box_copy_gbs"] > 100
