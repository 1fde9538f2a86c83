#!/usr/bin/env python3
"""bench.py -- ADMM (inPALM) iterations per second of the device-resident loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one inPALM iteration (phi-step, cone projection, q-step, multipliers, plus the
KKT block at the reference's own cadence, solver_socp_inPALM.m:361-379) on the synthetic
Gaussian-to-Gaussian problem of BASELINE.json (Example 5.1, examples/dot2d/gene_example1.m),
levelN = 1, method "inPALM" (tau = 1.9, sigma0 = 1, scaling on), tol = 0 so that every run executes
exactly W + K iterations.  Inputs are resident in HBM before the timed region starts.

N = 1: dot2d 1024 x 1024 x 128 (BASELINE.json configs[2]).  N > 1: the same grid split into N time
slabs, one process per GPU (strong scaling); `--grid 2048 2048 256 --gpus 8` is BASELINE.json configs[3].
N > 1 runs either under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` or plainly
as `python bench.py --gpus N ...`, which starts the N ranks itself (self_launch).  Rank 0 prints ONE JSON line; besides the contract
fields it carries `roofline` (cone-projection kernel, HIP-event timing on the launch stream) and
`cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
BUILD_ROUND = 4         # profiles/cone_proj_traffic.json is only quoted when it was measured on this round's build


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, nargs=3, default=None, metavar=("NY", "NX", "NT"),
                    help="override the grid (default 1024 1024 128)")
    ap.add_argument("--workload", choices=["dot2d", "wdot2d", "dot1d"], default="dot2d")
    ap.add_argument("--method", choices=["inPALM", "ALG2", "PALM", "acc-ADMM"], default="inPALM",
                    help="loop variant (diagnostic; BASELINE.json's metric is quoted on inPALM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nslabs", type=int, default=1,
                    help="diagnostics: run the time-slab algorithm with this many slabs inside ONE process / GPU")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="budget of the CPU baseline sample")
    ap.add_argument("--rank-share", type=int, default=0, metavar="N",
                    help="diagnostics: run ONE rank's slab (the middle one) of the N-way time-slab split on this GPU, with the "
                         "code path of a real rank and its neighbour messages as local copies (tools/loopback_rccl.cpp); "
                         "T(full grid) / T(this) is the compute-only ceiling of the N-GPU strong-scaling curve")
    return ap.parse_args()


def build_problem(D, workload, ny, nx, nt):
    """Synthetic inputs + one level of the driver (initialize + InitialScaling), cold start."""
    weight = None
    if workload == "dot1d":
        rho0, rho1 = D.get_example_1d("gaussian", ny)
        dim = 1
    else:
        # like demo_dot2d.m:40,63 the generator's output goes to the solver as is; its first
        # dimension becomes the solver's ny (socp/dot2d/utils/initialize.m:8-9)
        rho0, rho1 = D.get_example_2d("example1", ny, nx)
        dim = 2
        if workload == "wdot2d":
            barrier = D.gene_barrier_of_circle_pillar()
            weight = D.get_weight_by_barrier(nx, ny, nt, barrier)
            rho0, rho1, _ = D.ensure_barrier_validity(rho0, rho1, barrier)
    var, model = D.initialize(rho0, rho1, nt, lazy_zeros=True)
    if weight is not None:
        model.weight = weight
    D.InitialScaling(var, model, True, None, dim=dim, weighted=weight is not None)
    return var, model, rho0, rho1, weight


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, ny, nx, nt, budget_s):
    """the CPU oracle (numpy + C restatement of the reference dataflow: sparse A / A', FFT-based DCT,
    single-threaded MEX-equivalent loops, the reference's temporaries) timed on this box's host cores, with all cores
    (scipy's DCT threaded like MATLAB's fft; the rest is single-threaded in the reference too) and with ONE thread,
    median of >= 3 single iterations each.  Bounded sample: the same spatial grid with nt = 32 time nodes (a 32-point
    t-axis DCT, all stencils at full spatial size); throughput is scaled to the full grid by the node-count ratio --
    every step of the loop is O(N) up to the log factor of the t-axis transform.  A reported baseline, not a target."""
    from oracle import driver as OD
    from oracle import examples as OE
    from oracle import model as OM
    from oracle.inpalm import InPALMState
    nts = min(nt, 32)
    while ny * nx * nts > 34_000_000 and nts > 4:       # set-up (sparse kron) and ~14 GB of temporaries at 33.6 M nodes
        nts -= 1
    weight = None
    if workload == "dot1d":
        rho0, rho1 = OE.get_example_1d("gaussian", ny)
        nts = nt
    else:
        rho0, rho1 = OE.get_example_2d("example1", ny, nx)
        if workload == "wdot2d":
            barrier = OE.gene_barrier_of_circle_pillar()
            weight = OE.get_weight_by_barrier(nx, ny, nts, barrier)
            rho0, rho1, _ = OE.ensure_barrier_validity(rho0, rho1, barrier)
    var, model, o = OD.make_level(rho0, rho1, nts, dict(tol=0.0, maxit=10 ** 6), "inPALM", weight)
    st = InPALMState(var, o, model, weighted=weight is not None)
    st.run(1)                                   # untimed first iteration (page faults, FFT plans)
    t_start = time.perf_counter()

    def sample(workers, share):
        OM.FFT_WORKERS = workers
        ts = []
        while len(ts) < 3 or (len(ts) < 15 and time.perf_counter() - t_start < budget_s * share):
            t0 = time.perf_counter()
            st.run(1)
            ts.append(time.perf_counter() - t0)
        return ts

    t_all = sample(-1, 0.5)
    t_one = sample(1, 1.0)
    OM.FFT_WORKERS = -1
    nodes_s = ny * (nx if workload != "dot1d" else 1) * nts
    nodes_f = ny * (nx if workload != "dot1d" else 1) * nt
    scale = nodes_s / nodes_f
    med_all, med_one = float(np.median(t_all)), float(np.median(t_one))
    return {
        "value": scale / med_all,
        "unit": "iterations/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "extrapolated": nts != nt,
        "value_1thread": scale / med_one,
        "cpu_model": _cpu_model(),
        "sample_grid": [ny, nx, nts],
        "sample_its_all_cores": 1.0 / med_all,
        "sample_its_1thread": 1.0 / med_one,
        "repeats": [len(t_all), len(t_one)],
        "sample": (f"oracle (restated reference, no MATLAB available) on {ny}x{nx}x{nts}: median of {len(t_all)} single "
                   f"iterations with all {os.cpu_count()} host threads in the DCTs = {1.0 / med_all:.4g} it/s, median of "
                   f"{len(t_one)} with 1 thread = {1.0 / med_one:.4g} it/s (the MEX-equivalent loops and numpy passes are "
                   f"single-threaded in both, like the reference's); scaled by the node ratio {nodes_s}/{nodes_f} to "
                   f"{ny}x{nx}x{nt}; CPU: {_cpu_model()}"),
    }


def copy_rate_gbs(torch, device):
    """Device-to-device copy rate of a 1 GiB buffer in GB/s (bytes read + written over HIP-event time, best of 5): tells
    the pool's two kinds of boxes apart from the bench line itself."""
    try:
        a = torch.empty(1 << 27, dtype=torch.float64, device=f"cuda:{device}")
        b = torch.empty_like(a)
        a.zero_()
        b.copy_(a)
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            e1.synchronize()
            best = max(best, 2.0 * a.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        torch.cuda.empty_cache()
        return round(best, 1)
    except Exception:
        return None


def self_launch(n):
    """`python bench.py --gpus N` without an outer launcher: start the N ranks as fresh child processes (this
    process has neither imported torch nor touched HIP, and it never execs), relay rank 0's JSON line and exit
    with the worst child's return code.  Under `python -m torch.distributed.run` WORLD_SIZE is already set and
    this function is not reached."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    limit = float(os.environ.get("DOTSOCP_BENCH_TIMEOUT", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + limit
    out0, worst, got0 = "", 0, False
    while True:
        if not got0:
            try:
                out0, _ = procs[0].communicate(timeout=2.0)
                got0 = True
            except subprocess.TimeoutExpired:
                pass
        else:
            time.sleep(0.2)
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):      # a rank died: give the others a moment, then stop them
            time.sleep(5.0)
            worst = 1
            break
        if time.time() > deadline:
            worst = 124
            break
    for p in procs:                       # exact PIDs of the children started above, nothing else
        if p.poll() is None:
            p.send_signal(signal.SIGKILL)
            p.wait()
            worst = worst or 124
        elif p.returncode != 0:
            worst = worst or (p.returncode if p.returncode > 0 else 1)
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if worst == 0 and len(lines) != 1:
        worst = 1
    if worst == 0:
        print(lines[0], flush=True)
    else:
        sys.stderr.write(f"bench.py: {n}-rank run failed (rc {worst}); rank 0 stdout tail:\n{out0[-2000:]}\n")
    raise SystemExit(worst)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    share = args.rank_share if args.rank_share > 1 else 0
    if share:
        if world != 1 or args.nslabs != 1 or args.workload != "dot2d":
            raise SystemExit("--rank-share runs one process on one GPU on the dot2d workload")
        lb = os.path.join(ROOT, "tools", "libloopback_rccl.so")
        if not os.path.exists(lb):
            raise SystemExit("tools/libloopback_rccl.so is missing: run __graft_entry__.build()")
        os.environ["DOTSOCP_RCCL_LIB"] = lb           # read when the communicator is attached
    import torch
    import dotsocp_amd as D
    if args.grid:
        ny, nx, nt = args.grid
    elif args.workload == "dot1d":
        ny, nx, nt = 128, 1, 32
    elif args.workload == "wdot2d":
        ny, nx, nt = 512, 512, 128
    else:
        ny, nx, nt = 1024, 1024, 128
    if D.capi.lib().dotsocp_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libdotsocp has no CPU fallback")
    # rehearsal on a box with fewer GPUs than ranks (tests/test_gpu_multiprocess.py): ranks share devices
    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    dist, rccl = None, None
    if world > 1:
        # one process per GPU: torch.distributed (RCCL) carries the rendezvous, the barriers and the
        # max-over-ranks of the timing; the solver's own halo / transpose traffic runs on a second RCCL
        # communicator inside libdotsocp, created from a unique id broadcast here
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # DOTSOCP_BENCH_BACKEND=gloo: rehearsal only (several ranks on one GPU, with DOTSOCP_RCCL_LIB pointing at
        # the test stand-in); the driver's runs use RCCL for both communicators
        backend = os.environ.get("DOTSOCP_BENCH_BACKEND", "nccl")
        tdev = "cuda" if backend == "nccl" else "cpu"
        dist.init_process_group(backend, rank=rank, world_size=world)

        def fresh_rccl():
            """a communicator of the solver's own: unique id from rank 0, broadcast over torch.distributed"""
            uid = torch.zeros(128, dtype=torch.uint8, device=tdev)
            if rank == 0:
                uid = torch.tensor(list(D.capi.rccl_unique_id()), dtype=torch.uint8, device=tdev)
            dist.broadcast(uid, 0)
            return (bytes(uid.cpu().tolist()), rank, world)

        rccl = fresh_rccl()
        if args.workload != "dot2d":
            raise SystemExit("multi-GPU bench runs the dot2d workload")

    # --rank-share and N > 1: a rank's iteration is ~1.7 ms and issues ~40 launches; the ~20 per-phase HIP events per iteration that feed
    # `kernel_ms` cost it 4-5 % (they cost the 11 ms full-grid iteration nothing measurable).  Its timed pass therefore runs
    # WITHOUT them and a second, untimed pass of the same length WITH them fills `kernel_ms` / `roofline`
    two_pass = (bool(share) or world > 1) and not os.environ.get("DOTSOCP_BENCH_NOPROF")
    opts = dict(tau=1.0 if args.method == "ALG2" else 1.9, sigma=1.0, tol=0.0, maxit=args.warmup + args.steps, scaling=True,
                ifCheckStepByStep=False, time_limit=1e9)
    full_ms = None
    # the device arrays of a context are about 46 doubles per node (49 GB at 1024 x 1024 x 128)
    full_fits = share and 46 * 8 * ny * nx * nt < 0.92 * torch.cuda.mem_get_info(device)[0]
    if share and not full_fits:
        # configs[3] (2048 x 2048 x 256 = 392 GB): the grid exists only as its slabs; the share is timed alone
        weight = None
        share_rank = share // 2
        rccl = (bytes(128), share_rank, share)
        rho0, rho1 = D.get_example_2d("example1", ny, nx)
        t0s, t1s = D.capi.slab_range(nt, share, share_rank)
        var, model = D.initialize_slab(rho0, rho1, nt, t0s, t1s)
        D.InitialScaling(var, model, True, None, dim=2)
    elif share:
        # the full grid first, in this same process on this same box (W warm-up, K timed iterations, no per-phase events): the
        # ceiling T(full grid) / T(rank's share) is only meaningful as a same-run pair
        fvar, fmodel, _, _, _ = build_problem(D, "dot2d", ny, nx, nt)
        fopts = dict(opts, maxit=args.warmup + args.steps)
        fctx = D.InPALMContext(fvar, fopts, fmodel, weighted=False, device=device, profiling=False, method=args.method)
        assert fctx.run(args.warmup) == args.warmup
        fctx.synchronize()
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        assert fctx.run(args.steps) == args.steps
        fctx.synchronize()
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - tf0) / args.steps * 1e3
        fctx.finish(download=False)
        fctx.close()
        del fctx, fvar, fmodel
        # the middle slab of the N-way split: both neighbours exist, every exchange of a real rank takes place
        weight = None
        share_rank = share // 2
        rccl = (bytes(128), share_rank, share)
        rho0, rho1 = D.get_example_2d("example1", ny, nx)
        t0s, t1s = D.capi.slab_range(nt, share, share_rank)
        var, model = D.initialize_slab(rho0, rho1, nt, t0s, t1s)
        D.InitialScaling(var, model, True, None, dim=2)
    elif world == 1:
        var, model, rho0, rho1, weight = build_problem(D, args.workload, ny, nx, nt)
    else:
        weight = None
        rho0, rho1 = D.get_example_2d("example1", ny, nx)
        t0s, t1s = D.capi.slab_range(nt, world, rank)
        var, model = D.initialize_slab(rho0, rho1, nt, t0s, t1s)
        D.InitialScaling(var, model, True, None, dim=2)
    ctx = D.InPALMContext(var, opts, model, weighted=weight is not None, device=device, profiling=False, rccl=rccl,
                          nslabs=args.nslabs, method=args.method)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    box_copy_gbs = copy_rate_gbs(torch, device)
    done = ctx.run(args.warmup)
    assert done == args.warmup
    if not os.environ.get("DOTSOCP_BENCH_NOPROF") and not two_pass:
        D.capi.check(D.capi.lib().dotsocp_set_profiling(ctx._ctx, 1))
    fence()
    t0 = time.perf_counter()
    done = ctx.run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    assert done == args.steps
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    hist, sigma = ctx.finish(download=False)

    def emit(times, fallback_note=None):
        """build and print the JSON line (rank 0) from the timed pass's figures and the phase timers `times`"""
        # per-launch sizes of THIS rank's slab (the whole grid at N = 1)
        t0s, t1s = D.capi.slab_range(nt, share, share // 2) if share else D.capi.slab_range(nt, world, rank)
        ntl = t1s - t0s
        ncl = ntl if t1s < nt else ntl - 1
        Nz = ny * nx * ncl
        Nq = Nz + (ny * (nx - 1) + (ny - 1) * nx) * (ncl + 1)
        # Dominant kernel: the fused cone kernel in its steady-state mode (deferred multiplier update +
        # cone projection + adjoint gather).  Algorithmic bytes per launch:
        # beta in + beta out (20 Nz) + q^{k-1}, q^k in (2 Nq) + adjoint sums out (Nq), fp64.
        if args.method == "acc-ADMM":
            # multiplier + z-step + Halpern step of z, beta + next adjoint sums: z, beta, anchors in, z, beta out
            kname, (proj_ms, proj_n) = "k_acc_cone<1,4> (multiplier + projection + Halpern + gather)", times["acc_cone"]
            alg_bytes = 8.0 * (60 * Nz + 2 * Nq)
        elif args.method == "PALM" and times["cone_fused_b"][1] > 0:
            # one pass over beta per iteration: beta in + out, q~^{k-1}, q^k, q~^k in, two adjoint gathers out
            kname, (proj_ms, proj_n) = "k_cone_fused<5,4> (beta update + cone projection + two adjoint gathers)", times["cone_fused_b"]
            alg_bytes = 8.0 * (20 * Nz + 5 * Nq)
        elif args.method == "PALM":
            kname, (proj_ms, proj_n) = "k_cone_fused<0,4> (cone projection + adjoint gather)", times["cone_fused_a"]
            alg_bytes = 8.0 * (10 * Nz + 2 * Nq)
        elif times["cone_fused_b"][1] > 0:
            kname, (proj_ms, proj_n) = "k_cone_fused<1,4> (beta update + cone projection + adjoint gather)", times["cone_fused_b"]
            alg_bytes = 8.0 * (20 * Nz + 3 * Nq)
        else:                                     # DOTSOCP_FUSED=0: plain projection kernel, beta in + q in + z out
            kname, (proj_ms, proj_n) = "k_cone_march<0> (cone projection)", times["cone_proj"]
            alg_bytes = 8.0 * (20 * Nz + Nq)
        # time-slab mode: the cone pass of an iteration runs as two timed intervals (the chunks in front of the last one, then the
        # last chunk, which alone reads the q halo): the average interval covers half of the slab's cells
        slab_mode = world > 1 or args.nslabs > 1 or bool(share)
        cone_parts = 2 if (slab_mode and ncl >= 12 and os.environ.get("DOTSOCP_OVERLAP", "1") != "0"
                           and args.method in ("inPALM", "ALG2")) else 1
        alg_bytes /= cone_parts
        achieved = alg_bytes / (proj_ms * 1e-3) / 1e9 if proj_ms > 0 else 0.0
        # HBM bytes of the dominant kernel by the PMC counters: a figure of the builder's profiling run of THIS round's build
        # (profiles/cone_proj_traffic.json, separate --pmc FETCH_SIZE / WRITE_SIZE passes), not of this run -- tagged as such,
        # and left out when the record is from another round, grid, method or kernel
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, "profiles", "cone_proj_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if (rec.get("grid") == [ny, nx, nt] and args.method in ("inPALM", "ALG2") and rec.get("round") == BUILD_ROUND
                        and kname.startswith(rec.get("kernel", "?").replace(" ", "")[:14])):
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = (f"profiles/cone_proj_traffic.json (builder's rocprofv3 PMC passes of round {rec.get('round')}, "
                                      f"commit {rec.get('commit', 'n/a')}; not a counter of this run)")
            except Exception:
                traffic = None
        # the other kernels of a plain iteration, so that the line shows the one furthest below the roofline; phase timers
        # (HIP events on the launch stream) with the algorithmic bytes of DESIGN.md section 3; Nphi = nodes of this slab
        Nphi = ny * nx * ntl

        def krow(name, nbytes, key, launches_per_phase=1):
            ms, n = times[key]
            if ms <= 0 or n <= 0:
                return None
            gbs = nbytes / (ms * 1e-3) / 1e9
            return {"name": name, "algorithmic_bytes": nbytes, "avg_ms": round(ms, 4), "launches_per_phase": launches_per_phase,
                    "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}

        kernels = [r for r in (
            krow(kname, alg_bytes, {"acc-ADMM": "acc_cone", "PALM": "cone_fused_b" if times["cone_fused_b"][1] > 0 else "cone_fused_a"}.get(args.method, "cone_fused_b" if times["cone_fused_b"][1] > 0 else "cone_proj")),
            krow("k_qstep_rhs (A phi, q-step, alpha update, next rhs)", 8.0 * (3 * Nphi + 4 * Nq), "qstep"),
            krow("Poisson solve: y, x forward, fused t pass, x, y inverse (five launches; k_dct_* / k_pfa_*)", 8.0 * 10 * Nphi,
                 "poisson", 5),
        ) if r]
        out = {
            "metric": "ADMM iters/sec on NxNxT dot2d staggered grid at 1/2/4/8 MI355X",
            "value": args.steps / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} {ny}x{nx}x{nt} Gaussian-to-Gaussian (Example 5.1), " + {
                           "inPALM": "inPALM tau=1.9", "ALG2": "ALG2 tau=1.0", "PALM": "PALM tau=1.9",
                           "acc-ADMM": "acc-ADMM (Halpern, rho=2, restart=100)"}[args.method] + ", levelN=1",
                       "grid": [ny, nx, nt],
                       "kkt_checks_in_timed_region": int(np.sum(hist["iter"] > args.warmup)),
                       # which kind of box this run landed on: device-to-device copy rate of a 1 GiB buffer measured right
                       # before the warm-up (read + write bytes / time; the pool's two kinds differ by ~10 % in every HBM-bound kernel)
                       "box_copy_gbs": box_copy_gbs,
                       # the timed region runs with the library's per-phase HIP events switched on (they feed roofline and
                       # kernel_ms); DOTSOCP_BENCH_NOPROF=1 times it without them
                       "per_phase_hip_events_in_timed_region": not bool(os.environ.get("DOTSOCP_BENCH_NOPROF")) and not two_pass,
                       "parallelism": (f"rank share: slab {share // 2} of {share} time slabs on 1 GPU, neighbour messages as local "
                                       f"copies (timing only, not a valid solve)") if share else
                                      ("1 GPU" if world == 1 else f"{world} time slabs")},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": proj_ms, "launches": proj_n,
                         # SURVEY.md 8d's narrower figure for the projection alone, 8 (20 Nz + Nq): what the same launch scores
                         # if only beta in, q in and z out are counted (this kernel also reads q^{k-1} and writes the adjoint sums)
                         "frac_projection_only": (8.0 * (20 * Nz + Nq) / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if proj_ms > 0 else 0.0,
                         "kernels": kernels},
            "kernel_ms": {k: round(v[0], 4) for k, v in times.items()},
        }
        if two_pass and not share:
            out["config"]["timed_pass"] = ("without per-phase HIP events (they cost a rank's 1.7 ms iteration 4-5 %); kernel_ms / roofline "
                                           "come from a second, untimed run of the same W + K iterations with them")
        if share:
            out["rank_share"] = {"n": share, "slab": share // 2, "time_nodes": int(ntl),
                                 "kernel_ms_sum": round(sum(v[0] * v[1] for v in times.values()) / max(args.steps, 1), 4),
                                 "full_grid_ms_per_step": full_ms,
                                 "ceiling": (full_ms / (dt / args.steps * 1e3)) if full_ms else None,
                                 "ceiling_note": ("T(full grid) / T(this rank's share), both timed in THIS run (same process, same box, same "
                                                  "W and K, no per-phase events): the compute-only ceiling of the N-GPU strong-scaling curve")
                                 if full_ms else "the full grid does not fit one GPU: the share is timed alone",
                                 "timed_pass": "without per-phase HIP events; kernel_ms / roofline come from a second run of the same W + K "
                                               "iterations with them" if two_pass else "with per-phase HIP events",
                                 "note": "kernel_ms_sum = per-iteration sum of the phase timers (HIP events on the launch streams; "
                                         "phases on the second stream overlap the main one); ms_per_step - overlap-free kernel time = "
                                         "launch / dependency chain + host"}
        if slab_mode:
            out["roofline"]["note"] = (f"time slabs: the cone pass of an iteration is {cone_parts} timed interval(s) (chunks of time cells, "
                                       "one launch each); bytes and time are per interval; the N = 1 line is the kernel's roofline figure")
        if fallback_note:
            out["config"]["timed_pass"] = fallback_note
        if rank == 0:
            if not args.no_cpu_baseline and world == 1 and args.method == "inPALM" and not share:
                out["cpu_baseline"] = cpu_baseline(args.workload, ny, nx, nt, args.cpu_seconds)
            print(json.dumps(out), flush=True)
    phase_keys = ("rhs", "poisson", "cone_fused_a", "cone_fused_b", "cone_proj", "qstep", "beta", "materialise", "kkt", "comm",
                  "interp", "acc_cone", "acc_gather", "qstep_first", "transpose")
    emit_lock, emitted = threading.Lock(), []
    watchdog = None
    if two_pass and dist is not None:
        # several real ranks: should the instrumented pass (a second communicator, the same iterations again) ever hang, the
        # timed pass's result is not lost with it -- after DOTSOCP_BENCH_PASS2_LIMIT seconds rank 0 prints the line without
        # phase timers and every rank leaves
        def bail():
            with emit_lock:
                if not emitted:
                    emitted.append(True)
                    emit({k: (0.0, 0) for k in phase_keys},
                         "the instrumented second pass did not finish in time: kernel_ms and roofline are empty, value is "
                         "the timed pass's (run without per-phase HIP events)")
                    os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("DOTSOCP_BENCH_PASS2_LIMIT", "240")), bail)
        watchdog.daemon = True
        watchdog.start()
    if two_pass:
        # the instrumented pass: the SAME W + K iterations again in a context of its own (its own communicator), per-phase
        # events on during the K -- untimed; `kernel_ms` and `roofline` come from here
        ctx.close()
        ctx = D.InPALMContext(var, opts, model, weighted=weight is not None, device=device, profiling=False,
                              rccl=(fresh_rccl() if dist is not None else rccl), nslabs=args.nslabs, method=args.method)
        assert ctx.run(args.warmup) == args.warmup
        D.capi.check(D.capi.lib().dotsocp_set_profiling(ctx._ctx, 1))
        assert ctx.run(args.steps) == args.steps
        fence()
        ctx.finish(download=False)
    times = {k: ctx.kernel_time(k) for k in phase_keys}
    ctx.close()
    if watchdog is not None:
        watchdog.cancel()
    with emit_lock:
        if not emitted:
            emitted.append(True)
            emit(times)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
