"""Solver-level boundary B1 and the drivers above it, with the reference's names and option
structs:

    [runHist, sigma] = solver_socp_inPALM(var, opts, model)        socp/dot2d|dot1d/algorithms/solver_socp_inPALM.m:1
    [runHist, sigma] = solver_wsocp_inPALM(var, opts, model)       socp/wdot2d/algorithms/solver_wsocp_inPALM.m:1
    [output, timeML, runHistML, runHist] = solver_dotsocp2d(rho0, rho1, nt, levelN, opts, method)
                                                                   socp/dot2d/solver_dotsocp2d.m:1
    ... solver_dotsocp1d (socp/dot1d/solver_dotsocp1d.m:1), solver_wdotsocp2d (socp/wdot2d/solver_wdotsocp2d.m:1)

The whole iteration loop runs on the GPU inside libdotsocp (dot-socp_amd/csrc/solver.hip);
this module only marshals VarHandle / ModelHandle fields across the C ABI.
"""
import ctypes
import time

import numpy as np

from . import capi
from .model import (InitialScaling, ModelHandle, VarHandle, check_massConservation, initialize,
                    recover_q, recover_RhoE, recoverOrgVar)

TIME_NAMES = ['Step_1_1_FFT', 'Step_1_2_ProjSOC', 'Step_2_Q_Step', 'Step_3_Multiplier', 'KKT',
              'Total_Time', 'Iters']
# solver_socp_accADMM.m:438-439 (the weighted file lists 'Step_4_Interp' before 'KKT', :443-444)
ACC_TIME_NAMES = ['Step_1_Q_Step', 'Step_2_Multiplier', 'Step_3_1_FFT', 'Step_3_2_ProjSOC', 'KKT', 'Interp',
                  'Total_Time', 'Iters']
WACC_TIME_NAMES = ['Step_1_Q_Step', 'Step_2_Multiplier', 'Step_3_1_FFT', 'Step_3_2_ProjSOC', 'Step_4_Interp', 'KKT',
                   'Total_Time', 'Iters']
PALM_TIME_NAMES = ['Step_1_Q_Step', 'Step_2_1_FFT', 'Step_2_2_ProjSOC', 'Step_3_Q_Step', 'Step_4_Multiplier', 'KKT',
                   'Total_Time', 'Iters']      # solver_socp_PALM.m:351-352
METHODS = {"inPALM": capi.METHOD_INPALM, "ALG2": capi.METHOD_INPALM, "PALM": capi.METHOD_PALM,
           "acc-ADMM": capi.METHOD_ACCADMM}


def _get(opts, name, default=None):
    if isinstance(opts, dict):
        return opts.get(name, default)
    return getattr(opts, name, default)


def _has(opts, name):
    return (name in opts) if isinstance(opts, dict) else hasattr(opts, name)


class InPALMContext:
    """Stateful handle on one device-resident loop (create -> upload -> begin -> run* -> finish)."""

    def __init__(self, var, opts, model, weighted=False, device=0, nslabs=1, profiling=False, rccl=None,
                 method="inPALM", warm_from=None, ngpu=None, z_unread=False):
        """rccl = (unique_id_bytes, rank, world): one process per GPU, this process owns time slab
        `rank`; var / model then hold the LOCAL slab of every field (model.nt stays the global nt).
        method: which loop file of the reference runs ("inPALM"/"ALG2" by opts.tau, "PALM", "acc-ADMM").
        warm_from: the finished context of the previous (coarser) multilevel level: phi, q, alpha, z, beta are
        then produced on the device by jump_nextLevel.m's transfer instead of being uploaded from `var`.
        ngpu: single-process multi-GPU (dotsocp_create_multi): that many time slabs, slab r on device
        (device + r) mod #devices; nslabs (diagnostic) keeps all slabs on `device`.
        z_unread: the caller will run at least one iteration of the inPALM / ALG2 loop, which overwrites z before its first
        use (solver_socp_inPALM.m:199; the rescale block, the only other reader, needs it >= 10): var.z is not uploaded --
        10 of the 27 N doubles of the state (tests/test_gpu_solver.py::test_iterations_from_a_random_state)."""
        L = capi.lib()
        self.method = method
        one_d = not hasattr(model, "ny")
        p = capi.Problem()
        p.dim = 1 if one_d else 2
        p.weighted = 1 if weighted else 0
        p.ny = 1 if one_d else int(model.ny)
        p.nx, p.nt = int(model.nx), int(model.nt)
        p.D, p.E, p.cScale, p.dScale = float(var.D), float(var.E), float(var.cScale), float(var.dScale)
        p.normc, p.normd = float(model.normc), float(model.normd)
        self.var, self.model, self.weighted = var, model, weighted
        if ngpu is not None and int(ngpu) > 1:
            self._ctx = L.dotsocp_create_multi(ctypes.byref(p), int(device), int(ngpu))
        else:
            self._ctx = L.dotsocp_create(ctypes.byref(p), int(device), int(nslabs))
        if not self._ctx:
            raise capi.DotsocpError(-1, L.dotsocp_last_error().decode())
        try:
            if rccl is not None:
                uid, rk, wd = rccl
                buf = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(uid))
                capi.check(L.dotsocp_attach_rccl(self._ctx, buf, int(rk), int(wd)))
            # a field left as None keeps the device default (zeros), e.g. z, beta, q, alpha of a cold start
            skip_z = z_unread and METHODS[method] == capi.METHOD_INPALM and int(_get(opts, "maxit")) >= 1
            state = () if warm_from is not None else ((capi.F_PHI, var.phi), (capi.F_Q, var.q), (capi.F_ALPHA, var.alpha),
                                                      (capi.F_Z, None if skip_z else var.z), (capi.F_BETA, var.beta))
            for f, a in state:
                if a is not None:
                    self.upload(f, a)
            ends = getattr(model, "_c_ends", None)
            if ends is not None and model.c.size > 2 * ends and rccl is None:
                # initialize(lazy_zeros=True): c is zero between its first and last layer, as the device array is
                nlay = model.c.size // ends
                for t, part in ((0, model.c[:ends]), (nlay - 1, model.c[model.c.size - ends:])):
                    capi.check(L.dotsocp_upload_layers(self._ctx, capi.F_C, capi.fptr(np.ascontiguousarray(part)), t, 1))
            else:
                self.upload(capi.F_C, model.c)
            if weighted:
                self.upload(capi.F_WEIGHT, model.weight)
            if warm_from is not None:
                capi.check(L.dotsocp_jump_next_level(warm_from._ctx, self._ctx))
            if profiling:
                capi.check(L.dotsocp_set_profiling(self._ctx, 1))
            o = capi.Opts()
            # required fields (solver_socp_inPALM.m:33-37)
            o.tau = float(_get(opts, "tau", 1.0) if method == "acc-ADMM" else _get(opts, "tau"))
            o.sigma, o.tol = float(_get(opts, "sigma")), float(_get(opts, "tol"))
            o.maxit = int(_get(opts, "maxit"))
            o.ifCheckStepByStep = int(bool(_get(opts, "ifCheckStepByStep", False)))
            # optional fields (:20-30,64-68)
            o.checkPrimDualFeas = int(bool(_get(opts, "checkPrimDualFeas"))) if _has(opts, "checkPrimDualFeas") else -1
            o.scaling = int(bool(_get(opts, "scaling", False)))
            o.time_limit = float(_get(opts, "time_limit", 3600))
            if METHODS[method] == capi.METHOD_INPALM:
                capi.check(L.dotsocp_begin(self._ctx, ctypes.byref(o)))
            else:
                a = capi.AccOpts()                  # solver_socp_accADMM.m:12-28; 0 = reference default
                a.restart = int(_get(opts, "restart", 0) or 0)
                a.rho = float(_get(opts, "rho", 0) or 0)
                a.theta = float(_get(opts, "theta", 0) or 0)
                capi.check(L.dotsocp_begin_method(self._ctx, ctypes.byref(o), METHODS[method], ctypes.byref(a)))
        except Exception:
            self.close()
            raise

    def upload(self, field, arr):
        a = np.asfortranarray(arr, dtype=np.float64)
        capi.check(capi.lib().dotsocp_upload(self._ctx, field, capi.fptr(a)))

    def download(self, field, like):
        out = np.empty(like.shape, dtype=np.float64, order="F")
        capi.check(capi.lib().dotsocp_download(self._ctx, field, capi.fptr(out)))
        return out

    def run(self, n_iters=-1):
        done = capi.i64()
        capi.check(capi.lib().dotsocp_run(self._ctx, int(n_iters), ctypes.byref(done)))
        return done.value

    def synchronize(self):
        capi.check(capi.lib().dotsocp_synchronize(self._ctx))

    def kernel_time(self, name):
        ms, n = capi.dbl(), capi.i64()
        capi.check(capi.lib().dotsocp_kernel_time(self._ctx, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def finish(self, download=True):
        """solver_socp_inPALM.m:329-357: write iterates and scaling factors back into `var`."""
        L = capi.lib()
        res = capi.Result()
        capi.check(L.dotsocp_finish(self._ctx, ctypes.byref(res)))
        var = self.var
        var.name = {"acc-ADMM": 'Accelerated ADMM', "PALM": 'Proximal ALM'}.get(self.method, 'Inexact Proximal ALM')
        if download:
            var.phi = self.download(capi.F_PHI, var.phi)
            var.q = self.download(capi.F_Q, var.q)
            var.z = self.download(capi.F_Z, var.z)
            var.alpha = self.download(capi.F_ALPHA, var.alpha)       # = sigma * alpha
            var.beta = self.download(capi.F_BETA, var.beta)          # = sigma * beta
        t = list(res.times)
        if self.method == "acc-ADMM":
            tm = dict(Step_1_Q_Step=t[2], Step_2_Multiplier=t[3], Step_3_1_FFT=t[0], Step_3_2_ProjSOC=t[1], KKT=t[4],
                      Interp=res.time_extra, Step_4_Interp=res.time_extra, Total_Time=t[5], Iters=t[6])
            var.time = {k: tm[k] for k in (WACC_TIME_NAMES if self.weighted else ACC_TIME_NAMES)}
        elif self.method == "PALM":
            var.time = dict(zip(PALM_TIME_NAMES, [res.time_extra] + t))
        else:
            var.time = dict(zip(TIME_NAMES, t))
        var.cScale, var.dScale = res.cScale, res.dScale
        n = int(res.hist_len)
        kkt = np.empty((n, 7), order="F")
        tm, itr, gap = np.empty(n), np.empty(n), np.empty(n)
        capi.check(L.dotsocp_get_history(self._ctx, capi.fptr(kkt) if n else None, capi.fptr(tm) if n else None,
                                         capi.fptr(itr) if n else None, capi.fptr(gap) if n else None))
        runHist = dict(kkt=kkt, time=tm, iter=itr, pdGap=gap, len=n)
        self.result = res
        return runHist, res.sigma

    def outputs(self):
        """solver_dotsocp2d.m:262-281 on the device (recoverOrgVar + recover_RhoE + recover_q): dict with rho, Ex,
        [Ey,] q0, bx[, by]; call after finish()."""
        m = self.model
        one_d = not hasattr(m, "ny")
        nt = int(m.nt)
        shp = (int(m.nx),) if one_d else (int(m.ny), int(m.nx))
        names = ("rho", "Ex", "q0", "bx") if one_d else ("rho", "Ex", "Ey", "q0", "bx", "by")
        out = {k: np.empty(shp + ((nt - 1,) if k in ("q0", "bx", "by") else (nt,)), order="F") for k in names}
        r0 = np.asfortranarray(m.rho0, dtype=np.float64)
        r1 = np.asfortranarray(m.rho1, dtype=np.float64)
        ptr = lambda k: capi.fptr(out[k]) if k in out else None      # noqa: E731
        capi.check(capi.lib().dotsocp_recover_outputs(self._ctx, capi.fptr(r0), capi.fptr(r1), ptr("rho"), ptr("Ex"),
                                                      ptr("Ey"), ptr("q0"), ptr("bx"), ptr("by")))
        return out

    def close(self):
        if getattr(self, "_ctx", None):
            capi.lib().dotsocp_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()


def solver_socp_inPALM(var, opts, model, device=0, nslabs=1):
    """[runHist, sigma] = solver_socp_inPALM(var, opts, model); `var` is mutated in place.
    nslabs > 1 runs the multi-GPU time-slab algorithm with all slabs on this one device."""
    ctx = InPALMContext(var, opts, model, weighted=False, device=device, nslabs=nslabs, z_unread=True)
    try:
        ctx.run(-1)
        return ctx.finish()
    finally:
        ctx.close()


def solver_wsocp_inPALM(var, opts, model, device=0, nslabs=1):
    """[runHist, sigma] = solver_wsocp_inPALM(var, opts, model) (model.weight required)."""
    ctx = InPALMContext(var, opts, model, weighted=True, device=device, nslabs=nslabs, z_unread=True)
    try:
        ctx.run(-1)
        return ctx.finish()
    finally:
        ctx.close()


def solver_socp_PALM(var, opts, model, device=0, nslabs=1):
    """[runHist, sigma] = solver_socp_PALM(var, opts, model)   socp/dot2d/algorithms/solver_socp_PALM.m:1
    nslabs > 1: the time-slab algorithm with all slabs on this one device (as for solver_socp_inPALM)."""
    ctx = InPALMContext(var, opts, model, weighted=False, device=device, method="PALM", nslabs=nslabs)
    try:
        ctx.run(-1)
        return ctx.finish()
    finally:
        ctx.close()


def solver_socp_accADMM(var, opts, model, device=0, nslabs=1):
    """[runHist, sigma] = solver_socp_accADMM(var, opts, model)   socp/dot2d/algorithms/solver_socp_accADMM.m:1
    opts: sigma, maxit, tol, ifCheckStepByStep (+ restart, rho, theta, checkPrimDualFeas, time_limit, scaling)."""
    ctx = InPALMContext(var, opts, model, weighted=False, device=device, method="acc-ADMM", nslabs=nslabs)
    try:
        ctx.run(-1)
        return ctx.finish()
    finally:
        ctx.close()


def solver_wsocp_accADMM(var, opts, model, device=0, nslabs=1):
    """socp/wdot2d/algorithms/solver_wsocp_accADMM.m:1"""
    ctx = InPALMContext(var, opts, model, weighted=True, device=device, method="acc-ADMM", nslabs=nslabs)
    try:
        ctx.run(-1)
        return ctx.finish()
    finally:
        ctx.close()


# --------------------------------------------------------------------------------------
# drivers
# --------------------------------------------------------------------------------------
def _driver_opts(opts, method, weighted, dim=2):
    """solver_dotsocp2d.m:76-151 / solver_dotsocp1d.m / solver_wdotsocp2d.m:85-162."""
    ok = method in ("inPALM", "ALG2") or (method == "acc-ADMM" and dim == 2) or (
        method == "PALM" and dim == 2 and not weighted)                   # solver_dotsocp2d.m:205-226
    if not ok:
        raise ValueError("Invalid input at position 6 (Solving method)")
    o = dict(opts) if isinstance(opts, dict) else dict(vars(opts))
    o.setdefault("ifCheckStepByStep", False)
    o.setdefault("scaling", True)
    o.setdefault("maxit", 10000 if weighted else 3000)
    if method != "acc-ADMM":
        o["tau"] = 1.0 if method == "ALG2" else 1.9                       # :133-137
    o.setdefault("sigma", 1.0)
    o.setdefault("time_limit", 3600)
    return o


def _solve_levels(rho0, rho1, nt, levelN, opts, method, dim, weighted, device, barrier=None, transfer="device"):
    """The level loop of solver_dotsocp2d.m:154-250 (dot1d / wdot2d twins): restrict the data to
    levelN grids, solve coarse to fine with warm starts; every solve runs on the device.
    transfer = "device": the state never leaves the GPU -- jump_nextLevel and the output recovery run there
    (dotsocp_jump_next_level / dotsocp_recover_outputs); "host": download, numpy twins of jump_nextLevel.m /
    recover_RhoE.m / recover_q.m, upload (the two agree to rounding, tests/test_multilevel.py)."""
    from . import multilevel as ML
    from .examples import ensure_barrier_validity
    if not (isinstance(levelN, (int, np.integer)) and levelN >= 1):
        raise ValueError("Invalid input at position 4 (Number of levels in multilevel strategy)")
    if transfer not in ("device", "host"):
        raise ValueError("transfer must be 'device' or 'host'")
    o = _driver_opts(opts, method, weighted, dim)
    t_all = time.perf_counter()
    tolFactor = -1.0 if o["tol"] > 0.99e-3 else -0.5                     # :124-128
    tolLB = 1e-4 if dim == 2 else 1e-5                                   # :130, solver_dotsocp1d.m:121
    L = int(levelN)
    rho0s, rho1s, nts, tols, ws = [None] * L, [None] * L, [None] * L, [None] * L, [None] * L
    rho0s[-1], rho1s[-1] = np.asarray(rho0, dtype=np.float64), np.asarray(rho1, dtype=np.float64)
    nts[-1], tols[-1] = int(nt), o["tol"]
    if weighted:
        ws[-1] = np.asarray(_get(opts, "weight"), dtype=np.float64)
    for lv in range(L - 2, -1, -1):                                      # :166-178
        if (nts[lv + 1] - 1) % 2 or any((n - 1) % 2 for n in rho0s[lv + 1].shape):
            raise ValueError("multilevel needs 2^k*m+1 grid sizes on every level (solver_dotsocp2d.m:167)")
        nts[lv] = (nts[lv + 1] - 1) // 2 + 1
        tols[lv] = max(tols[lv + 1] * 2 ** tolFactor, tolLB)
        rho0s[lv], rho1s[lv] = ML.downSample_phi(rho0s[lv + 1]), ML.downSample_phi(rho1s[lv + 1])
        if weighted:
            nyf, nxf = rho0s[lv + 1].shape
            if barrier is not None:                                      # solver_wdotsocp2d.m:186-189
                rho0s[lv], rho1s[lv], _ = ensure_barrier_validity(rho0s[lv], rho1s[lv], barrier)
                ws[lv] = ML.downSample_barrier(nts[lv + 1], nxf, nyf, ws[lv + 1])
                continue
            ws[lv] = ML.downSample_q(nts[lv + 1], nxf, nyf, ws[lv + 1])
        N = rho0s[lv].size
        rho0s[lv] = rho0s[lv] / (rho0s[lv].sum() / N)
        rho1s[lv] = rho1s[lv] / (rho1s[lv].sum() / N)
    on_device = transfer == "device"
    # opts.ngpu (extension, as in the MEX gateway): every level is cut into time slabs on that many devices of this one
    # process (a level with few time nodes gets fewer slabs: at least two nodes per slab)
    ngpu = int(_get(opts, "ngpu", 1) or 1)
    var, model = initialize(rho0s[0], rho1s[0], nts[0], lazy_zeros=on_device)
    if weighted:
        model.weight = ws[0]
    timeML, runHistML, runHist, last = [], None, None, None
    ctx = prev = None
    try:
        for lv in range(L):
            InitialScaling(var, model, o["scaling"], last, dim=dim, weighted=weighted)
            ctx = InPALMContext(var, dict(o, tol=tols[lv]), model, weighted=weighted, device=device, method=method,
                                warm_from=prev, ngpu=max(1, min(ngpu, nts[lv] // 2)))
            if prev is not None:
                prev.close()
                prev = None
            ctx.run(-1)
            last_level = lv == L - 1
            runHist, sigma = ctx.finish(download=not on_device)
            timeML.append(var.time)
            if runHistML is None:                                        # catRunHist, :389-407
                runHistML = {k: np.array(v, copy=True) if isinstance(v, np.ndarray) else v for k, v in runHist.items()}
            else:
                runHist["time"] = runHistML["time"][-1] + runHist["time"]
                runHistML["kkt"] = np.concatenate([runHistML["kkt"], runHist["kkt"]], axis=0)
                runHistML["pdGap"] = np.concatenate([runHistML["pdGap"], runHist["pdGap"]])
                runHistML["time"] = np.concatenate([runHistML["time"], runHist["time"]])
                runHistML["iter"] = np.concatenate([runHistML["iter"], runHistML["iter"][-1] + runHist["iter"]])
                runHistML["len"] = runHistML["len"] + runHist["len"]
            if on_device and last_level:
                output = ctx.outputs()
            if not on_device:
                ctx.close()
                recoverOrgVar(var)
            if not last_level:
                o["time_limit"] = o["time_limit"] - var.time["Total_Time"]   # :244
                o["sigma"] = 10 ** (np.log10(o["sigma"] * sigma) / 2)         # :245
                last = runHist["kkt"][-1]
                wf = ws[lv + 1] if weighted else None
                if on_device:
                    E2 = var.E2
                    var, model = initialize(rho0s[lv + 1], rho1s[lv + 1], nts[lv + 1], lazy_zeros=True, phi=False)
                    var.E2 = E2                         # the state comes from the coarse level on the device
                    if weighted:
                        model.weight = wf
                    prev = ctx
                else:
                    var, model = ML.jump_nextLevel(var, model, rho0s[lv + 1], rho1s[lv + 1], nts[lv + 1], wf)
        if not on_device:
            rho_E = recover_RhoE(var, model, weighted=weighted)
            qs = recover_q(var, model)
            if dim == 2:
                output = dict(rho=rho_E[0], Ex=rho_E[1], Ey=rho_E[2], q0=qs[0], bx=qs[1], by=qs[2])
            else:
                output = dict(rho=rho_E[0], Ex=rho_E[1], q0=qs[0], bx=qs[1])
    finally:
        for c in (ctx, prev):
            if c is not None:
                c.close()
    timeML.append({"ML_Time": time.perf_counter() - t_all})
    name = ("Weighted-" if weighted else "") + "DOT-SOCP"
    mname = (f"{method} for {name}") if L == 1 else (f"Multilevel-{method} for {name}")
    runHist["method"] = runHistML["method"] = mname
    return output, timeML, runHistML, runHist


def solver_dotsocp2d(rho0, rho1, nt, levelN, opts, method="inPALM", device=0, transfer="device"):
    """[output, timeML, runHistML, runHist] = solver_dotsocp2d(rho0, rho1, nt, levelN, opts, method)
    socp/dot2d/solver_dotsocp2d.m:1"""
    output, timeML, runHistML, runHist = _solve_levels(rho0, rho1, nt, levelN, opts, method, 2, False, device,
                                                       transfer=transfer)
    if not check_massConservation(output["rho"], 1e-2):
        print("Warning: The mass conservation constraint violation exceeds 0.01")
    return output, timeML, runHistML, runHist


def solver_dotsocp1d(rho0, rho1, nt, levelN, opts, method="inPALM", device=0, transfer="device"):
    """socp/dot1d/solver_dotsocp1d.m:1"""
    output, timeML, runHistML, runHist = _solve_levels(rho0, rho1, nt, levelN, opts, method, 1, False, device,
                                                       transfer=transfer)
    if not check_massConservation(output["rho"], 1e-2):
        print("Warning: The mass conservation constraint violation exceeds 0.01")
    return output, timeML, runHistML, runHist


def solver_wdotsocp2d(rho0, rho1, nt, levelN, opts, method="inPALM", barrier=None, device=0, transfer="device"):
    """socp/wdot2d/solver_wdotsocp2d.m:1"""
    output, timeML, runHistML, runHist = _solve_levels(rho0, rho1, nt, levelN, opts, method, 2, True, device,
                                                       barrier=barrier, transfer=transfer)
    if not check_massConservation(output["rho"], 1e-2):
        print("Warning: The tolerance of mass conservation constraint is under 0.01")
    return output, timeML, runHistML, runHist
