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


def _get(opts, name, default=None):
    if isinstance(opts, dict):
        return opts.get(name, default)
    return getattr(opts, name, default)


def _has(opts, name):
    return (name in opts) if isinstance(opts, dict) else hasattr(opts, name)


class InPALMContext:
    """Stateful handle on one device-resident loop (create -> upload -> begin -> run* -> finish)."""

    def __init__(self, var, opts, model, weighted=False, device=0, nslabs=1, profiling=False, rccl=None):
        """rccl = (unique_id_bytes, rank, world): one process per GPU, this process owns time slab
        `rank`; var / model then hold the LOCAL slab of every field (model.nt stays the global nt)."""
        L = capi.lib()
        one_d = not hasattr(model, "ny")
        p = capi.Problem()
        p.dim = 1 if one_d else 2
        p.weighted = 1 if weighted else 0
        p.ny = 1 if one_d else int(model.ny)
        p.nx, p.nt = int(model.nx), int(model.nt)
        p.D, p.E, p.cScale, p.dScale = float(var.D), float(var.E), float(var.cScale), float(var.dScale)
        p.normc, p.normd = float(model.normc), float(model.normd)
        self.var, self.model, self.weighted = var, model, weighted
        self._ctx = L.dotsocp_create(ctypes.byref(p), int(device), int(nslabs))
        if not self._ctx:
            raise capi.DotsocpError(-1, L.dotsocp_last_error().decode())
        try:
            if rccl is not None:
                uid, rk, wd = rccl
                buf = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(uid))
                capi.check(L.dotsocp_attach_rccl(self._ctx, buf, int(rk), int(wd)))
            # a field left as None keeps the device default (zeros), e.g. z, beta, q, alpha of a cold start
            for f, a in ((capi.F_PHI, var.phi), (capi.F_Q, var.q), (capi.F_ALPHA, var.alpha),
                         (capi.F_Z, var.z), (capi.F_BETA, var.beta), (capi.F_C, model.c)):
                if a is not None:
                    self.upload(f, a)
            if weighted:
                self.upload(capi.F_WEIGHT, model.weight)
            if profiling:
                capi.check(L.dotsocp_set_profiling(self._ctx, 1))
            o = capi.Opts()
            # required fields (solver_socp_inPALM.m:33-37)
            o.tau, o.sigma, o.tol = float(_get(opts, "tau")), float(_get(opts, "sigma")), float(_get(opts, "tol"))
            o.maxit = int(_get(opts, "maxit"))
            o.ifCheckStepByStep = int(bool(_get(opts, "ifCheckStepByStep", False)))
            # optional fields (:20-30,64-68)
            o.checkPrimDualFeas = int(bool(_get(opts, "checkPrimDualFeas"))) if _has(opts, "checkPrimDualFeas") else -1
            o.scaling = int(bool(_get(opts, "scaling", False)))
            o.time_limit = float(_get(opts, "time_limit", 3600))
            capi.check(L.dotsocp_begin(self._ctx, ctypes.byref(o)))
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
        var.name = 'Inexact Proximal ALM'
        if download:
            var.phi = self.download(capi.F_PHI, var.phi)
            var.q = self.download(capi.F_Q, var.q)
            var.z = self.download(capi.F_Z, var.z)
            var.alpha = self.download(capi.F_ALPHA, var.alpha)       # = sigma * alpha
            var.beta = self.download(capi.F_BETA, var.beta)          # = sigma * beta
        var.time = dict(zip(TIME_NAMES, list(res.times)))
        var.cScale, var.dScale = res.cScale, res.dScale
        n = int(res.hist_len)
        kkt = np.empty((n, 7), order="F")
        tm, itr, gap = np.empty(n), np.empty(n), np.empty(n)
        capi.check(L.dotsocp_get_history(self._ctx, capi.fptr(kkt) if n else None, capi.fptr(tm) if n else None,
                                         capi.fptr(itr) if n else None, capi.fptr(gap) if n else None))
        runHist = dict(kkt=kkt, time=tm, iter=itr, pdGap=gap, len=n)
        self.result = res
        return runHist, res.sigma

    def close(self):
        if getattr(self, "_ctx", None):
            capi.lib().dotsocp_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()


def solver_socp_inPALM(var, opts, model, device=0, nslabs=1):
    """[runHist, sigma] = solver_socp_inPALM(var, opts, model); `var` is mutated in place.
    nslabs > 1 runs the multi-GPU time-slab algorithm with all slabs on this one device."""
    ctx = InPALMContext(var, opts, model, weighted=False, device=device, nslabs=nslabs)
    try:
        ctx.run(-1)
        return ctx.finish()
    finally:
        ctx.close()


def solver_wsocp_inPALM(var, opts, model, device=0, nslabs=1):
    """[runHist, sigma] = solver_wsocp_inPALM(var, opts, model) (model.weight required)."""
    ctx = InPALMContext(var, opts, model, weighted=True, device=device, nslabs=nslabs)
    try:
        ctx.run(-1)
        return ctx.finish()
    finally:
        ctx.close()


# --------------------------------------------------------------------------------------
# drivers (levelN = 1; the multilevel transfer of jump_nextLevel.m is a "next" row)
# --------------------------------------------------------------------------------------
def _driver_opts(opts, method, weighted):
    """solver_dotsocp2d.m:76-151 / solver_dotsocp1d.m / solver_wdotsocp2d.m:85-162."""
    if method not in ("inPALM", "ALG2"):
        raise ValueError("Invalid input at position 6 (Solving method)")
    o = dict(opts) if isinstance(opts, dict) else dict(vars(opts))
    o.setdefault("ifCheckStepByStep", False)
    o.setdefault("scaling", True)
    o.setdefault("maxit", 10000 if weighted else 3000)
    o["tau"] = 1.9 if method == "inPALM" else 1.0
    o.setdefault("sigma", 1.0)
    o.setdefault("time_limit", 3600)
    return o


def _solve_levels(rho0, rho1, nt, levelN, opts, method, dim, weighted, device):
    if not (isinstance(levelN, (int, np.integer)) and levelN >= 1):
        raise ValueError("Invalid input at position 4 (Number of levels in multilevel strategy)")
    if levelN != 1:
        raise NotImplementedError("levelN > 1 (jump_nextLevel.m) is not part of this round; use levelN = 1")
    o = _driver_opts(opts, method, weighted)
    t0 = time.perf_counter()
    var, model = initialize(rho0, rho1, nt)
    if weighted:
        model.weight = np.asarray(_get(opts, "weight"), dtype=np.float64)
    InitialScaling(var, model, o["scaling"], None, dim=dim, weighted=weighted)
    solve = solver_wsocp_inPALM if weighted else solver_socp_inPALM
    runHist, sigma = solve(var, o, model, device=device)
    recoverOrgVar(var)
    ml_time = time.perf_counter() - t0
    timeML = [var.time, {"ML_Time": ml_time}]
    runHistML = dict(runHist)
    name = ("Weighted-" if weighted else "") + "DOT-SOCP"
    runHist["method"] = runHistML["method"] = f"{method} for {name}"
    return var, model, timeML, runHistML, runHist


def solver_dotsocp2d(rho0, rho1, nt, levelN, opts, method="inPALM", device=0):
    var, model, timeML, runHistML, runHist = _solve_levels(rho0, rho1, nt, levelN, opts, method, 2, False, device)
    rho, Ex, Ey = recover_RhoE(var, model)
    q0, bx, by = recover_q(var, model)
    output = dict(rho=rho, Ex=Ex, Ey=Ey, q0=q0, bx=bx, by=by)
    if not check_massConservation(rho, 1e-2):
        print("Warning: The mass conservation constraint violation exceeds 0.01")
    return output, timeML, runHistML, runHist


def solver_dotsocp1d(rho0, rho1, nt, levelN, opts, method="inPALM", device=0):
    var, model, timeML, runHistML, runHist = _solve_levels(rho0, rho1, nt, levelN, opts, method, 1, False, device)
    rho, Ex = recover_RhoE(var, model)
    q0, bx = recover_q(var, model)
    output = dict(rho=rho, Ex=Ex, q0=q0, bx=bx)
    if not check_massConservation(rho, 1e-2):
        print("Warning: The mass conservation constraint violation exceeds 0.01")
    return output, timeML, runHistML, runHist


def solver_wdotsocp2d(rho0, rho1, nt, levelN, opts, method="inPALM", barrier=None, device=0):
    var, model, timeML, runHistML, runHist = _solve_levels(rho0, rho1, nt, levelN, opts, method, 2, True, device)
    rho, Ex, Ey = recover_RhoE(var, model, weighted=True)
    q0, bx, by = recover_q(var, model)
    output = dict(rho=rho, Ex=Ex, Ey=Ey, q0=q0, bx=bx, by=by)
    if not check_massConservation(rho, 1e-2):
        print("Warning: The tolerance of mass conservation constraint is under 0.01")
    return output, timeML, runHistML, runHist
