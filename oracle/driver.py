"""Single-level restatement of the reference drivers (the callers of the hot path).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  socp/dot2d/solver_dotsocp2d.m, socp/dot1d/solver_dotsocp1d.m,
  socp/wdot2d/solver_wdotsocp2d.m
for option defaults, InitialScaling, recoverOrgVar and output recovery.  The
multilevel transfer (jump_nextLevel.m) is restated in oracle/multilevel.py.
"""
import numpy as np

from .inpalm import InPALMState
from .model import initialize, normL2


def default_opts(opts, method="inPALM", weighted=False):
    """solver_dotsocp2d.m:76-151 (solver_wdotsocp2d.m:85-162)."""
    o = dict(opts)
    o.setdefault("ifCheckStepByStep", False)
    o.setdefault("scaling", True)
    o.setdefault("maxit", 10000 if weighted else 3000)
    if method in ("inPALM", "PALM"):     # :133-137
        o["tau"] = 1.9
    elif method == "ALG2":
        o["tau"] = 1.0
    elif method != "acc-ADMM":           # acc-ADMM reads no tau (solver_socp_accADMM.m:12-60)
        raise ValueError("Invalid input at position 6 (Solving method)")
    o.setdefault("sigma", 1.0)          # `isfield(opts,"scaling")` always holds after :81-86
    o.setdefault("time_limit", 3600)
    return o


def InitialScaling(var, model, scalingYes, lastLevelKKT=None, dim=2, weighted=False):
    """solver_dotsocp2d.m:304-365; 1-D hMean = h^(1/2) (solver_dotsocp1d.m:265);
    weighted `adjust` and E2 safeguard (solver_wdotsocp2d.m:297-343)."""
    h = 1.0 / var.phi.size
    hMean = h ** (1.0 / 3.0) if dim == 2 else h ** 0.5
    if lastLevelKKT is None or not hasattr(var, "E2"):
        Escale2 = np.sqrt(2.0)
    elif weighted:
        Escale2 = var.E2 * min(4.0, max(0.25, np.sqrt(lastLevelKKT[0] / lastLevelKKT[1])))
    else:
        ratio = np.sqrt(lastLevelKKT[0] / lastLevelKKT[1])
        if ratio < 0.8333:
            Escale2 = var.E2 * max(1 / np.sqrt(2.0), ratio / 0.8333)
        else:
            Escale2 = var.E2 * min(np.sqrt(2.0), max(1.0, ratio))
    if scalingYes:
        norm_c = normL2(model.c, h) * np.sqrt(model.nt)
        norm_d = np.sqrt(2.0)
        adjust = 1.0
        if weighted:
            adjust = 10.0 ** np.mean(np.log10(model.weight + 1e-10))
        D = np.sqrt(2.0) * np.sqrt(hMean) * adjust
        E = D / Escale2
        cScale = max(1.0, norm_c * np.sqrt(hMean) / adjust)
        dScale = E * norm_d * np.sqrt(adjust)
        model.normc = norm_c / cScale
        model.normd = norm_d * E / dScale
        model.c = (1.0 / cScale) * model.c
        model.grad = D * model.grad
        var.phi = (1.0 / dScale) * var.phi
        var.q = (D / dScale) * var.q
        var.z = (E / dScale) * var.z
        var.alpha = (1.0 / cScale / D) * var.alpha
        var.beta = (1.0 / cScale / E) * var.beta
    else:
        cScale = dScale = D = E = 1.0
        model.normc = normL2(model.c, h)
        model.normd = np.sqrt(2.0)
    var.cScale, var.dScale, var.D, var.E, var.E2 = cScale, dScale, D, E, Escale2


def recoverOrgVar(var):
    """solver_dotsocp2d.m:368-386"""
    cScale, dScale, D, E = var.cScale, var.dScale, var.D, var.E
    var.phi = dScale * var.phi
    var.z = (dScale / E) * var.z
    var.q = (dScale / D) * var.q
    var.alpha = (cScale * D) * var.alpha
    var.beta = (cScale * E) * var.beta


def recover_RhoE(var, model, weighted=False):
    """socp/dot2d/utils/recover_RhoE.m:14-25 (wdot2d: alpha = weight .* alpha, :11)."""
    ny, nx, nt = model.ny, model.nx, model.nt
    qInd = var.qInd
    alpha = var.alpha if not weighted else model.weight * var.alpha
    rho = alpha[:qInd.bx].reshape((ny, nx, nt - 1), order="F")
    rho = np.concatenate([model.rho0[:, :, None], (rho[:, :, :-1] + rho[:, :, 1:]) / 2,
                          model.rho1[:, :, None]], axis=2)
    Ex = alpha[qInd.bx:qInd.by].reshape((ny, nx - 1, nt), order="F").copy()
    Ex[:, :, [0, -1]] *= 2
    Ex = np.concatenate([np.zeros((ny, 1, nt)), (Ex[:, :-1, :] + Ex[:, 1:, :]) / 2, np.zeros((ny, 1, nt))], axis=1)
    Ey = alpha[qInd.by:].reshape((ny - 1, nx, nt), order="F").copy()
    Ey[:, :, [0, -1]] *= 2
    Ey = np.concatenate([np.zeros((1, nx, nt)), (Ey[:-1] + Ey[1:]) / 2, np.zeros((1, nx, nt))], axis=0)
    return rho, Ex, Ey


def recover_RhoE_1d(var, model):
    """socp/dot1d/utils/recover_RhoE.m (same averaging, one space axis)."""
    nx, nt = model.nx, model.nt
    alpha = var.alpha
    nb = var.qInd.bx
    rho = alpha[:nb].reshape((nx, nt - 1), order="F")
    rho = np.concatenate([model.rho0.reshape(nx, 1), (rho[:, :-1] + rho[:, 1:]) / 2,
                          model.rho1.reshape(nx, 1)], axis=1)
    Ex = alpha[nb:].reshape((nx - 1, nt), order="F").copy()
    Ex[:, [0, -1]] *= 2
    Ex = np.concatenate([np.zeros((1, nt)), (Ex[:-1] + Ex[1:]) / 2, np.zeros((1, nt))], axis=0)
    return rho, Ex


def check_massConservation(rho, tol=1e-2):
    """socp/dot2d/utils/check_massConservation.m:16-34 with integralL2 = mean per layer."""
    nt = rho.shape[-1]
    rho2 = rho.reshape((-1, nt), order="F")
    sumRho = rho2.mean(axis=0)
    sumNega = np.where(rho2 < 0, rho2, 0.0).mean(axis=0)
    err = max(np.max(np.abs(sumRho - 1)), np.max(np.abs(sumNega)))
    return err <= tol, sumRho, sumNega


def make_level(rho0, rho1, nt, opts, method="inPALM", weight=None):
    """One level of solver_dotsocp2d.m:187-192: initialize + InitialScaling.
    Returns (var, model, optsML) ready for solver_socp_inPALM."""
    weighted = weight is not None
    o = default_opts(opts, method, weighted)
    var, model = initialize(rho0, rho1, nt)
    if weighted:
        model.weight = np.asarray(weight, dtype=np.float64)      # solver_wdotsocp2d.m:208
    dim = 2 if np.ndim(rho0) == 2 else 1
    InitialScaling(var, model, o["scaling"], None, dim=dim, weighted=weighted)
    return var, model, o


def make_state(var, o, model, method="inPALM", weighted=False):
    """The loop object the drivers dispatch to (solver_dotsocp2d.m:205-226)."""
    if method == "acc-ADMM":
        from .accadmm import AccADMMState
        return AccADMMState(var, o, model, weighted=weighted)
    if method == "PALM":
        from .palm import PALMState
        return PALMState(var, o, model, weighted=weighted)
    return InPALMState(var, o, model, weighted=weighted)


def solve_single_level(rho0, rho1, nt, opts, method="inPALM", weight=None):
    """levelN = 1 path of solver_dotsocp2d.m:190-250 / solver_dotsocp1d.m / solver_wdotsocp2d.m."""
    var, model, o = make_level(rho0, rho1, nt, opts, method, weight)
    st = make_state(var, o, model, method, weighted=weight is not None)
    st.run()
    runHist, sigma = st.finish()
    recoverOrgVar(var)
    return var, model, runHist, sigma
