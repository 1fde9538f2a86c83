"""Host-side mirror of the reference's model set-up and output recovery (the O(N),
once-per-level code either side of the hot loop).  Matrix-free: `model.grad` is not built --
the staggered gradient lives in the HIP stencils (dot-socp_amd/csrc/stencil.hip).

Every function names the reference code it mirrors; vectors are 1-D float64 arrays in MATLAB
column-major order, z / beta are (Nz, 10) (1-D: (Nz, 6)) Fortran-ordered matrices.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace

import numpy as np


class VarHandle(SimpleNamespace):
    """socp/dot2d/utils/VarHandle.m:1-32 -- phi, q, z, alpha, beta, cScale, dScale, D, E, E2,
    qInd, name, time; mutated in place by the solvers."""


class ModelHandle(SimpleNamespace):
    """socp/dot2d/utils/ModelHandle.m:1-32 -- nx, ny, nt, c, normc, normd, rho0, rho1[, weight]."""


def normL2(x, h):
    """socp/dot2d/utils/normL2.m:4"""
    return np.sqrt(h) * np.linalg.norm(np.ravel(x))


def initialize(rho0, rho1, nt, lazy_zeros=False, phi=True):
    """[var, model] = initialize(rho0, rho1, nt)
    socp/dot2d/utils/initialize.m:1-65 (rho0: ny x nx) / socp/dot1d/utils/initialize.m:1-58.
    lazy_zeros: z, beta, q, alpha stay None (the device default is the all-zero start) and InitialScaling is told that c
    is zero between its first and last layer; phi=False: var.phi stays None too (a multilevel warm start makes it on the
    device) -- at 1025 x 1025 x 129 these arrays are 1 GB each and the level loop is 0.8 s."""
    rho0 = np.asarray(rho0, dtype=np.float64)
    rho1 = np.asarray(rho1, dtype=np.float64)
    var, model = VarHandle(), ModelHandle()
    model.rho0, model.rho1 = rho0, rho1
    ht = 1.0 / (nt - 1)
    if rho0.ndim == 2:
        ny, nx = rho0.shape
        n = nx * ny * nt
        hx, hy = 1.0 / (nx - 1), 1.0 / (ny - 1)
        bx = (nt - 1) * nx * ny
        var.qInd = SimpleNamespace(bx=bx, by=nt * (nx - 1) * ny + bx)       # 0-based offsets
        model.nx, model.ny, model.nt = nx, ny, nt
        nq = bx + nt * (nx - 1) * ny + nt * nx * (ny - 1)
        model.c = np.zeros(n)
        model.c[:nx * ny] = -rho0.ravel(order="F") / ht
        model.c[n - nx * ny:] = rho1.ravel(order="F") / ht
        xx, yy = np.meshgrid(np.arange(nx) * hx, np.arange(ny) * hy)
        var.phi = np.tile((0.5 * (xx ** 2 + yy ** 2)).ravel(order="F"), nt) if phi else None
        K = 10
    else:
        nx = rho0.size
        n = nx * nt
        hx = 1.0 / (nx - 1)
        bx = (nt - 1) * nx
        var.qInd = SimpleNamespace(bx=bx)
        model.nx, model.nt = nx, nt
        nq = bx + nt * (nx - 1)
        model.c = np.zeros(n)
        model.c[:nx] = -rho0.ravel() / ht
        model.c[n - nx:] = rho1.ravel() / ht
        var.phi = np.tile(0.5 * (np.arange(nx) * hx) ** 2, nt) if phi else None
        K = 6
    if lazy_zeros:
        # the all-zero start (initialize.m:52-59) is the device default: nothing to allocate or upload
        var.z = var.beta = var.q = var.alpha = None
        var.zshape, var.nq = (bx, K), nq
        model.n_global = n
        model._c_ends = n // nt          # c is zero except for its first and last layer (of that many nodes)
    else:
        var.z = np.zeros((bx, K), order="F")
        var.beta = np.zeros((bx, K), order="F")
        var.q = np.zeros(nq)
        var.alpha = np.zeros(nq)
    model.grad = None        # matrix-free (see module docstring)
    return var, model


def initialize_slab(rho0, rho1, nt, t0, t1):
    """The time slab [t0, t1) of what initialize(rho0, rho1, nt) returns (cold start: q, alpha, z,
    beta stay on the device as zeros), for the one-process-per-GPU mode: var.phi and model.c hold
    only the slab's nodes; model.nt stays global; model.n_global / model.c_sumsq carry what
    InitialScaling needs from the full arrays (initialize.m:42-50)."""
    rho0 = np.asarray(rho0, dtype=np.float64)
    rho1 = np.asarray(rho1, dtype=np.float64)
    var, model = VarHandle(), ModelHandle()
    model.rho0, model.rho1 = rho0, rho1
    ht = 1.0 / (nt - 1)
    ntl = t1 - t0
    if rho0.ndim == 2:
        ny, nx = rho0.shape
        model.nx, model.ny, model.nt = nx, ny, nt
        xx, yy = np.meshgrid(np.arange(nx) * (1.0 / (nx - 1)), np.arange(ny) * (1.0 / (ny - 1)))
        prof = (0.5 * (xx ** 2 + yy ** 2)).ravel(order="F")
        plane = nx * ny
        r0, r1 = rho0.ravel(order="F"), rho1.ravel(order="F")
    else:
        nx = rho0.size
        model.nx, model.nt = nx, nt
        prof = 0.5 * (np.arange(nx) * (1.0 / (nx - 1))) ** 2
        plane = nx
        r0, r1 = rho0.ravel(), rho1.ravel()
    var.phi = np.tile(prof, ntl)
    model.c = np.zeros(plane * ntl)
    c0, c1 = -r0 / ht, r1 / ht
    if t0 == 0:
        model.c[:plane] = c0
    if t1 == nt:
        model.c[plane * (ntl - 1):] = c1
    model.n_global = plane * nt
    model.c_sumsq = float(np.dot(c0, c0) + np.dot(c1, c1))
    model.slab = (t0, t1)
    var.z = var.beta = var.q = var.alpha = None
    var.qInd = None
    model.grad = None
    return var, model


def InitialScaling(var, model, scalingYes, lastLevelKKT=None, dim=2, weighted=False):
    """socp/dot2d/solver_dotsocp2d.m:304-365; 1-D: solver_dotsocp1d.m:263-300 (hMean = h^(1/2));
    weighted: solver_wdotsocp2d.m:297-343 (`adjust`, E2 safeguard 4)."""
    h = 1.0 / (model.n_global if hasattr(model, "n_global") else var.phi.size)
    ends = getattr(model, "_c_ends", None)      # initialize(lazy_zeros=True): only c's two end layers are non-zero
    hMean = h ** (1.0 / 3.0) if dim == 2 else h ** 0.5
    if lastLevelKKT is None or not hasattr(var, "E2"):
        Escale2 = np.sqrt(2.0)
    elif weighted:
        Escale2 = var.E2 * min(4.0, max(0.25, np.sqrt(lastLevelKKT[0] / lastLevelKKT[1])))
    else:
        ratio = np.sqrt(lastLevelKKT[0] / lastLevelKKT[1])
        lowerRatio = 0.8333
        if ratio < lowerRatio:
            Escale2 = var.E2 * max(1 / np.sqrt(2.0), ratio / lowerRatio)
        else:
            Escale2 = var.E2 * min(np.sqrt(2.0), max(1.0, ratio))
    def _norm_c():
        # slab mode: ||c|| of the full vector from its two non-zero layers
        if hasattr(model, "c_sumsq"):
            return np.sqrt(h) * np.sqrt(model.c_sumsq)
        if ends is not None and model.c.size > 2 * ends:
            return normL2(np.concatenate([model.c[:ends], model.c[model.c.size - ends:]]), h)
        return normL2(model.c, h)

    if scalingYes:
        norm_c = _norm_c() * np.sqrt(model.nt)
        norm_d = np.sqrt(2.0)
        adjust = 10.0 ** np.mean(np.log10(model.weight + 1e-10)) if weighted else 1.0
        D = np.sqrt(2.0) * np.sqrt(hMean) * adjust
        E = D / Escale2
        cScale = max(1.0, norm_c * np.sqrt(hMean) / adjust)
        dScale = E * norm_d * np.sqrt(adjust)
        model.normc = norm_c / cScale
        model.normd = norm_d * E / dScale
        if ends is not None and model.c.size > 2 * ends:      # the same products, without a pass over the zeros in between
            model.c[:ends] = (1.0 / cScale) * model.c[:ends]
            model.c[model.c.size - ends:] = (1.0 / cScale) * model.c[model.c.size - ends:]
        else:
            model.c = (1.0 / cScale) * model.c
        if var.phi is not None:                 # None: the state is produced on the device (multilevel warm start)
            var.phi = (1.0 / dScale) * var.phi
        if var.q is not None:
            var.q = (D / dScale) * var.q
            var.z = (E / dScale) * var.z
            var.alpha = (1.0 / cScale / D) * var.alpha
            var.beta = (1.0 / cScale / E) * var.beta
    else:
        cScale = dScale = D = E = 1.0
        model.normc = _norm_c()
        model.normd = np.sqrt(2.0)
    var.cScale, var.dScale, var.D, var.E, var.E2 = cScale, dScale, D, E, Escale2


def recoverOrgVar(var):
    """socp/dot2d/solver_dotsocp2d.m:368-386"""
    cScale, dScale, D, E = var.cScale, var.dScale, var.D, var.E
    var.phi = dScale * var.phi
    var.z = (dScale / E) * var.z
    var.q = (dScale / D) * var.q
    var.alpha = (cScale * D) * var.alpha
    var.beta = (cScale * E) * var.beta


def recover_RhoE(var, model, weighted=False):
    """socp/dot2d/utils/recover_RhoE.m:14-25, socp/dot1d/utils/recover_RhoE.m (1-D),
    socp/wdot2d/utils/recover_RhoE.m:11 (alpha = weight .* alpha)."""
    alpha = model.weight * var.alpha if weighted else var.alpha
    nt = model.nt
    if hasattr(model, "ny"):
        ny, nx, qi = model.ny, model.nx, var.qInd
        rho = alpha[:qi.bx].reshape((ny, nx, nt - 1), order="F")
        rho = np.concatenate([model.rho0[:, :, None], (rho[:, :, :-1] + rho[:, :, 1:]) / 2,
                              model.rho1[:, :, None]], axis=2)
        Ex = alpha[qi.bx:qi.by].reshape((ny, nx - 1, nt), order="F").copy()
        Ex[:, :, [0, -1]] *= 2
        Ex = np.concatenate([np.zeros((ny, 1, nt)), (Ex[:, :-1] + Ex[:, 1:]) / 2, np.zeros((ny, 1, nt))], axis=1)
        Ey = alpha[qi.by:].reshape((ny - 1, nx, nt), order="F").copy()
        Ey[:, :, [0, -1]] *= 2
        Ey = np.concatenate([np.zeros((1, nx, nt)), (Ey[:-1] + Ey[1:]) / 2, np.zeros((1, nx, nt))], axis=0)
        return rho, Ex, Ey
    nx, nb = model.nx, var.qInd.bx
    rho = alpha[:nb].reshape((nx, nt - 1), order="F")
    rho = np.concatenate([model.rho0.reshape(nx, 1), (rho[:, :-1] + rho[:, 1:]) / 2, model.rho1.reshape(nx, 1)], axis=1)
    Ex = alpha[nb:].reshape((nx - 1, nt), order="F").copy()
    Ex[:, [0, -1]] *= 2
    Ex = np.concatenate([np.zeros((1, nt)), (Ex[:-1] + Ex[1:]) / 2, np.zeros((1, nt))], axis=0)
    return rho, Ex


def recover_q(var, model):
    """socp/dot2d/utils/recover_q.m:12-22, socp/dot1d/utils/recover_q.m."""
    q, nt = var.q, model.nt
    if hasattr(model, "ny"):
        ny, nx, qi = model.ny, model.nx, var.qInd
        q0 = q[:qi.bx].reshape((ny, nx, nt - 1), order="F")
        bx = q[qi.bx:qi.by].reshape((ny, nx - 1, nt), order="F")
        bx = np.concatenate([np.zeros((ny, 1, nt)), (bx[:, :-1] + bx[:, 1:]) / 2, np.zeros((ny, 1, nt))], axis=1)
        bx = (bx[:, :, :-1] + bx[:, :, 1:]) / 2
        by = q[qi.by:].reshape((ny - 1, nx, nt), order="F")
        by = np.concatenate([np.zeros((1, nx, nt)), (by[:-1] + by[1:]) / 2, np.zeros((1, nx, nt))], axis=0)
        by = (by[:, :, :-1] + by[:, :, 1:]) / 2
        return q0, bx, by
    nx, nb = model.nx, var.qInd.bx
    q0 = q[:nb].reshape((nx, nt - 1), order="F")
    bx = q[nb:].reshape((nx - 1, nt), order="F")
    bx = np.concatenate([np.zeros((1, nt)), (bx[:-1] + bx[1:]) / 2, np.zeros((1, nt))], axis=0)
    bx = (bx[:, :-1] + bx[:, 1:]) / 2
    return q0, bx


def check_massConservation(rho, tol=1e-2):
    """socp/dot2d/utils/check_massConservation.m:16-34 (integralL2 = per-layer mean)."""
    nt = rho.shape[-1]
    rho2 = rho.reshape((-1, nt), order="F")
    if rho2.size >= (1 << 22) and rho2.flags.f_contiguous:
        # layer by layer (a layer stays in cache between its two sums, no array-sized temporary) on a few threads:
        # 0.26 -> 0.03 s on the 1 GB rho of a 1025 x 1025 x 129 solve
        def layer(t):
            col = rho2[:, t]
            return col.sum(), np.minimum(col, 0.0).sum()
        with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
            sums = np.array(list(ex.map(layer, range(nt)))) / rho2.shape[0]
        sumRho, sumNega = sums[:, 0], sums[:, 1]
    else:
        sumRho = rho2.mean(axis=0)
        sumNega = np.where(rho2 < 0, rho2, 0.0).mean(axis=0)
    err = max(np.max(np.abs(sumRho - 1)), np.max(np.abs(sumNega)))
    return bool(err <= tol)
