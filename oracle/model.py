"""Discrete model + small operators of the reference, restated with numpy/scipy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  All vectors are 1-D float64 arrays in
MATLAB column-major order (y fastest, then x, then t); z and beta are (Nz, 10) /
(Nz, 6) Fortran-ordered matrices.
"""
from types import SimpleNamespace

import numpy as np
import scipy.fft as sfft
import scipy.sparse as sp


class VarHandle(SimpleNamespace):
    """socp/dot2d/utils/VarHandle.m:1-32 -- by-reference carrier of the iterates."""


class ModelHandle(SimpleNamespace):
    """socp/dot2d/utils/ModelHandle.m:1-32 -- by-reference carrier of the model."""


# ----------------------------------------------------------------------------------
# initialize.m
# ----------------------------------------------------------------------------------
def _fwd_diff(n, h):
    """spdiags([-1/h, 1/h], [0, 1], n-1, n)  (initialize.m:68-69,77-78,84-85)."""
    e = np.full(n - 1, 1.0 / h)
    return sp.diags([-e, e], [0, 1], shape=(n - 1, n), format="csc")


def initialize(rho0, rho1, nt):
    """socp/dot2d/utils/initialize.m:1-87 (2-D: rho0 is ny x nx) and
    socp/dot1d/utils/initialize.m:1-72 (1-D: rho0 is a vector of nx)."""
    rho0 = np.asarray(rho0, dtype=np.float64)
    rho1 = np.asarray(rho1, dtype=np.float64)
    var, model = VarHandle(), ModelHandle()
    model.rho0, model.rho1 = rho0, rho1
    ht = 1.0 / (nt - 1)
    if rho0.ndim == 2:
        ny, nx = rho0.shape
        n = nx * ny * nt
        hx, hy = 1.0 / (nx - 1), 1.0 / (ny - 1)
        # qInd (1-based in MATLAB; stored 0-based offsets here) -- initialize.m:18-20
        bx = (nt - 1) * nx * ny
        by = nt * (nx - 1) * ny + bx
        var.qInd = SimpleNamespace(bx=bx, by=by)
        model.nx, model.ny, model.nt = nx, ny, nt
        It, Ix, Iy = sp.identity(nt, format="csc"), sp.identity(nx, format="csc"), sp.identity(ny, format="csc")
        Dt = sp.kron(_fwd_diff(nt, ht), sp.identity(nx * ny, format="csc"), format="csc")   # :67-72
        Dx = sp.kron(sp.kron(It, _fwd_diff(nx, hx), format="csc"), Iy, format="csc")        # :74-80
        Dy = sp.kron(sp.identity(nt * nx, format="csc"), _fwd_diff(ny, hy), format="csc")   # :82-87
        model.grad = sp.vstack([Dt, Dx, Dy], format="csc")                                  # :35-39
        model.c = np.zeros(n)                                                               # :42-44
        model.c[: nx * ny] = -rho0.ravel(order="F") / ht
        model.c[n - nx * ny:] = rho1.ravel(order="F") / ht
        xx, yy = np.meshgrid(np.arange(nx) * hx, np.arange(ny) * hy)                        # :48-50
        phi2 = 0.5 * (xx ** 2 + yy ** 2)
        var.phi = np.tile(phi2.ravel(order="F"), nt)
        lenA = (nt - 1) * nx * ny                                                            # :53-55
        var.z = np.zeros((lenA, 10), order="F")
        var.beta = np.zeros((lenA, 10), order="F")
    else:
        nx = rho0.size
        n = nx * nt
        hx = 1.0 / (nx - 1)
        var.qInd = SimpleNamespace(bx=(nt - 1) * nx)                                        # dot1d initialize.m:15-16
        model.nx, model.nt = nx, nt
        Dt = sp.kron(_fwd_diff(nt, ht), sp.identity(nx, format="csc"), format="csc")        # :60-65
        Dx = sp.kron(sp.identity(nt, format="csc"), _fwd_diff(nx, hx), format="csc")        # :67-72
        model.grad = sp.vstack([Dt, Dx], format="csc")
        model.c = np.zeros(n)                                                               # :35-37
        model.c[:nx] = -rho0.ravel() / ht
        model.c[n - nx:] = rho1.ravel() / ht
        xx = np.arange(nx) * hx                                                             # :41-43
        var.phi = np.tile(0.5 * xx ** 2, nt)
        lenA = (nt - 1) * nx
        var.z = np.zeros((lenA, 6), order="F")
        var.beta = np.zeros((lenA, 6), order="F")
    m = model.grad.shape[0]
    var.q = np.zeros(m)                                                                     # initialize.m:58-59
    var.alpha = np.zeros(m)
    return var, model


# ----------------------------------------------------------------------------------
# initialize_FFTkernel.m / oper_q.m / oper_poisson*.m / mirt_dctn.m
# ----------------------------------------------------------------------------------
def initialize_FFTkernel(nt, nx, ny=None):
    """socp/dot2d/utils/initialize_FFTkernel.m:6-15 (3 axes) and
    socp/dot1d/utils/initialize_FFTkernel.m:6-13 (2 axes).  Returns an array shaped
    (ny, nx, nt) resp. (nx, nt)."""
    CT = (2.0 * (nt - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(nt) / nt))
    CX = (2.0 * (nx - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(nx) / nx))
    if ny is None:
        kernel = CX[:, None] + CT[None, :]
    else:
        CY = (2.0 * (ny - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(ny) / ny))
        kernel = (CY[:, None, None] + CX[None, :, None]) + CT[None, None, :]
    kernel[kernel == 0] = 1.0
    return kernel


def oper_q(dims, D, E, weight=None):
    """socp/dot2d/utils/oper_q.m:13-26, socp/dot1d/utils/oper_q.m:8-20 and (weighted)
    socp/wdot2d/utils/oper_q.m:15-28.  dims = (ny, nx, nt) or (nx, nt)."""
    tmp = (E / D) ** 2
    one = 0.0 if weight is not None else 1.0
    c1, c2 = one + 2.0 * tmp, one + tmp
    if len(dims) == 3:
        ny, nx, nt = dims
        a = np.full((ny, nx, nt - 1), c1)
        b = np.full((ny, nx - 1, nt), c1)
        c = np.full((ny - 1, nx, nt), c1)
        b[:, :, [0, -1]] = c2
        c[:, :, [0, -1]] = c2
        diag = np.concatenate([a.ravel(order="F"), b.ravel(order="F"), c.ravel(order="F")])
    else:
        nx, nt = dims
        a = np.full((nx, nt - 1), c1)
        b = np.full((nx - 1, nt), c1)
        b[:, [0, -1]] = c2
        diag = np.concatenate([a.ravel(order="F"), b.ravel(order="F")])
    if weight is not None:
        diag = diag + weight ** 2
    return diag


def mirt_dct_1d(a):
    """1-D orthonormal DCT-II along axis 0, literally as
    socp/dot2d/utils/mirt_dctn.m:69-71 (weights/indices) and :100-141 (reorder, fft,
    multiply weights, real part)."""
    n = a.shape[0]
    ww = 2.0 * np.exp((-1j * np.pi / (2 * n)) * np.arange(n)) / np.sqrt(2 * n)
    ww[0] = ww[0] / np.sqrt(2.0)
    ind = np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)[::-1]])
    v = np.fft.fft(a[ind], axis=0)
    return np.real(ww.reshape((n,) + (1,) * (a.ndim - 1)) * v)


def mirt_idct_1d(a):
    """1-D orthonormal DCT-III along axis 0, literally as
    socp/dot2d/utils/mirt_idctn.m:66-72 and :98-128: multiply by the *same* weights,
    forward fft, reorder, real part."""
    n = a.shape[0]
    ww = 2.0 * np.exp((-1j * np.pi / (2 * n)) * np.arange(n)) / np.sqrt(2 * n)
    ww[0] = ww[0] / np.sqrt(2.0)
    tmp = np.empty(n, dtype=np.int64)
    tmp[0::2] = np.arange(0, (n + 1) // 2)
    tmp[1::2] = np.arange(n - 1, (n + 1) // 2 - 1, -1)
    v = np.fft.fft(ww.reshape((n,) + (1,) * (a.ndim - 1)) * a, axis=0)
    return np.real(v[tmp])


def mirt_dctn(a):
    """N-D version: the 1-D transform along every axis in turn (mirt_dctn.m:78-96)."""
    for ax in range(a.ndim):
        a = np.moveaxis(mirt_dct_1d(np.moveaxis(a, ax, 0)), 0, ax)
    return a


def mirt_idctn(a):
    for ax in range(a.ndim):
        a = np.moveaxis(mirt_idct_1d(np.moveaxis(a, ax, 0)), 0, ax)
    return a


FFT_WORKERS = -1      # threads of scipy's DCT (-1: all cores, like MATLAB's implicitly multithreaded fft); bench.py's
                      # cpu_baseline leg sets 1 for its single-thread figure


def oper_poisson(kernel, rhs, fast=True, workers=None):
    """socp/dot2d/utils/oper_poisson3dim.m:4 and socp/dot1d/utils/oper_poisson.m:4:
    res = idctn(dctn(rhs) ./ kernel).  `rhs` is shaped like `kernel` in Fortran sense.
    fast=True uses scipy's orthonormal DCT-II/III (identical transform, see
    tests/test_oracle_invariants.py::test_mirt_dct_equals_scipy)."""
    if workers is None:
        workers = FFT_WORKERS
    if fast:
        return sfft.idctn(sfft.dctn(rhs, norm="ortho", workers=workers) / kernel, norm="ortho", workers=workers)
    return mirt_idctn(mirt_dctn(rhs) / kernel)


# ----------------------------------------------------------------------------------
# norms, sigma rule, DOT complementarity
# ----------------------------------------------------------------------------------
def normL2(x, h):
    """socp/dot2d/utils/normL2.m:4"""
    return np.sqrt(h) * np.linalg.norm(np.ravel(x))


def FnormL2(x, h):
    """socp/dot2d/utils/FnormL2.m:4"""
    return np.sqrt(h) * np.linalg.norm(np.ravel(x))


UPDATE_RULE = np.array([  # solver_socp_inPALM.m:39-51
    [1.1, 1.10], [1.2, 1.15], [1.5, 1.20], [2, 1.26], [2.5, 1.28], [3.33, 1.32],
    [5, 1.35], [10, 1.40], [20, 1.60], [40, 1.80], [50, 2.00]])


def _get_factor(xi, rule):
    """adjust_lagrangianParam.m:49-60"""
    factor = 1.0
    for i in range(rule.shape[0]):
        if xi >= rule[i, 0]:
            factor = rule[i, 1]
        else:
            break
    return factor


def adjust_lagrangianParam(sigma, xi, rule=UPDATE_RULE, bound=(1e-3, 1e3)):
    """socp/dot2d/utils/adjust_lagrangianParam.m:14-39"""
    factor = 1.0
    if xi >= 1:
        factor = _get_factor(xi, rule)
    elif xi < 1:
        factor = 1.0 / _get_factor(1.0 / xi, rule)
    if factor != 1:
        old = sigma
        sigma = max(min(sigma * factor, bound[1]), bound[0])
        factor = sigma / old
    return sigma, factor


def IfAdjustSigma(it, last_it):
    """solver_socp_inPALM.m:361-379"""
    passed = it - last_it
    if it < 20 and passed >= 3:
        return True
    if it < 50 and passed >= 6:
        return True
    if it < 100 and passed >= 10:
        return True
    if it < 200 and passed >= 15:
        return True
    if it < 500 and passed >= 25:
        return True
    return passed >= 40


def _pairmean(a, axis):
    """movmean(a, 2, axis, "Endpoints", "discard")"""
    sl0 = [slice(None)] * a.ndim
    sl1 = [slice(None)] * a.ndim
    sl0[axis] = slice(0, -1)
    sl1[axis] = slice(1, None)
    return (a[tuple(sl0)] + a[tuple(sl1)]) / 2.0


def compute_kkt_dot_complement(q, alpha, z2, sigma, h, dims, qInd, cScale, dScale, D, E, weight=None):
    """socp/dot2d/utils/compute_kkt_dot_complement.m:2-18,
    socp/dot1d/utils/compute_kkt_dot_complement.m:2-15,
    socp/wdot2d/utils/compute_kkt_dot_complement.m:2-19 (Dalpha = weight .* alpha)."""
    Dalpha = alpha if weight is None else weight * alpha
    nb = qInd.bx
    rhoT = (sigma * cScale * D) * Dalpha[:nb]
    K = z2.shape[1]
    rhoFq = rhoT + (dScale / D) * q[:nb] + np.sum(((dScale / E) * z2[:, 1:K - 1]) ** 2, axis=1) / 4.0
    rhoFq[rhoFq < 0] = 0.0
    dotcomplem = normL2(rhoT - rhoFq, h)
    normRho = normL2(rhoT, h)
    norm_rhoFq = normL2(rhoFq, h)
    if len(dims) == 3:
        ny, nx, nt = dims
        pad = np.zeros((ny, nx, nt + 1))
        pad[:, :, 1:nt] = rhoT.reshape((ny, nx, nt - 1), order="F")
        rho = _pairmean(pad, 2)
        rhoBx = (dScale / D) * (_pairmean(rho, 1).ravel(order="F") * q[qInd.bx:qInd.by])
        rhoBy = (dScale / D) * (_pairmean(rho, 0).ravel(order="F") * q[qInd.by:])
        mx = (sigma * cScale * D) * Dalpha[qInd.bx:qInd.by]
        my = (sigma * cScale * D) * Dalpha[qInd.by:]
        mRhoB = np.sqrt(normL2(mx - rhoBx, h) ** 2 + normL2(my - rhoBy, h) ** 2)
        normM = np.sqrt(normL2(mx, h) ** 2 + normL2(my, h) ** 2)
        normRhoB = np.sqrt(normL2(rhoBx, h) ** 2 + normL2(rhoBy, h) ** 2)
    else:
        nx, nt = dims
        pad = np.zeros((nx, nt + 1))
        pad[:, 1:nt] = rhoT.reshape((nx, nt - 1), order="F")
        rho = _pairmean(pad, 1)
        rhoBx = (dScale / D) * (_pairmean(rho, 0).ravel(order="F") * q[qInd.bx:])
        mx = (sigma * cScale * D) * Dalpha[qInd.bx:]
        normM = normL2(mx, h)
        normRhoB = normL2(rhoBx, h)
        mRhoB = normL2(mx - rhoBx, h)
    return dotcomplem, normRho, norm_rhoFq, mRhoB, normM, normRhoB
