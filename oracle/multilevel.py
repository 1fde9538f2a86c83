"""Multilevel (coarse-to-fine) part of the reference drivers, restated with numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  socp/dot2d/solver_dotsocp2d.m:154-250 (level preparation and loop),
  socp/dot2d/utils/{jump_nextLevel.m, interpolate.m, downSample_phi.m},
  the dot1d twins, and socp/wdot2d/{solver_wdotsocp2d.m:173-247, utils/downSample_barrier.m,
  utils/downSample_q.m, utils/jump_nextLevel.m}.
"""
import numpy as np

from . import mexops
from .driver import InitialScaling, default_opts, make_state, recoverOrgVar
from .model import initialize


def _pairmean(a, axis):
    """movmean(a, 2, axis, 'Endpoints', 'discard')  (interpolate.m:35-37)"""
    s0 = [slice(None)] * a.ndim
    s1 = [slice(None)] * a.ndim
    s0[axis], s1[axis] = slice(0, -1), slice(1, None)
    return (a[tuple(s0)] + a[tuple(s1)]) / 2.0


def downSample_phi(v):
    """socp/dot2d/utils/downSample_phi.m:5-34 (full weighting; the original uses the row index range for
    both dimensions and has `v(2,1)` twice in the (1,1) corner -- restated literally) and
    socp/dot1d/utils/downSample_phi.m:4-12."""
    v = np.asarray(v, dtype=np.float64)
    if v.ndim == 1:
        n = v.size - 1
        nc = n // 2
        out = np.zeros(nc + 1)
        ind = np.arange(2, n - 1, 2)                      # MATLAB 3:2:len-1 (0-based)
        out[1:nc] = 0.5 * v[ind] + 0.25 * (v[ind - 1] + v[ind + 1])
        out[0] = (2 / 3) * v[0] + (1 / 3) * v[1]
        out[-1] = (1 / 3) * v[-2] + (2 / 3) * v[-1]
        return out
    Mx, My = v.shape[0] - 1, v.shape[1] - 1
    Mxc, Myc = Mx // 2, My // 2
    vc = np.zeros((Mxc + 1, Myc + 1))
    ind = np.arange(2, Mx - 1, 2)                         # 3:2:(Mx-1), used for rows AND columns
    I, J = np.ix_(ind, ind)
    vc[1:Mxc, 1:Myc] = (4 * v[I, J] + 2 * (v[I - 1, J] + v[I + 1, J] + v[I, J - 1] + v[I, J + 1])
                        + (v[I - 1, J - 1] + v[I - 1, J + 1] + v[I + 1, J - 1] + v[I + 1, J + 1])) / 16
    vc[0, 1:Myc] = (4 * v[0, ind] + 2 * (v[1, ind] + v[0, ind - 1] + v[0, ind + 1]) + (v[1, ind - 1] + v[1, ind + 1])) / 12
    vc[Mxc, 1:Myc] = (4 * v[Mx, ind] + 2 * (v[Mx - 1, ind] + v[Mx, ind - 1] + v[Mx, ind + 1])
                      + (v[Mx - 1, ind - 1] + v[Mx - 1, ind + 1])) / 12
    vc[1:Mxc, 0] = (4 * v[ind, 0] + 2 * (v[ind - 1, 0] + v[ind + 1, 0] + v[ind, 1]) + (v[ind - 1, 1] + v[ind + 1, 1])) / 12
    vc[1:Mxc, Myc] = (4 * v[ind, My] + 2 * (v[ind - 1, My] + v[ind + 1, My] + v[ind, My - 1])
                      + (v[ind - 1, My - 1] + v[ind + 1, My - 1])) / 12
    vc[0, 0] = (4 * v[0, 0] + 2 * (v[1, 0] + v[0, 1]) + v[1, 0]) / 9
    vc[0, Myc] = (4 * v[0, My] + 2 * (v[1, My] + v[0, My - 1]) + v[1, My - 1]) / 9
    vc[Mxc, 0] = (4 * v[Mx, 0] + 2 * (v[Mx - 1, 0] + v[Mx, 1]) + v[Mx - 1, 1]) / 9
    vc[Mxc, Myc] = (4 * v[Mx, My] + 2 * (v[Mx - 1, My] + v[Mx, My - 1]) + v[Mx - 1, My - 1]) / 9
    return vc


def interpolate_phi(phi, dims):
    """interpolate.m:46-71 (2-D: trilinear) / dot1d interpolate.m:41-61"""
    f = np.asarray(phi).reshape(dims, order="F")
    for ax in range(f.ndim):
        n = f.shape[ax]
        shp = list(f.shape)
        shp[ax] = 2 * (n - 1) + 1
        g = np.zeros(shp)
        s_odd = [slice(None)] * f.ndim
        s_even = [slice(None)] * f.ndim
        s_odd[ax], s_even[ax] = slice(0, None, 2), slice(1, None, 2)
        g[tuple(s_odd)] = f
        g[tuple(s_even)] = _pairmean(f, ax)
        f = g
    return f.ravel(order="F")


def interpolate_z(z, dims):
    """interpolate.m:20-44,73-84: per cone column, nearest in t (each coarse cell -> two fine cells),
    linear in the space axes.  dims = (ny, nx, nt) or (nx, nt) of the COARSE grid."""
    K = z.shape[1]
    sp = dims[:-1]
    ntc = dims[-1] - 1
    cols = []
    for j in range(K):
        f = z[:, j].reshape(tuple(sp) + (ntc,), order="F")
        f = np.repeat(f, 2, axis=-1)                       # fR(.., oddT) = fR(.., evenT) = f
        for ax in range(len(sp)):
            n = f.shape[ax]
            shp = list(f.shape)
            shp[ax] = 2 * (n - 1) + 1
            g = np.zeros(shp)
            s_odd = [slice(None)] * f.ndim
            s_even = [slice(None)] * f.ndim
            s_odd[ax], s_even[ax] = slice(0, None, 2), slice(1, None, 2)
            g[tuple(s_odd)] = f
            g[tuple(s_even)] = _pairmean(f, ax)
            f = g
        cols.append(f.ravel(order="F"))
    return np.asfortranarray(np.stack(cols, axis=1))


def _restrict_linear(nC):
    """(P ./ sum(P,1))' for gene_prolongMat1dim_linear(nC)  (downSample_barrier.m:26-32,10-12)"""
    nR = 2 * (nC - 1) + 1
    P = np.zeros((nR, nC))
    P[np.arange(0, nR, 2), np.arange(nC)] = 1.0
    P[np.arange(1, nR - 1, 2), np.arange(nC - 1)] = 0.5
    P[np.arange(1, nR - 1, 2), np.arange(1, nC)] = 0.5
    return (P / P.sum(axis=0)).T


def _restrict_nearest(nC):
    nR = 2 * nC
    P = np.zeros((nR, nC))
    P[np.arange(0, nR, 2), np.arange(nC)] = 1.0
    P[np.arange(1, nR, 2), np.arange(nC)] = 1.0
    return (P / P.sum(axis=0)).T


def _apply3(v, shape, Ry, Rx, Rt):
    """kron(kron(Rt, Rx), Ry) * v for v stored y fastest, then x, then t"""
    a = v.reshape(shape, order="F")
    a = np.tensordot(Ry, a, axes=(1, 0))
    a = np.moveaxis(np.tensordot(Rx, a, axes=(1, 1)), 0, 1)
    a = np.moveaxis(np.tensordot(Rt, a, axes=(1, 2)), 0, 2)
    return a.ravel(order="F")


def downSample_q(nt, nx, ny, q, log_mean=False):
    """socp/wdot2d/utils/downSample_q.m:4-21; log_mean=True is downSample_barrier.m:4-21
    (restriction of log(weight), then exp)."""
    nt2, nx2, ny2 = (nt + 1) // 2, (nx + 1) // 2, (ny + 1) // 2
    bx = (nt - 1) * nx * ny
    by = bx + nt * (nx - 1) * ny
    v = np.log(q) if log_mean else q
    out = np.concatenate([
        _apply3(v[:bx], (ny, nx, nt - 1), _restrict_linear(ny2), _restrict_linear(nx2), _restrict_nearest(nt2 - 1)),
        _apply3(v[bx:by], (ny, nx - 1, nt), _restrict_linear(ny2), _restrict_nearest(nx2 - 1), _restrict_linear(nt2)),
        _apply3(v[by:], (ny - 1, nx, nt), _restrict_nearest(ny2 - 1), _restrict_linear(nx2), _restrict_linear(nt2))])
    return np.exp(out) if log_mean else out


def downSample_barrier(nt, nx, ny, weight):
    return downSample_q(nt, nx, ny, weight, log_mean=True)


def _grad_times(phi, dims):
    """modelR.grad * phi with the unscaled forward differences of initialize.m:35-39,67-87"""
    f = phi.reshape(dims, order="F")
    parts = [(np.diff(f, axis=-1) * (dims[-1] - 1)).ravel(order="F")]
    if len(dims) == 3:
        parts.append((np.diff(f, axis=1) * (dims[1] - 1)).ravel(order="F"))
        parts.append((np.diff(f, axis=0) * (dims[0] - 1)).ravel(order="F"))
    else:
        parts.append((np.diff(f, axis=0) * (dims[0] - 1)).ravel(order="F"))
    return np.concatenate(parts)


def jump_nextLevel(var, model, rho0, rho1, nt, weight=None):
    """socp/dot2d/utils/jump_nextLevel.m:5-16 (dot1d twin; wdot2d/utils/jump_nextLevel.m with weight)"""
    one_d = not hasattr(model, "ny")
    cdims = (model.nx, model.nt) if one_d else (model.ny, model.nx, model.nt)
    phiR = interpolate_phi(var.phi, cdims)
    betaR = interpolate_z(var.beta, cdims)
    var_init, modelR = initialize(rho0, rho1, nt)
    var.phi, var.beta = phiR, betaR                        # interpolate() returns the same (handle) object
    var.qInd, var.z = var_init.qInd, var_init.z
    fdims = (modelR.nx, modelR.nt) if one_d else (modelR.ny, modelR.nx, modelR.nt)
    var.q = _grad_times(var.phi, fdims)
    alpha = var_init.alpha
    nb = np.asfortranarray(-var.beta)
    if one_d:
        mexops.mexBFdConj1d(alpha, nb, modelR.nt, modelR.nx, 1.0)
    else:
        mexops.mexBFdConj(alpha, nb, modelR.nt, modelR.nx, modelR.ny, 1.0)
    var.alpha = alpha
    if weight is not None:
        modelR.weight = weight
        var.q = var.q / weight
        var.alpha = var.alpha / weight
    return var, modelR


def solve_multilevel(rho0, rho1, nt, levelN, opts, method="inPALM", weight=None, barrier=None, ensure_barrier=None):
    """The level loop of solver_dotsocp2d.m:154-250 / solver_dotsocp1d.m / solver_wdotsocp2d.m:173-247.
    Returns (var, model, runHistML(list per level), sigma)."""
    weighted = weight is not None
    dim = 2 if np.ndim(rho0) == 2 else 1
    o = default_opts(opts, method, weighted)
    tolFactor = -1.0 if o["tol"] > 0.99e-3 else -0.5                      # :124-128
    tolLB = 1e-4 if dim == 2 else 1e-5                                    # :130 / dot1d :121
    rho0s, rho1s, nts, tols, ws = [None] * levelN, [None] * levelN, [None] * levelN, [None] * levelN, [None] * levelN
    rho0s[-1], rho1s[-1], nts[-1], tols[-1], ws[-1] = np.asarray(rho0, float), np.asarray(rho1, float), nt, o["tol"], weight
    for lv in range(levelN - 2, -1, -1):                                  # :166-178
        nts[lv] = (nts[lv + 1] - 1) // 2 + 1
        tols[lv] = max(tols[lv + 1] * 2 ** tolFactor, tolLB)
        rho0s[lv], rho1s[lv] = downSample_phi(rho0s[lv + 1]), downSample_phi(rho1s[lv + 1])
        if weighted:
            nyf, nxf = rho0s[lv + 1].shape
            if barrier is not None:
                rho0s[lv], rho1s[lv], _ = ensure_barrier(rho0s[lv], rho1s[lv], barrier)
                ws[lv] = downSample_barrier(nts[lv + 1], nxf, nyf, ws[lv + 1])
                continue
            ws[lv] = downSample_q(nts[lv + 1], nxf, nyf, ws[lv + 1])
        N = rho0s[lv].size
        rho0s[lv] = rho0s[lv] / (rho0s[lv].sum() / N)
        rho1s[lv] = rho1s[lv] / (rho1s[lv].sum() / N)
    var, model = initialize(rho0s[0], rho1s[0], nts[0])
    if weighted:
        model.weight = ws[0]
    hists, last = [], None
    sigma = None
    for lv in range(levelN):
        InitialScaling(var, model, o["scaling"], last, dim=dim, weighted=weighted)
        o2 = dict(o, tol=tols[lv])
        st = make_state(var, o2, model, method, weighted=weighted)      # solver_dotsocp2d.m:205-226
        st.run()
        runHist, sigma = st.finish()
        recoverOrgVar(var)
        hists.append(runHist)
        if lv < levelN - 1:
            o["time_limit"] = o["time_limit"] - var.time["Total_Time"]
            o["sigma"] = 10 ** (np.log10(o["sigma"] * sigma) / 2)          # :245
            var, model = jump_nextLevel(var, model, rho0s[lv + 1], rho1s[lv + 1], nts[lv + 1],
                                        ws[lv + 1] if weighted else None)
            last = runHist["kkt"][-1]
    return var, model, hists, sigma
