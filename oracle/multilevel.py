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


# ---------------------------------------------------------------------------------------------------------------
# The restriction / prolongation operators below are written STRAIGHT from the .m files, statement by statement and
# with the reference's own 1-based index vectors (explicit loops, sparse kron products) -- deliberately NOT the
# vectorised form of dot-socp_amd/multilevel.py, so that tests/test_multilevel.py compares two implementations.
# ---------------------------------------------------------------------------------------------------------------
import scipy.sparse as sp


def downSample_phi(v):
    """socp/dot2d/utils/downSample_phi.m:5-34, element by element in the reference's 1-based indices (the original
    uses `ind = 3:2:(Mx-1)` for rows AND columns and has `v(2,1)` twice in the (1,1) corner -- kept), and
    socp/dot1d/utils/downSample_phi.m:4-12."""
    v = np.asarray(v, dtype=np.float64)
    if v.ndim == 1:
        ln = v.size - 1                                   # dot1d downSample_phi.m:4
        lenc = ln // 2
        phic = np.zeros(lenc + 1)
        V = lambda i: v[i - 1]                            # 1-based access  # noqa: E731
        for c, i in enumerate(range(3, ln - 1 + 1, 2)):   # ind = 3:2:len-1 -> phic(2:lenc)
            phic[1 + c] = 0.5 * V(i) + 0.25 * (V(i - 1) + V(i + 1))
        phic[0] = (2 / 3) * V(1) + (1 / 3) * V(2)
        phic[lenc] = (1 / 3) * V(ln) + (2 / 3) * V(ln + 1)
        return phic
    Mx, My = v.shape[0] - 1, v.shape[1] - 1               # :5
    Mxc, Myc = Mx // 2, My // 2
    vc = np.zeros((Mxc + 1, Myc + 1))
    V = lambda i, j: v[i - 1, j - 1]                      # noqa: E731
    ind = list(range(3, Mx - 1 + 1, 2))                   # :9  (one vector for both dimensions)
    if len(ind) != Myc - 1:
        # vc(2:Mxc,2:Myc) = f(v(ind,ind)): MATLAB raises a dimension mismatch unless the array is square
        raise ValueError("downSample_phi: the reference indexes both axes with the row range (square arrays only)")
    for a, i in enumerate(ind):                           # :11-15 -> vc(2:Mxc, 2:Myc)
        for b, j in enumerate(ind):
            vc[1 + a, 1 + b] = (4 * V(i, j) + 2 * (V(i - 1, j) + V(i + 1, j) + V(i, j - 1) + V(i, j + 1))
                                + (V(i - 1, j - 1) + V(i - 1, j + 1) + V(i + 1, j - 1) + V(i + 1, j + 1))) / 16
    for b, j in enumerate(ind):                           # :17-22 first / last row
        vc[0, 1 + b] = (4 * V(1, j) + 2 * (V(2, j) + V(1, j - 1) + V(1, j + 1)) + (V(2, j - 1) + V(2, j + 1))) / 12
        vc[Mxc, 1 + b] = (4 * V(Mx + 1, j) + 2 * (V(Mx, j) + V(Mx + 1, j - 1) + V(Mx + 1, j + 1))
                          + (V(Mx, j - 1) + V(Mx, j + 1))) / 12
    for a, i in enumerate(ind):                           # :23-28 first / last column
        vc[1 + a, 0] = (4 * V(i, 1) + 2 * (V(i - 1, 1) + V(i + 1, 1) + V(i, 2)) + (V(i - 1, 2) + V(i + 1, 2))) / 12
        vc[1 + a, Myc] = (4 * V(i, My + 1) + 2 * (V(i - 1, My + 1) + V(i + 1, My + 1) + V(i, My))
                          + (V(i - 1, My) + V(i + 1, My))) / 12
    vc[0, 0] = (4 * V(1, 1) + 2 * (V(2, 1) + V(1, 2)) + V(2, 1)) / 9                       # :30
    vc[0, Myc] = (4 * V(1, My + 1) + 2 * (V(2, My + 1) + V(1, My)) + V(2, My)) / 9         # :31
    vc[Mxc, 0] = (4 * V(Mx + 1, 1) + 2 * (V(Mx, 1) + V(Mx + 1, 2)) + V(Mx, 2)) / 9         # :32
    vc[Mxc, Myc] = (4 * V(Mx + 1, My + 1) + 2 * (V(Mx, My + 1) + V(Mx + 1, My)) + V(Mx, My)) / 9   # :33-34
    return vc


def _movmean2(f, d):
    """movmean(f, 2, d, 'Endpoints', 'discard')  (interpolate.m:17-19): means of neighbouring pairs along dim d"""
    n = f.shape[d]
    out_shape = list(f.shape)
    out_shape[d] = n - 1
    out = np.empty(out_shape)
    for i in range(n - 1):
        lo = [slice(None)] * f.ndim
        hi = [slice(None)] * f.ndim
        dst = [slice(None)] * f.ndim
        lo[d], hi[d], dst[d] = i, i + 1, i
        out[tuple(dst)] = (f[tuple(lo)] + f[tuple(hi)]) / 2.0
    return out


def interpolate_phi(phi, dims):
    """interpolate.m:46-71 (2-D: y, then x, then t linear, on the odd / even index vectors of the reference) and
    socp/dot1d/utils/interpolate.m (x, then t)."""
    if len(dims) == 3:
        ny, nx, nt = dims
        nyR, nxR, ntR = 2 * (ny - 1) + 1, 2 * (nx - 1) + 1, 2 * (nt - 1) + 1
        oddY, evenY = np.arange(0, nyR, 2), np.arange(1, nyR - 1, 2)        # 1:2:nyR, 2:2:nyR-1 (0-based here)
        oddX, evenX = np.arange(0, nxR, 2), np.arange(1, nxR - 1, 2)
        oddT, evenT = np.arange(0, ntR, 2), np.arange(1, ntR - 1, 2)
        phiR = np.zeros((nyR, nxR, ntR))
        phiR[np.ix_(oddY, oddX, oddT)] = np.asarray(phi).reshape((ny, nx, nt), order="F")
        phiR[np.ix_(evenY, oddX, oddT)] = _movmean2(phiR[np.ix_(oddY, oddX, oddT)], 0)
        phiR[np.ix_(np.arange(nyR), evenX, oddT)] = _movmean2(phiR[np.ix_(np.arange(nyR), oddX, oddT)], 1)
        phiR[:, :, evenT] = _movmean2(phiR[:, :, oddT], 2)
        return phiR.reshape(-1, order="F")
    nx, nt = dims
    nxR, ntR = 2 * (nx - 1) + 1, 2 * (nt - 1) + 1
    oddX, evenX = np.arange(0, nxR, 2), np.arange(1, nxR - 1, 2)
    oddT, evenT = np.arange(0, ntR, 2), np.arange(1, ntR - 1, 2)
    phiR = np.zeros((nxR, ntR))
    phiR[np.ix_(oddX, oddT)] = np.asarray(phi).reshape((nx, nt), order="F")
    phiR[np.ix_(evenX, oddT)] = _movmean2(phiR[np.ix_(oddX, oddT)], 0)
    phiR[:, evenT] = _movmean2(phiR[:, oddT], 1)
    return phiR.reshape(-1, order="F")


def _interpolate_tStagger(f):
    """interpolate.m:22-44: nearest in t (each coarse cell fills fine cells 2t-1 and 2t), linear in y, then in x"""
    if f.ndim == 3:
        ny, nx, nt = f.shape
        nyR, nxR, ntR = 2 * (ny - 1) + 1, 2 * (nx - 1) + 1, 2 * nt
        oddY, evenY = np.arange(0, nyR, 2), np.arange(1, nyR - 1, 2)
        oddX, evenX = np.arange(0, nxR, 2), np.arange(1, nxR - 1, 2)
        oddT, evenT = np.arange(0, ntR - 1, 2), np.arange(1, ntR, 2)
        fR = np.zeros((nyR, nxR, ntR))
        fR[np.ix_(oddY, oddX, oddT)] = f
        fR[np.ix_(oddY, oddX, evenT)] = f
        fR[np.ix_(evenY, oddX, np.arange(ntR))] = _movmean2(fR[np.ix_(oddY, oddX, np.arange(ntR))], 0)
        fR[:, evenX, :] = _movmean2(fR[:, oddX, :], 1)
        return fR
    nx, nt = f.shape
    nxR, ntR = 2 * (nx - 1) + 1, 2 * nt
    oddX, evenX = np.arange(0, nxR, 2), np.arange(1, nxR - 1, 2)
    oddT, evenT = np.arange(0, ntR - 1, 2), np.arange(1, ntR, 2)
    fR = np.zeros((nxR, ntR))
    fR[np.ix_(oddX, oddT)] = f
    fR[np.ix_(oddX, evenT)] = f
    fR[evenX, :] = _movmean2(fR[oddX, :], 0)
    return fR


def interpolate_z(z, dims):
    """interpolate.m:73-84: column by column through interpolate_tStagger.  dims = (ny, nx, nt) or (nx, nt) of the
    COARSE grid."""
    K = z.shape[1]
    sp_dims = tuple(dims[:-1])
    ntc = dims[-1] - 1
    first = _interpolate_tStagger(z[:, 0].reshape(sp_dims + (ntc,), order="F"))
    zR = np.zeros((first.size, K), order="F")
    zR[:, 0] = first.reshape(-1, order="F")
    for j in range(1, K):
        zR[:, j] = _interpolate_tStagger(z[:, j].reshape(sp_dims + (ntc,), order="F")).reshape(-1, order="F")
    return zR


def _gene_prolongMat1dim_linear(nC):
    """downSample_q.m:25-31 (triplets iVec / jVec / vVec, 1-based in the reference)"""
    nR = 2 * (nC - 1) + 1
    iVec = list(range(1, nR + 1, 2)) + list(range(2, nR, 2)) * 2
    jVec = list(range(1, nC + 1)) + list(range(1, nC)) + list(range(2, nC + 1))
    vVec = [1.0] * nC + [0.5] * (2 * (nC - 1))
    return sp.csc_matrix((vVec, (np.array(iVec) - 1, np.array(jVec) - 1)), shape=(nR, nC))


def _gene_prolongMat1dim_nearest(nC):
    """downSample_q.m:33-39"""
    nR = 2 * nC
    iVec = list(range(1, nR + 1, 2)) + list(range(2, nR + 1, 2))
    jVec = list(range(1, nC + 1)) * 2
    return sp.csc_matrix(([1.0] * (2 * nC), (np.array(iVec) - 1, np.array(jVec) - 1)), shape=(nR, nC))


def _restri(P):
    """transpose(P ./ sum(P, 1))  (downSample_q.m:10-12)"""
    colsum = np.asarray(P.sum(axis=0)).ravel()
    return (P @ sp.diags(1.0 / colsum)).T.tocsr()


def downSample_q(nt, nx, ny, q, log_mean=False):
    """socp/wdot2d/utils/downSample_q.m:4-21 with the three Kronecker prolongation matrices formed explicitly as sparse
    matrices, like the reference does; log_mean=True is downSample_barrier.m:4-21 (restriction of log(weight), exp)."""
    nt2, nx2, ny2 = (nt + 1) // 2, (nx + 1) // 2, (ny + 1) // 2
    lin, near = _gene_prolongMat1dim_linear, _gene_prolongMat1dim_nearest
    ProlongT = sp.kron(sp.kron(near(nt2 - 1), lin(nx2)), lin(ny2), format="csc")
    ProlongX = sp.kron(sp.kron(lin(nt2), near(nx2 - 1)), lin(ny2), format="csc")
    ProlongY = sp.kron(sp.kron(lin(nt2), lin(nx2)), near(ny2 - 1), format="csc")
    bxInd = (nt - 1) * nx * ny + 1
    byInd = bxInd + nt * (nx - 1) * ny
    v = np.log(q) if log_mean else np.asarray(q, dtype=np.float64)
    out = np.concatenate([_restri(ProlongT) @ v[:bxInd - 1], _restri(ProlongX) @ v[bxInd - 1:byInd - 1],
                          _restri(ProlongY) @ v[byInd - 1:]])
    return np.power(np.e, out) if log_mean else out


def downSample_barrier(nt, nx, ny, weight):
    return downSample_q(nt, nx, ny, weight, log_mean=True)


def jump_nextLevel(var, model, rho0, rho1, nt, weight=None):
    """socp/dot2d/utils/jump_nextLevel.m:5-16 (dot1d twin; wdot2d/utils/jump_nextLevel.m with weight)"""
    one_d = not hasattr(model, "ny")
    cdims = (model.nx, model.nt) if one_d else (model.ny, model.nx, model.nt)
    phiR = interpolate_phi(var.phi, cdims)                 # :5 interpolate(var, model)
    betaR = interpolate_z(var.beta, cdims)
    var_init, modelR = initialize(rho0, rho1, nt)          # :9
    var.phi, var.beta = phiR, betaR                        # interpolate() returns the same (handle) object
    var.qInd, var.z = var_init.qInd, var_init.z            # :10-11
    var.q = np.asarray(modelR.grad @ var.phi).ravel()      # :14  q = grad * phi with the sparse matrix of initialize.m
    alpha = var_init.alpha                                 # :15
    nb = np.asfortranarray(-var.beta)
    if one_d:
        mexops.mexBFdConj1d(alpha, nb, modelR.nt, modelR.nx, 1.0)
    else:
        mexops.mexBFdConj(alpha, nb, modelR.nt, modelR.nx, modelR.ny, 1.0)      # :16
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
