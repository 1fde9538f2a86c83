"""Pins for the CPU oracle (the reference has no tests of its own -- SURVEY.md 4/8c):
algebraic invariants that any faithful restatement of the reference operators obeys."""
import numpy as np
import pytest
import scipy.fft as sfft

from oracle import mexops
from oracle.model import (adjust_lagrangianParam, IfAdjustSigma, initialize, initialize_FFTkernel,
                          mirt_dctn, mirt_idctn, oper_poisson, oper_q)

rng = np.random.default_rng(1234)


def _sizes(nt, nx, ny):
    Nz = ny * nx * (nt - 1)
    return Nz, Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt


@pytest.mark.parametrize("nt,nx,ny", [(4, 6, 5), (3, 2, 2), (5, 3, 7), (2, 4, 3)])
def test_bfd_adjoint_and_diagonal(nt, nx, ny):
    Nz, Nq = _sizes(nt, nx, ny)
    s = 0.73
    q = rng.standard_normal(Nq)
    w = np.asfortranarray(rng.standard_normal((Nz, 10)))
    # structurally-zero slots must not receive adjoint contributions either: BFq never writes them
    z = np.zeros((Nz, 10), order="F")
    mexops.mexBFd(z, q, nt, nx, ny, s, 0.0)
    qa = np.zeros(Nq)
    mexops.mexBFdConj(qa, w, nt, nx, ny, s)
    assert abs(np.vdot(z, w) - np.vdot(q, qa)) <= 1e-13 * (np.linalg.norm(z) * np.linalg.norm(w) + 1)
    # F*B*BF is diagonal and equals oper_q - 1   (socp/dot2d/utils/oper_q.m:13-26)
    D, E = 1.3, 1.3 * s
    diag = oper_q((ny, nx, nt), D, E) - 1.0
    e = np.zeros(Nq)
    for k in rng.choice(Nq, size=min(Nq, 25), replace=False):
        e[:] = 0
        e[k] = 1.0
        z[:] = 0
        mexops.mexBFd(z, e, nt, nx, ny, s, 0.0)
        col = np.zeros(Nq)
        mexops.mexBFdConj(col, z, nt, nx, ny, s)
        ref = np.zeros(Nq)
        ref[k] = diag[k]
        np.testing.assert_allclose(col, ref, atol=1e-14)


@pytest.mark.parametrize("nt,nx", [(4, 6), (3, 2), (5, 9)])
def test_bfd1d_adjoint_and_diagonal(nt, nx):
    Nz = nx * (nt - 1)
    Nq = Nz + (nx - 1) * nt
    s = 1.21
    q = rng.standard_normal(Nq)
    w = np.asfortranarray(rng.standard_normal((Nz, 6)))
    z = np.zeros((Nz, 6), order="F")
    mexops.mexBFd1d(z, q, nt, nx, s, 0.0)
    qa = np.zeros(Nq)
    mexops.mexBFdConj1d(qa, w, nt, nx, s)
    assert abs(np.vdot(z, w) - np.vdot(q, qa)) <= 1e-13 * (np.linalg.norm(z) * np.linalg.norm(w) + 1)
    diag = oper_q((nx, nt), 1.0, s) - 1.0
    for k in range(Nq):
        e = np.zeros(Nq)
        e[k] = 1
        z[:] = 0
        mexops.mexBFd1d(z, e, nt, nx, s, 0.0)
        col = np.zeros(Nq)
        mexops.mexBFdConj1d(col, z, nt, nx, s)
        assert abs(col[k] - diag[k]) < 1e-14 and abs(col.sum() - diag[k]) < 1e-13


def test_bfd_dF_and_boundary_slots():
    nt, nx, ny = 3, 4, 5
    Nz, Nq = _sizes(nt, nx, ny)
    q = rng.standard_normal(Nq)
    z = np.full((Nz, 10), 7.0, order="F")          # sentinel: unwritten slots keep it
    mexops.mexBFd(z, q, nt, nx, ny, 0.5, 2.0)
    zz = z.reshape((ny, nx, nt - 1, 10), order="F")
    np.testing.assert_allclose(zz[..., 0] + zz[..., 9], 4.0, atol=1e-15)
    assert np.all(zz[:, 0, :, 1] == 7.0) and np.all(zz[:, 0, :, 3] == 7.0)       # x-1/2 at x=1
    assert np.all(zz[:, -1, :, 2] == 7.0) and np.all(zz[:, -1, :, 4] == 7.0)     # x+1/2 at x=nx
    assert np.all(zz[0, :, :, 5] == 7.0) and np.all(zz[0, :, :, 7] == 7.0)       # y-1/2 at y=1
    assert np.all(zz[-1, :, :, 6] == 7.0) and np.all(zz[-1, :, :, 8] == 7.0)     # y+1/2 at y=ny
    assert np.count_nonzero(z == 7.0) == 2 * (nt - 1) * (2 * ny + 2 * nx)


def test_proj_soc_properties():
    M, K = 500, 10
    x = np.asfortranarray(rng.standard_normal((M, K)) * 3)
    p = np.zeros_like(x, order="F")
    mexops.mexProjSoc(p, x)
    nrm = np.linalg.norm(p[:, 1:], axis=1)
    assert np.all(p[:, 0] >= nrm - 1e-12)                       # cone membership
    p2 = np.zeros_like(x, order="F")
    mexops.mexProjSoc(p2, p)
    apex = np.all(p == 0, axis=1)                               # an all-zero row re-projects to NaN (0/0), as in the reference
    assert apex.any() and np.all(np.isnan(p2[apex]))
    np.testing.assert_allclose(p2[~apex], p[~apex], atol=1e-12)  # idempotent
    # Moreau: x = P_K(x) + P_{-K}(x) with <P_K x, x - P_K x> = 0
    np.testing.assert_allclose(np.sum(p * (x - p), axis=1), 0, atol=1e-11)
    # closed form against the textbook three-case formula
    t, v = x[:, 0], x[:, 1:]
    n = np.linalg.norm(v, axis=1)
    ref = np.where((n <= t)[:, None], x, 0.0)
    mid = (n > np.abs(t))
    c = (t + n) / (2 * n)
    ref[mid, 0] = (c * n)[mid]
    ref[mid, 1:] = (c[:, None] * v)[mid]
    np.testing.assert_allclose(p, ref, atol=1e-12)


def test_proj_soc_edge_cases():
    """n = 0 with x1>0 -> unchanged, x1<0 -> zero row, x1 = 0 -> NaN row."""
    x = np.zeros((4, 6), order="F")
    x[0, 0] = 2.0
    x[1, 0] = -2.0
    x[3] = [1.0, 1.0, 0, 0, 0, 0]                               # on the boundary of the cone
    p = np.empty_like(x, order="F")
    mexops.mexProjSoc(p, x)
    assert np.array_equal(p[0], x[0])
    assert np.all(p[1] == 0)
    assert np.all(np.isnan(p[2]))
    np.testing.assert_allclose(p[3], x[3], atol=1e-15)


@pytest.mark.parametrize("shape", [(8, 5, 4), (7, 3, 9), (16, 12), (5, 1, 6)])
def test_mirt_dct_equals_scipy(shape):
    a = rng.standard_normal(shape)
    np.testing.assert_allclose(mirt_dctn(a), sfft.dctn(a, norm="ortho"), atol=1e-13)
    np.testing.assert_allclose(mirt_idctn(a), sfft.idctn(a, norm="ortho"), atol=1e-13)
    np.testing.assert_allclose(mirt_idctn(mirt_dctn(a)), a, atol=1e-13)


@pytest.mark.parametrize("ny,nx,nt", [(5, 6, 4), (8, 8, 8), (9, 5, 3)])
def test_poisson_inverts_AtA(ny, nx, nt):
    """D^2 A'A * poisson(r) = r - mean(r), mean(phi) = mean(r)/D^2   (SURVEY.md Appendix B)."""
    rho = np.ones((ny, nx))
    var, model = initialize(rho, rho, nt)
    A = model.grad
    D = 0.8
    kernel = D ** 2 * initialize_FFTkernel(nt, nx, ny)
    r = rng.standard_normal(ny * nx * nt)
    for fast in (True, False):
        phi = oper_poisson(kernel, r.reshape((ny, nx, nt), order="F"), fast=fast).ravel(order="F")
        np.testing.assert_allclose(D ** 2 * (A.T @ (A @ phi)), r - r.mean(), atol=1e-9)
        assert abs(phi.mean() - r.mean() / D ** 2) < 1e-12      # zero mode: kernel 0 -> 1, then scaled by D^2


def test_poisson_1d_inverts_AtA():
    nx, nt = 9, 6
    var, model = initialize(np.ones(nx), np.ones(nx), nt)
    A = model.grad
    kernel = initialize_FFTkernel(nt, nx)
    r = rng.standard_normal(nx * nt)
    phi = oper_poisson(kernel, r.reshape((nx, nt), order="F")).ravel(order="F")
    np.testing.assert_allclose(A.T @ (A @ phi), r - r.mean(), atol=1e-9)


def test_sigma_rule_and_cadence():
    """adjust_lagrangianParam.m:14-39,49-60 and solver_socp_inPALM.m:361-379."""
    assert adjust_lagrangianParam(1.0, 1.05) == (1.0, 1.0)
    s, f = adjust_lagrangianParam(1.0, 3.0)
    assert f == pytest.approx(1.28) and s == pytest.approx(1.28)
    s, f = adjust_lagrangianParam(1.0, 1 / 60.0)
    assert f == pytest.approx(0.5)
    s, f = adjust_lagrangianParam(900.0, 100.0)
    assert s == 1e3 and f == pytest.approx(1e3 / 900)
    s, f = adjust_lagrangianParam(1e-3, 1e-9)
    assert s == 1e-3 and f == 1.0
    fired, last = [], -np.inf
    for it in range(1, 700):
        if IfAdjustSigma(it, last):
            fired.append(it)
            last = it
    assert fired[:8] == [1, 4, 7, 10, 13, 16, 19, 25]
    gaps = np.diff(fired)
    assert gaps[-1] == 40 and set(gaps) <= {3, 6, 10, 15, 25, 40}
