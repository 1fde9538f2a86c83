"""An anchor that does not depend on the reference's code at all: between two Gaussians on the line the optimal
transport is known in closed form -- the displacement interpolation is the Gaussian with mean (1-t) m0 + t m1 and
standard deviation (1-t) s0 + t s1, and the transport cost (the kinetic energy the dynamic formulation minimises) is
W2^2 = (m1 - m0)^2 + (s1 - s0)^2.  The reference's own 1-D example (examples/dot1d/gene_example_gaussian.m:5-21:
m = 0.3 / 0.7, s^2 = 0.01 / 0.0025 on [0, 1]) is such a pair, up to the truncation of the tails at the boundary.
Both the CPU oracle and the GPU path must land on it; this pins WHAT the loop computes, independently of the
line-by-line restatement (which has no fixture from the reference to lean on)."""
import numpy as np
import pytest

from oracle import driver as OD
from oracle.examples import get_example_1d

M0, S0, M1, S1 = 0.3, 0.1, 0.7, 0.05
W2SQ = (M1 - M0) ** 2 + (S1 - S0) ** 2


def _check(rho, Ex, tol_l1, tol_cost):
    nx, nt = rho.shape
    x = np.linspace(0, 1, nx)
    worst = 0.0
    for k, t in enumerate(np.linspace(0, 1, nt)):
        g = np.exp(-0.5 * ((x - ((1 - t) * M0 + t * M1)) / ((1 - t) * S0 + t * S1)) ** 2)
        g /= g.mean()                                    # the examples are normalised to mean 1 (get_example.m:45-46)
        worst = max(worst, float(np.mean(np.abs(rho[:, k] - g))))
    assert worst <= tol_l1, worst
    ok = rho > 1e-8
    cost = float(np.mean(np.where(ok, Ex ** 2 / np.where(ok, rho, 1.0), 0.0)))      # int int m^2 / rho
    assert abs(cost - W2SQ) <= tol_cost * W2SQ, (cost, W2SQ)
    return worst, cost


def test_oracle_reaches_the_analytic_geodesic():
    rho0, rho1 = get_example_1d("gaussian", 129)
    var, model, hist, sigma = OD.solve_single_level(rho0, rho1, 33, dict(tol=1e-5, maxit=8000))
    rho, Ex = OD.recover_RhoE_1d(var, model)
    _check(rho, Ex, tol_l1=0.02, tol_cost=0.005)          # observed: 1.5 % in L1, cost 0.16234 vs 0.1625


@pytest.mark.gpu
def test_gpu_reaches_the_analytic_geodesic():
    import dotsocp_amd as D
    rho0, rho1 = get_example_1d("gaussian", 513)
    out, timeML, histML, hist = D.solver_dotsocp1d(rho0, rho1, 129, 3, dict(tol=1e-5, maxit=20000), "inPALM")
    worst, cost = _check(out["rho"], out["Ex"], tol_l1=0.012, tol_cost=0.004)      # finer grid: closer
    assert D.check_massConservation(out["rho"], 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["inPALM", "ALG2", "PALM", "acc-ADMM"])
def test_gpu_translation_of_a_gaussian_in_2d(method):
    """Two equal isotropic Gaussians well inside the unit square: the optimal map is the translation, the density at
    time t the same Gaussian centred on the straight line between the centres, the cost |shift|^2."""
    import dotsocp_amd as D
    n, nt, s = 129, 33, 0.06
    x = np.linspace(0, 1, n)
    Y, X = np.meshgrid(x, x, indexing="ij")                 # arrays are (ny, nx)

    def bump(cy, cx):
        g = np.exp(-0.5 * (((Y - cy) / s) ** 2 + ((X - cx) / s) ** 2))
        return g / g.mean()

    (cy0, cx0), (cy1, cx1) = (0.3, 0.35), (0.7, 0.6)
    out, timeML, histML, hist = D.solver_dotsocp2d(bump(cy0, cx0), bump(cy1, cx1), nt, 3, dict(tol=1e-5, maxit=20000),
                                                   method)
    assert D.check_massConservation(out["rho"], 1e-2)
    worst = 0.0
    for k, t in enumerate(np.linspace(0, 1, nt)):
        g = bump((1 - t) * cy0 + t * cy1, (1 - t) * cx0 + t * cx1)
        worst = max(worst, float(np.mean(np.abs(out["rho"][:, :, k] - g))))
    assert worst <= 0.05, worst
    ok = out["rho"] > 1e-8
    cost = float(np.mean(np.where(ok, (out["Ex"] ** 2 + out["Ey"] ** 2) / np.where(ok, out["rho"], 1.0), 0.0)))
    w2 = (cy1 - cy0) ** 2 + (cx1 - cx0) ** 2
    assert abs(cost - w2) <= 0.02 * w2, (cost, w2)


@pytest.mark.gpu
@pytest.mark.parametrize("ny,nx,nt,levels", [(257, 257, 65, 4), (193, 193, 49, 3)])
def test_gpu_translation_of_a_gaussian_in_2d_at_other_sizes(ny, nx, nt, levels):
    """The same closed-form anchor at a second and a third size (round-3 verdict, item 8): a finer square grid, where the
    discretisation error of density and cost must shrink, and a grid whose lengths are neither powers of two nor 2^k + 1
    (193 = 3 * 64 + 1, 49 = 3 * 16 + 1: the dense DCT product on the matrix cores on every axis; square, because the
    reference's restriction downSample_phi.m:5-34 indexes rows and columns with the row range)."""
    import dotsocp_amd as D
    s = 0.06
    Y, X = np.meshgrid(np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing="ij")

    def bump(cy, cx):
        g = np.exp(-0.5 * (((Y - cy) / s) ** 2 + ((X - cx) / s) ** 2))
        return g / g.mean()

    (cy0, cx0), (cy1, cx1) = (0.3, 0.35), (0.7, 0.6)
    out, timeML, histML, hist = D.solver_dotsocp2d(bump(cy0, cx0), bump(cy1, cx1), nt, levels, dict(tol=1e-5, maxit=20000),
                                                   "inPALM")
    assert D.check_massConservation(out["rho"], 1e-2)
    worst = 0.0
    for k, t in enumerate(np.linspace(0, 1, nt)):
        g = bump((1 - t) * cy0 + t * cy1, (1 - t) * cx0 + t * cx1)
        worst = max(worst, float(np.mean(np.abs(out["rho"][:, :, k] - g))))
    ok = out["rho"] > 1e-8
    cost = float(np.mean(np.where(ok, (out["Ex"] ** 2 + out["Ey"] ** 2) / np.where(ok, out["rho"], 1.0), 0.0)))
    w2 = (cy1 - cy0) ** 2 + (cx1 - cx0) ** 2
    print(f"analytic 2-D anchor {ny}x{nx}x{nt}: worst L1 {worst:.4f}, cost {cost:.5f} vs {w2:.5f}")
    # 129 x 129 x 33 is held to 5 % / 2 %; the finer grid must do better (observed values in DESIGN.md section 5)
    assert worst <= (0.03 if ny == 257 else 0.05), worst
    assert abs(cost - w2) <= (0.012 if ny == 257 else 0.02) * w2, (cost, w2)
