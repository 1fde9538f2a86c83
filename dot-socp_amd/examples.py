"""Synthetic problem generators restated from the reference's examples/ directory.

Host-side input generators for the drivers, demos and bench.py (float64 numpy, MATLAB orientation).
"""
import numpy as np


def _normal2d(x, y, mean, sinv):
    """examples/dot2d/gene_example1.m:14-17"""
    det = sinv[0, 0] * sinv[1, 1] - sinv[0, 1] * sinv[1, 0]
    return np.sqrt(det) / (2 * np.pi) * np.exp(-0.5 * (
        sinv[0, 0] * (x - mean[0]) ** 2 + sinv[0, 1] * (x - mean[0]) * (y - mean[1])
        + sinv[1, 1] * (y - mean[1]) ** 2))


def gene_example1(nx, ny):
    """examples/dot2d/gene_example1.m:5-25 (Example 5.1, Gaussian -> Gaussian).
    Note the generator returns arrays of shape (nx, ny); the solver reads
    ny = size(rho0,1), nx = size(rho0,2) (socp/dot2d/utils/initialize.m:8-9)."""
    mu1, mu2, sigma = 0.25, 0.75, 0.05
    sinv = np.linalg.inv(np.array([[sigma, 0.0], [0.0, sigma]]))
    x = np.repeat(np.linspace(0, 1, nx).reshape(nx, 1), ny, axis=1)
    y = np.repeat(np.linspace(0, 1, ny).reshape(1, ny), nx, axis=0)
    rho0 = _normal2d(x, y, (mu1, mu2), sinv)
    rho1 = _normal2d(x, y, (mu2, mu1), sinv)
    return rho0, rho1


def get_example_2d(problem, nx, ny, lowerBound=0.0):
    """examples/dot2d/get_example.m:24-48 (only "example1" is restated)."""
    if problem != "example1":
        raise ValueError("Novalid input: 'Problem'")
    rho0, rho1 = gene_example1(nx, ny)
    rho0 = ((nx * ny / rho0.sum()) * rho0 + lowerBound) / (1 + lowerBound)
    rho1 = ((nx * ny / rho1.sum()) * rho1 + lowerBound) / (1 + lowerBound)
    return rho0, rho1


def gene_example_gaussian(nx):
    """examples/dot1d/gene_example_gaussian.m:5-21"""
    mu1, mu2 = 0.3, 0.7
    sigma1 = 0.01
    sigma2 = sigma1 / 4
    normal = lambda x, m, sinv: np.sqrt(sinv) / (2 * np.pi) * np.exp(-0.5 * (sinv * (x - m) ** 2))
    x = np.linspace(0, 1, nx)
    return normal(x, mu1, 1 / sigma1), normal(x, mu2, 1 / sigma2)


def get_example_1d(problem, nx, lowerBound=0.0):
    """examples/dot1d/get_example.m:1-22 (only "gaussian" is restated)."""
    if problem != "gaussian":
        raise ValueError("Novalid input: 'Problem'")
    rho0, rho1 = gene_example_gaussian(nx)
    rho0 = ((nx / rho0.sum()) * rho0 + lowerBound) / (1 + lowerBound)
    rho1 = ((nx / rho1.sum()) * rho1 + lowerBound) / (1 + lowerBound)
    return rho0, rho1


def gene_barrier_of_circle_pillar():
    """examples/wdot2d/gene_barrier_of_circle_pillar.m:4-8"""
    return lambda x, y: (((x >= 0.2) & (x <= 0.25) & (y >= 0.4) & (y <= 1.0))
                         | ((x >= 0.75) & (x <= 0.8) & (y >= 0.0) & (y <= 0.6))
                         | ((x - 0.5) ** 2 + (y - 0.5) ** 2 <= 0.15 ** 2))


def get_weight_by_barrier(nx, ny, nt, barrier, barrierWeight=1e6):
    """examples/wdot2d/get_weight_by_barrier.m:8-33.  The MATLAB code evaluates
    barrier(xx', yy') on transposed meshgrids and applies mask' -- i.e. weightX(iy, ix)
    is set where barrier(xStag(ix), yCent(iy)) holds."""
    hx, hy = 1.0 / (nx - 1), 1.0 / (ny - 1)
    xStag = np.linspace(0.5 * hx, 1 - 0.5 * hx, nx - 1)
    xCent = np.linspace(0, 1, nx)
    yStag = np.linspace(0.5 * hy, 1 - 0.5 * hy, ny - 1)
    yCent = np.linspace(0, 1, ny)
    xx, yy = np.meshgrid(xStag, yCent)          # (ny, nx-1)
    weightX = np.ones((ny, nx - 1))
    weightX[barrier(xx, yy) > 0] = barrierWeight
    xx, yy = np.meshgrid(xCent, yStag)          # (ny-1, nx)
    weightY = np.ones((ny - 1, nx))
    weightY[barrier(xx, yy) > 0] = barrierWeight
    weightT = np.ones(ny * nx * (nt - 1))
    wX = np.repeat(weightX[:, :, None], nt, axis=2).ravel(order="F")
    wY = np.repeat(weightY[:, :, None], nt, axis=2).ravel(order="F")
    return np.concatenate([weightT, wX, wY])


def ensure_barrier_validity(rho0, rho1, barrier):
    """examples/wdot2d/ensure_barrier_validity.m:4-14"""
    ny, nx = rho0.shape
    xx, yy = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    b = barrier(xx, yy).astype(np.float64)      # == (barrier(xx',yy'))'
    mask = b > b.mean()
    rho0 = rho0.copy()
    rho1 = rho1.copy()
    rho0[mask] = 0
    rho1[mask] = 0
    rho0 = (nx * ny / rho0.sum()) * rho0
    rho1 = (nx * ny / rho1.sum()) * rho1
    return rho0, rho1, mask
