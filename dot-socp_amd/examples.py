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


def _grid_YX(nx, ny):
    """`[Y, X] = meshgrid(linspace(0,1,nx), linspace(0,1,ny))` of the generators below: arrays of shape (ny, nx),
    Y running along the columns (length nx), X along the rows (length ny)."""
    Y, X = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    return Y, X


def _bump(X, Y, a, b, sigma):
    return np.exp(-((X - a) ** 2 + (Y - b) ** 2) / (2 * sigma ** 2))


def _four_bumps(X, Y, sigma, mu1=0.25):
    mu2 = 1 - mu1
    return (_bump(X, Y, mu1, mu1, sigma) + _bump(X, Y, mu1, mu2, sigma) + _bump(X, Y, mu2, mu1, sigma)
            + _bump(X, Y, mu2, mu2, sigma))


def gene_example2(nx, ny):
    """examples/dot2d/gene_example2.m:4-19 (Example 5.2): one wide Gaussian -> four narrow ones."""
    Y, X = _grid_YX(nx, ny)
    return _bump(X, Y, 0.25, 0.25, 0.1), _four_bumps(X, Y, 0.05)


def gene_example3(nx, ny):
    """examples/dot2d/gene_example3.m:3-18 (Example 5.3): exp(exp(-3|x-a| - 5|y-b|)) -> four Gaussians."""
    Y, X = _grid_YX(nx, ny)
    rho0 = np.exp(np.exp(-3 * np.abs(X - 0.25) - 5 * np.abs(Y - 0.25)))
    return rho0, _four_bumps(X, Y, 0.05)


def gene_example4(nx, ny):
    """examples/dot2d/gene_example4.m:4-16 (Example 5.4): quartic bowl -> four Gaussians."""
    Y, X = _grid_YX(nx, ny)
    return (X - 0.5) ** 4 + (Y - 0.5) ** 4, _four_bumps(X, Y, 0.05)


_DIRAC_X = [0.8323, 0.5339, 0.4031, 0.6536, 0.8200, 0.4918, 0.5108, 0.6082, 0.4633, 0.1500, 0.7227, 0.4967, 0.5318,
            0.6625, 0.4309, 0.1076, 0.3052, 0.4113, 0.4955, 0.4485, 0.5031, 0.7529, 0.4723, 0.3668, 0.4848, 0.5474,
            0.3867, 0.3192, 0.0676, 0.2382]
_DIRAC_Y = [0.4477, 0.6033, 0.4264, 0.5378, 0.8026, 0.7535, 0.3472, 0.2628, 0.4023, 0.4676, 0.4535, 0.5105, 0.5903,
            0.6705, 0.5134, 0.4471, 0.6960, 0.5068, 0.5040, 0.5468, 0.2641, 0.1783, 0.2195, 0.3484, 0.5056, 0.3925,
            0.4511, 0.2659, 0.4157, 0.8016]


def gene_example7(nx, ny):
    """examples/dot2d/gene_example7.m:20-44 (Example 5.7): a Gaussian -> thirty single-node masses (the instance of
    random points listed in the file; both indices are clipped to [1, nx] there, :38-39)."""
    hx, hy = 1.0 / (nx - 1), 1.0 / (ny - 1)
    xx, yy = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    rho0 = _bump(xx, yy, 0.5, 0.5, 0.1)
    rho1 = np.zeros((ny, nx))
    rnd = lambda v: np.floor(np.asarray(v) + 0.5).astype(int)          # MATLAB round on positive numbers
    # row index from the "X" list, column index from the "Y" list, as in the file (which clips both to nx and is
    # only used on square grids; here each is clipped to its own extent so that other shapes stay in bounds)
    ix = np.clip(rnd(np.array(_DIRAC_X) / hx), 1, ny) - 1
    iy = np.clip(rnd(np.array(_DIRAC_Y) / hy), 1, nx) - 1
    rho1[ix, iy] = 1.0
    return rho0, rho1


def gene_exampleCircle(nx, ny):
    """examples/dot2d/gene_exampleCircle.m:4-24: indicator of a disc -> indicator of a disc."""
    xx, yy = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    rho0 = ((xx - 0.25) ** 2 + (yy - 0.75) ** 2 < 0.25 ** 2).astype(np.float64)
    rho1 = ((xx - 0.75) ** 2 + (yy - 0.25) ** 2 < 0.25 ** 2).astype(np.float64)
    return rho0, rho1


def gene_exampleCircle2(nx, ny):
    """examples/wdot2d/gene_exampleCircle2.m:3-31: one disc -> two smaller discs (used with a rectangular barrier)."""
    xx, yy = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    r1, r2, r3 = 5 / 40, 4 / 40, 3 / 40
    disc = lambda cx, cy, r: (xx - cx) ** 2 + (yy - cy) ** 2 < r ** 2
    rho0 = disc(r1 / 2 + 0.1, 0.475, r1).astype(np.float64)
    rho1 = (disc(r2 / 2 + 0.1, 0.95 - r2, r2) | disc(r3 / 2 + 0.1, r3 + 0.05, r3)).astype(np.float64)
    return rho0, rho1


def _truncated_bumps(nx, ny, c1, c2, r, sigma):
    Y, X = _grid_YX(nx, ny)
    rho0 = _bump(X, Y, c1[0], c1[1], sigma)
    rho0[(X - c1[0]) ** 2 + (Y - c1[1]) ** 2 > r ** 2] = 0
    rho1 = _bump(X, Y, c2[0], c2[1], sigma)
    rho1[(X - c2[0]) ** 2 + (Y - c2[1]) ** 2 > r ** 2] = 0
    return rho0, rho1


def gene_example6(nx, ny):
    """examples/wdot2d/gene_example6.m:3-27 (Example 5.6, the maze): truncated Gaussians in two corners."""
    return _truncated_bumps(nx, ny, (0.925, 0.075), (0.075, 0.925), 0.09, 0.09 / 3)


def gene_exampleLoveHeart(nx, ny):
    """examples/wdot2d/gene_exampleLoveHeart.m:3-27"""
    return _truncated_bumps(nx, ny, (0.7, 0.3), (0.345, 0.625), 0.09, 0.09 / 3)


def gene_exampleMaze14(nx, ny):
    """examples/wdot2d/gene_exampleMaze14.m:3-27"""
    return _truncated_bumps(nx, ny, (0.075, 0.075), (0.925, 0.925), 0.075, 0.075 / 2)


_EXAMPLES_2D = {"example1": gene_example1, "example2": gene_example2, "example3": gene_example3,
                "example4": gene_example4, "example7": gene_example7, "circle": gene_exampleCircle,
                # examples/wdot2d/get_example.m:7-25 adds:
                "circle2": gene_exampleCircle2, "example6": gene_example6, "maze14": gene_exampleMaze14,
                "love-heart": gene_exampleLoveHeart}


def get_example_2d(problem, nx, ny, lowerBound=0.0):
    """examples/dot2d/get_example.m:24-48 and examples/wdot2d/get_example.m:7-31: the closed-form generators
    ("example5" and "DOTmark_4stitch" read image files and are not restated)."""
    if problem not in _EXAMPLES_2D:
        raise ValueError("Novalid input: 'Problem'")
    rho0, rho1 = _EXAMPLES_2D[problem](nx, ny)
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


def gene_example_box(nx):
    """examples/dot1d/gene_example_box.m:4-11: indicator of [0.1, 0.5] -> indicator of [0.85, 0.95]."""
    x = np.linspace(0, 1, nx)
    return ((x >= 0.1) & (x <= 0.5)).astype(np.float64), ((x >= 0.85) & (x <= 0.95)).astype(np.float64)


def get_example_1d(problem, nx, lowerBound=0.0):
    """examples/dot1d/get_example.m:1-22"""
    if problem == "gaussian":
        rho0, rho1 = gene_example_gaussian(nx)
    elif problem == "box":
        rho0, rho1 = gene_example_box(nx)
    else:
        raise ValueError("Novalid input: 'Problem'")
    rho0 = ((nx / rho0.sum()) * rho0 + lowerBound) / (1 + lowerBound)
    rho1 = ((nx / rho1.sum()) * rho1 + lowerBound) / (1 + lowerBound)
    return rho0, rho1


def gene_barrier_of_circle_pillar():
    """examples/wdot2d/gene_barrier_of_circle_pillar.m:4-8"""
    return lambda x, y: (((x >= 0.2) & (x <= 0.25) & (y >= 0.4) & (y <= 1.0))
                         | ((x >= 0.75) & (x <= 0.8) & (y >= 0.0) & (y <= 0.6))
                         | ((x - 0.5) ** 2 + (y - 0.5) ** 2 <= 0.15 ** 2))


def gene_barrier_of_love_heart():
    """examples/wdot2d/gene_barrier_of_love_heart.m:3-15: everything outside a large heart curve or inside a small
    one.  (The maze barriers of example6 / maze14 are read from image files and are not restated.)"""
    def heart(x, y, s):
        u, v = s * (x - 0.5), s * (y - 0.5)
        return (u ** 2 + v ** 2 - 1) ** 3 - u ** 2 * v ** 3
    return lambda x, y: (heart(x, y + 0.05, 2.5) > 0) | (heart(x, y, 15.0) <= 0)


def _gene_weight_radial(nt, nx, ny, fun):
    """examples/wdot2d/gene_weight_circle.m:8-28 / gene_weight_circleInv.m: a radial profile around (0.5, 0.5) on
    the staggered edges, both edge families normalised with ny (nx - 1) as in the files, weight 1 on the time edges."""
    hx, hy = 1.0 / (nx - 1), 1.0 / (ny - 1)
    xStag, xCent = np.linspace(0.5 * hx, 1 - 0.5 * hx, nx - 1), np.linspace(0, 1, nx)
    yStag, yCent = np.linspace(0.5 * hy, 1 - 0.5 * hy, ny - 1), np.linspace(0, 1, ny)
    xx, yy = np.meshgrid(xStag, yCent)
    wX = fun(np.sqrt((xx - 0.5) ** 2 + (yy - 0.5) ** 2))
    wX = wX * (ny * (nx - 1) / wX.sum())
    xx, yy = np.meshgrid(xCent, yStag)
    wY = fun(np.sqrt((xx - 0.5) ** 2 + (yy - 0.5) ** 2))
    wY = wY * (ny * (nx - 1) / wY.sum())
    return np.concatenate([np.ones(ny * nx * (nt - 1)), np.repeat(wX[:, :, None], nt, axis=2).ravel(order="F"),
                           np.repeat(wY[:, :, None], nt, axis=2).ravel(order="F")])


def gene_weight_circle(nt, nx, ny):
    """examples/wdot2d/gene_weight_circle.m: weight = distance from the centre"""
    return _gene_weight_radial(nt, nx, ny, lambda r: r)


def gene_weight_circleInv(nt, nx, ny):
    """examples/wdot2d/gene_weight_circleInv.m: weight = 1 / (0.1 + distance from the centre)"""
    return _gene_weight_radial(nt, nx, ny, lambda r: 1.0 / (0.1 + r))


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
