"""The algebra behind the division-free tridiagonal kernels of dot-socp_amd/csrc/tri.hip (round 4), restated in numpy
and held against dense solves: closed-form pivots piv_t = r N_{t+1} / N_t, the two running sums H_t = g_t + rho H_{t-1},
G_t = sum rho^s g_s from which both sweeps of a column come (tri_ends / tri_last / tri_spike / k_tri_final's backward
sweep), for blocks that start / end on a global Neumann row or couple to a neighbour slab.  A CPU test of the device
code's mathematics -- the kernels themselves are checked against scipy's DCT and the single slab in the GPU suite."""
import numpy as np
import pytest


def _block(ap, n, first, last):
    A = np.zeros((n, n))
    for t in range(n):
        A[t, t] = ap + 2 - (1 if first and t == 0 else 0) - (1 if last and t == n - 1 else 0)
        if t > 0:
            A[t, t - 1] = -1
        if t < n - 1:
            A[t, t + 1] = -1
    return A


def _coef(ap):                      # tri_coef
    s = np.sqrt(ap * (1 + 0.25 * ap))
    rm1 = 0.5 * ap + s
    r = 1 + rm1
    rho = 1 / r
    return dict(rho=rho, rho2=rho * rho, r=r, n0d=(rm1 * rho) * (1 + rho))


def _ends(c, bnd, n, rn1):          # tri_ends
    pe = c["rho"] if bnd else c["rho2"]
    s = 1.0 if bnd else -1.0
    N0 = (1 + c["rho"]) if bnd else c["n0d"]
    N1 = N0 if n == 1 else 1 + s * (rn1 * rn1) * pe
    return N0, N1, c["n0d"] + c["rho2"] * N1, s, pe


def _last(c, D, N1, Nn, bnd):       # tri_last
    return D / (c["r"] * Nn - N1) if bnd else c["rho"] * D / Nn


def _piece(c, g, xl, xr, first, last):
    """k_tri_local (Gf, Gl), tri_spike (vf, vl, wf, wl) and k_tri_final_reg (x) for one block"""
    n = len(g)
    rn1 = c["rho"] ** (n - 1)
    f0, f1, fn, sf, pef = _ends(c, first, n, rn1)
    b0, b1, bn, sb, peb = _ends(c, last, n, rn1)
    H = G = 0.0
    pw = 1.0
    Dv = np.zeros(n)
    for t in range(n):
        H = g[t] + c["rho"] * H
        G += pw * g[t]
        Dv[t] = H + (sf * pef * pw) * G
        if t + 1 < n:
            pw *= c["rho"]
    Gl = _last(c, Dv[-1], f1, fn, last)
    Gf = _last(c, G + (sb * peb * rn1) * H, b1, bn, first)
    cl = xl * f0
    xn = _last(c, (Dv[-1] + cl * pw) + xr * f1, f1, fn, last)
    x = np.zeros(n)
    x[-1] = xn
    Nt1 = f1
    for t in range(n - 2, -1, -1):
        pw *= c["r"]
        Nt = f0 if t == 0 else 1 + sf * (pw * pw) * pef
        xn = c["rho"] * ((Dv[t] + cl * pw) + Nt * xn) / Nt1
        x[t] = xn
        Nt1 = Nt
    if last:
        den = c["r"] * fn - f1
        vl, wl = rn1 * f0 / den, f1 / den
    else:
        vl, wl = rn1 * c["rho"] * f0 / fn, c["rho"] * f1 / fn
    if first:
        den = c["r"] * bn - b1
        wf, vf = rn1 * b0 / den, b1 / den
    else:
        wf, vf = rn1 * c["rho"] * b0 / bn, c["rho"] * b1 / bn
    return Gf, Gl, x, (vf, vl, wf, wl)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_closed_form_sweeps_against_dense_solves(seed):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for _ in range(1500):
        n = int(rng.integers(1, 70))
        ap = float(10 ** rng.uniform(-4, 4))        # a' = (CY + CX) / (nt - 1)^2: 6e-4 .. 5e5 on the grids of BASELINE.json
        first, last = bool(rng.integers(2)), bool(rng.integers(2))
        if n == 1 and first and last:
            continue
        A = _block(ap, n, first, last)
        g = rng.standard_normal(n)
        c = _coef(ap)
        xl, xr = rng.standard_normal(2)
        Gf, Gl, x, (vf, vl, wf, wl) = _piece(c, g, xl, xr, first, last)
        sol = np.linalg.solve(A, g)
        g2 = g.copy()
        g2[0] += xl
        g2[-1] += xr
        sol2 = np.linalg.solve(A, g2)
        Ai = np.linalg.inv(A)
        e = max(max(abs(Gf - sol[0]), abs(Gl - sol[-1])) / np.max(np.abs(sol)),
                np.max(np.abs(x - sol2)) / np.max(np.abs(sol2)),
                max(abs(vf - Ai[0, 0]), abs(vl - Ai[-1, 0]), abs(wf - Ai[0, -1]), abs(wl - Ai[-1, -1])) / np.max(np.abs(Ai)))
        worst = max(worst, e)
    assert worst <= 2e-11, worst           # observed 2e-12: the conditioning of the smallest modes (a' ~ 1e-4, n ~ 70)


def test_partitioned_solve_equals_the_whole_column():
    """NSUB pieces coupled through the reduced system (k_tri_reduced's sweep, also inside k_tsolve_single) give the solution
    of the whole Neumann column."""
    rng = np.random.default_rng(7)
    for _ in range(200):
        P = int(rng.integers(2, 9))
        sizes = [int(rng.integers(1, 40)) for _ in range(P)]
        if sizes[0] == 1 and P == 1:
            continue
        nt = sum(sizes)
        ap = float(10 ** rng.uniform(-3.5, 3))
        c = _coef(ap)
        g = rng.standard_normal(nt)
        whole = np.linalg.solve(_block(ap, nt, True, True), g)
        off = np.cumsum([0] + sizes)
        loc = []
        for p in range(P):
            gp = g[off[p]:off[p + 1]]
            Gf, Gl, _, spike = _piece(c, gp, 0.0, 0.0, p == 0, p == P - 1)
            vf, vl, wf, wl = spike
            if p == 0:
                vf = vl = 0.0
            if p == P - 1:
                wf = wl = 0.0
            loc.append((Gf, Gl, vf, vl, wf, wl))
        A_, B_, al, ga = [0.0] * P, [0.0] * P, [0.0] * P, [0.0] * P
        for p, (Gf, Gl, vf, vl, wf, wl) in enumerate(loc):
            if p == 0:
                A_[0], B_[0], al[0], ga[0] = Gf, wf, Gl, wl
            else:
                den = 1.0 - vf * ga[p - 1]
                A_[p] = (Gf + vf * al[p - 1]) / den
                B_[p] = wf / den
                al[p] = Gl + vl * (al[p - 1] + ga[p - 1] * A_[p])
                ga[p] = wl + vl * ga[p - 1] * B_[p]
        Fnext = 0.0
        x = np.zeros(nt)
        for p in range(P - 1, -1, -1):
            Fp = A_[p] + B_[p] * Fnext
            Lprev = al[p - 1] + ga[p - 1] * Fp if p > 0 else 0.0
            gp = g[off[p]:off[p + 1]]
            x[off[p]:off[p + 1]] = _piece(c, gp, Lprev, Fnext, p == 0, p == P - 1)[2]
            Fnext = Fp
        assert np.max(np.abs(x - whole)) <= 1e-10 * np.max(np.abs(whole))
