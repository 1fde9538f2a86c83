"""Time-slab restatement of the 2-D inPALM loop for world_size > 1 (numpy + torch.distributed).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference is a single-process code
("distributed communication backend: none"); the multi-GPU mode of this
repository cuts the grid along t.  This module restates exactly that
decomposition on the CPU -- which layers are owned, which halo layers travel in which direction
at which point of the iteration, the slab<->pencil transposes of the Poisson solve, and which
sums are all-reduced -- with matrix-free numpy operators, so that the plan can be checked
against the single-process oracle (oracle/inpalm.py) under the gloo backend
(tests/test_slabs_gloo.py).  The HIP implementation (dot-socp_amd/csrc/solver.hip) follows the
same plan; its neighbour exchanges are validated on the GPU with all slabs in one process.

Statement references are to socp/dot2d/algorithms/solver_socp_inPALM.m.
"""
import numpy as np
import scipy.fft as sfft

from . import mexops
from .model import IfAdjustSigma, UPDATE_RULE, adjust_lagrangianParam


def slab_range(nt, world, rank):
    """Same rule as dotsocp_slab_range() of the C ABI: nodes dealt evenly, lower ranks first."""
    base, rem = divmod(nt, world)
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


def pencil_range(plane, world, j):
    cut = lambda k: plane if k >= world else 2 * ((plane // 2) * k // world)
    return cut(j), cut(j + 1)


class LocalComm:
    """world = 1: nothing to exchange."""
    rank, world = 0, 1

    def shift(self, dirn, arr):
        return None

    def alltoall(self, pieces):
        return [snd for snd, _ in pieces]

    def allsum(self, v):
        return v


class GlooComm:
    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def shift(self, dirn, arr):
        """send `arr` to rank+dirn (if it exists); return what rank-dirn sent (or None)."""
        t, d = self.torch, self.dist
        to, frm = self.rank + dirn, self.rank - dirn
        reqs, out = [], None
        if 0 <= to < self.world:
            reqs.append(d.isend(t.from_numpy(np.ascontiguousarray(arr)), to))
        if 0 <= frm < self.world:
            out = t.empty(arr.shape, dtype=t.float64)
            reqs.append(d.irecv(out, frm))
        for r in reqs:
            r.wait()
        return None if out is None else out.numpy()

    def alltoall(self, pieces):
        """pieces[j] goes to rank j; returns the list of pieces received (index = source rank).
        Every rank must know the shapes it will receive: they are passed as (send, recv_shape)."""
        t, d = self.torch, self.dist
        reqs, out = [], [None] * self.world
        for j, (snd, shp) in enumerate(pieces):
            if j == self.rank:
                out[j] = snd
                continue
            reqs.append(d.isend(t.from_numpy(np.ascontiguousarray(snd)), j))
            buf = t.empty(shp, dtype=t.float64)
            out[j] = buf
            reqs.append(d.irecv(buf, j))
        for r in reqs:
            r.wait()
        return [o if isinstance(o, np.ndarray) else o.numpy() for o in out]

    def allsum(self, v):
        x = self.torch.from_numpy(np.asarray(v, dtype=np.float64).copy())
        self.dist.all_reduce(x)
        return x.numpy()


def _proj(v):
    """row-wise SOC projection of an (..., 10) array through the oracle's mexProjSoc"""
    shp = v.shape
    a = np.asfortranarray(v.reshape(-1, shp[-1]))
    out = np.empty_like(a, order="F")
    mexops.mexProjSoc(out, a)
    return out.reshape(shp)


class SlabInPALM:
    """State and iteration of ONE slab.  Inputs are the GLOBAL scaled level (var, model, opts as
    produced by oracle.driver.make_level); every rank cuts its own part."""

    def __init__(self, var, opts, model, comm, tsolve="tridiag"):
        """tsolve: how the Poisson solve crosses the slabs along t -- "tridiag": partitioned tridiagonal systems
        (the device default, csrc/tri.hip), "dct": slab <-> pencil transposes around the t-axis DCT."""
        self.comm = comm
        self.tsolve = tsolve
        r, w = comm.rank, comm.world
        ny, nx, nt = model.ny, model.nx, model.nt
        self.ny, self.nx, self.nt = ny, nx, nt
        self.t0, t1 = slab_range(nt, w, r)
        self.ntl = t1 - self.t0
        self.first, self.last = self.t0 == 0, t1 == nt
        self.ncl = self.ntl - 1 if self.last else self.ntl
        t0, ntl, ncl = self.t0, self.ntl, self.ncl
        F = lambda a, shp: np.asarray(a).reshape(shp, order="F")
        qi = var.qInd
        self.phi = F(var.phi, (ny, nx, nt))[:, :, t0:t1].copy()
        self.c = F(model.c, (ny, nx, nt))[:, :, t0:t1].copy()

        def split(v):
            return (F(v[:qi.bx], (ny, nx, nt - 1))[:, :, t0:t0 + ncl].copy(),
                    F(v[qi.bx:qi.by], (ny, nx - 1, nt))[:, :, t0:t1].copy(),
                    F(v[qi.by:], (ny - 1, nx, nt))[:, :, t0:t1].copy())
        self.q0, self.qbx, self.qby = split(var.q)
        self.a0, self.abx, self.aby = split(var.alpha)
        self.z = F(var.z, (ny, nx, nt - 1, 10))[:, :, t0:t0 + ncl].copy()
        self.beta = F(var.beta, (ny, nx, nt - 1, 10))[:, :, t0:t0 + ncl].copy()
        # scalars (:20-77,100-105)
        g = lambda k, d=None: opts.get(k, d)
        self.tau, self.sigma, self.maxit, self.tol = g("tau"), float(g("sigma")), int(g("maxit")), g("tol")
        self.checkSByS = bool(g("ifCheckStepByStep", False))
        self.cScale, self.dScale, self.D, self.E = var.cScale, var.dScale, var.D, var.E
        self.rescale = 1 if g("scaling", False) else 0
        self.maxFeas = self.relGap = np.inf
        self.use_feasOrg, self.tol_feasOrg = 0, 5 * self.tol
        self.h = 1.0 / (nx * ny * nt)
        self.norm_c, self.norm_d = model.normc, model.normd
        self.a0, self.abx, self.aby = self.a0 / self.sigma, self.abx / self.sigma, self.aby / self.sigma
        self.beta = self.beta / self.sigma
        self.c = self.c / self.sigma
        self.sigmaScale, self.lastSigmaIt, self.it = 1.0, -np.inf, 0
        self.hist = []
        # spectral kernel of this rank's pencil (columns l0..l1 of the ny*nx columns, all kt)
        self.l0, self.l1 = pencil_range(ny * nx, w, r)
        CT = (2.0 * (nt - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(nt) / nt))
        CX = (2.0 * (nx - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(nx) / nx))
        CY = (2.0 * (ny - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(ny) / ny))
        cols = np.arange(self.l0, self.l1)
        lam = (CY[cols % ny] + CX[cols // ny])[:, None] + CT[None, :]
        lam[lam == 0] = 1.0
        self.lam = lam
        self._coef()
        self._exchange_q_halo()

    # ------------------------------------------------------------------------------------
    def _coef(self):
        self.s = self.E / self.D
        self.sf = self.s / np.sqrt(2.0)
        self.dF = self.E / self.dScale
        D = self.D
        self.at, self.ax, self.ay = D * (self.nt - 1), D * (self.nx - 1), D * (self.ny - 1)
        s2 = (self.E / self.D) ** 2
        self.c1, self.c2 = 1 + 2 * s2, 1 + s2

    def _exchange_q_halo(self):
        """E2: first owned bx / by layer of every slab -> halo layer of its LEFT neighbour"""
        self.qbx_h = self.comm.shift(-1, self.qbx[:, :, 0])
        self.qby_h = self.comm.shift(-1, self.qby[:, :, 0])

    def _ext(self, a, halo):
        return a if halo is None else np.concatenate([a, halo[:, :, None]], axis=2)

    def _z2(self):
        """B F q + d on the owned cells (needs the bx / by halo layer unless this is the last slab)"""
        ny, nx, ncl = self.ny, self.nx, self.ncl
        bx, by = self._ext(self.qbx, self.qbx_h), self._ext(self.qby, self.qby_h)
        z2 = np.zeros((ny, nx, ncl, 10))
        z2[..., 0] = self.dF - self.s * self.q0
        z2[..., 9] = self.dF + self.s * self.q0
        for dt in (0, 1):
            z2[:, 1:, :, 1 + 2 * dt] = self.sf * bx[:, :, dt:dt + ncl]
            z2[:, :-1, :, 2 + 2 * dt] = self.sf * bx[:, :, dt:dt + ncl]
            z2[1:, :, :, 5 + 2 * dt] = self.sf * by[:, :, dt:dt + ncl]
            z2[:-1, :, :, 6 + 2 * dt] = self.sf * by[:, :, dt:dt + ncl]
        return z2

    def _adjoint(self, w):
        """F* B* w on the owned entries; the partial sums of the last owned cell for the first edge
        layer of the RIGHT neighbour travel as E4 (two layers) and are added to this slab's layer 0."""
        ny, nx, ncl, ntl = self.ny, self.nx, self.ncl, self.ntl
        g0 = self.s * (w[..., 9] - w[..., 0])
        gx = np.zeros((ny, nx - 1, ncl + 1))
        gy = np.zeros((ny - 1, nx, ncl + 1))
        gx[:, :, :ncl] += w[:, 1:, :, 1] + w[:, :-1, :, 2]
        gx[:, :, 1:] += w[:, 1:, :, 3] + w[:, :-1, :, 4]
        gy[:, :, :ncl] += w[1:, :, :, 5] + w[:-1, :, :, 6]
        gy[:, :, 1:] += w[1:, :, :, 7] + w[:-1, :, :, 8]
        tx = self.comm.shift(+1, gx[:, :, ncl])
        ty = self.comm.shift(+1, gy[:, :, ncl])
        if tx is not None:
            gx[:, :, 0] += tx
            gy[:, :, 0] += ty
        return g0, self.sf * gx[:, :, :ntl], self.sf * gy[:, :, :ntl]

    def _AT(self, u0, ux, uy, u0_prev):
        """A' u on the owned nodes (Neumann: missing staggered neighbours dropped)"""
        ny, nx, ntl, ncl = self.ny, self.nx, self.ntl, self.ncl
        r = np.zeros((ny, nx, ntl))
        r[:, :, 1:ntl] += self.at * u0[:, :, :ntl - 1]      # cell t-1/2 of node t (the last cell of a
        if u0_prev is not None:                             # non-last slab feeds the right neighbour: E1)
            r[:, :, 0] += self.at * u0_prev
        r[:, :, :ncl] -= self.at * u0
        r[:, 1:, :] += self.ax * ux
        r[:, :-1, :] -= self.ax * ux
        r[1:, :, :] += self.ay * uy
        r[:-1, :, :] -= self.ay * uy
        return r

    def _A(self):
        phi_h = self.comm.shift(-1, self.phi[:, :, 0])          # E3: phi head -> left neighbour
        pe = self._ext(self.phi, phi_h)
        t0 = self.at * (pe[:, :, 1:self.ncl + 1] - pe[:, :, :self.ncl])
        tx = self.ax * (self.phi[:, 1:, :] - self.phi[:, :-1, :])
        ty = self.ay * (self.phi[1:, :, :] - self.phi[:-1, :, :])
        return t0, tx, ty

    # -- the t direction by partitioned tridiagonal systems (mirrors csrc/tri.hip) ------------------------
    def _delta(self, ap, t, n, first, last):
        return ap + 2.0 - (1.0 if (first and t == 0) else 0.0) - (1.0 if (last and t == n - 1) else 0.0)

    def _ends(self, ap, g, n, first, last):
        """first and last entry of A^{-1} g for the slab block (eliminations from both ends); g: (modes, n)"""
        piv, d = self._delta(ap, 0, n, first, last), g[:, 0].copy()
        for t in range(1, n):
            inv = 1.0 / piv
            d = g[:, t] + d * inv
            piv = self._delta(ap, t, n, first, last) - inv
        last_v = d / piv
        piv, d = self._delta(ap, n - 1, n, first, last), g[:, n - 1].copy()
        for t in range(n - 2, -1, -1):
            inv = 1.0 / piv
            d = g[:, t] + d * inv
            piv = self._delta(ap, t, n, first, last) - inv
        return d / piv, last_v

    def _poisson_t_tridiag(self, flat):
        """flat: (ny*nx modes, ntl) after the y, x transforms -> the same array solved along t"""
        ny, nx, nt, w, r = self.ny, self.nx, self.nt, self.comm.world, self.comm.rank
        plane, n = ny * nx, self.ntl
        beta = float((nt - 1) ** 2)
        CX = (2.0 * (nx - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(nx) / nx))
        CY = (2.0 * (ny - 1) ** 2) * (1.0 - np.cos(np.pi * np.arange(ny) / ny))
        modes = np.arange(plane)
        ap_all = (CY[modes % ny] + CX[modes // ny]) / beta
        g = flat / (self.D ** 2 * beta)
        Gf, Gl = self._ends(ap_all, g, n, self.first, self.last)
        sizes = [slab_range(nt, w, j)[1] - slab_range(nt, w, j)[0] for j in range(w)]
        nmax = max(sizes)
        pieces = []
        for j in range(w):                                      # 2 numbers per mode (+ the zero-mode line) to owner j
            l0, l1 = pencil_range(plane, w, j)
            extra = np.zeros(nmax)
            if j == 0:
                extra[:n] = g[0, :]
            pieces.append((np.concatenate([Gf[l0:l1], Gl[l0:l1], extra]), (2 * (self.l1 - self.l0) + nmax,)))
        got = self.comm.alltoall(pieces)
        nl = self.l1 - self.l0
        ap = ap_all[self.l0:self.l1]
        A, B, al, ga = [None] * w, [None] * w, [None] * w, [None] * w
        with np.errstate(divide="ignore", invalid="ignore"):
            for p in range(w):
                first, last, npn = p == 0, p == w - 1, sizes[p]
                e_first = np.zeros((nl, npn)); e_first[:, 0] = 1.0
                e_last = np.zeros((nl, npn)); e_last[:, npn - 1] = 1.0
                vf, vl = self._ends(ap, e_first, npn, first, last)
                wf, wl = self._ends(ap, e_last, npn, first, last)
                if first:
                    vf, vl = 0.0 * vf, 0.0 * vl
                if last:
                    wf, wl = 0.0 * wf, 0.0 * wl
                gf, gl = got[p][:nl], got[p][nl:2 * nl]
                if p == 0:
                    A[0], B[0], al[0], ga[0] = gf, wf, gl, wl
                else:
                    den = 1.0 - vf * ga[p - 1]
                    A[p] = (gf + vf * al[p - 1]) / den
                    B[p] = wf / den
                    al[p] = gl + vl * (al[p - 1] + ga[p - 1] * A[p])
                    ga[p] = wl + vl * ga[p - 1] * B[p]
            back = [None] * w
            Fnext = np.zeros(nl)
            for p in range(w - 1, -1, -1):
                F = A[p] + B[p] * Fnext
                Lprev = al[p - 1] + ga[p - 1] * F if p > 0 else np.zeros(nl)
                back[p] = np.concatenate([Lprev, Fnext, np.zeros(nmax)])
                Fnext = F
        if self.l0 == 0 and nl > 0:                              # the singular (0, 0) mode, whole line
            line = np.concatenate([got[p][2 * nl:2 * nl + sizes[p]] for p in range(w)])
            gbar = line.mean()
            x = np.zeros(nt)
            for t in range(nt - 1):
                gt = line[t] - gbar
                x[t + 1] = x[t] - gt if t == 0 else 2.0 * x[t] - x[t - 1] - gt
            x = x - x.mean() + beta * gbar
            o = 0
            for p in range(w):
                back[p][0] = back[p][nl] = 0.0
                back[p][2 * nl:2 * nl + sizes[p]] = x[o:o + sizes[p]]
                o += sizes[p]
        # the reply of owner j has the length of what was sent to it
        shapes = [(2 * (pencil_range(plane, w, j)[1] - pencil_range(plane, w, j)[0]) + nmax,) for j in range(w)]
        ret = self.comm.alltoall([(back[j], shapes[j]) for j in range(w)])
        xl, xr = np.zeros(plane), np.zeros(plane)
        zero_line = None
        for j in range(w):
            l0, l1 = pencil_range(plane, w, j)
            xl[l0:l1], xr[l0:l1] = ret[j][:l1 - l0], ret[j][l1 - l0:2 * (l1 - l0)]
            if j == 0:
                zero_line = ret[j][2 * (l1 - l0):2 * (l1 - l0) + n]
        # local Thomas solve with the neighbours' interface values on the right-hand side
        gt = g.copy()
        gt[:, 0] += xl
        gt[:, n - 1] += xr
        pinv = np.zeros((plane, n))
        xs = np.zeros((plane, n))
        with np.errstate(divide="ignore", invalid="ignore"):
            piv = self._delta(ap_all, 0, n, self.first, self.last)
            d = gt[:, 0].copy()
            inv = 1.0 / piv
            xs[:, 0], pinv[:, 0] = d * inv, inv
            for t in range(1, n):
                d = gt[:, t] + d * inv
                piv = self._delta(ap_all, t, n, self.first, self.last) - inv
                inv = 1.0 / piv
                xs[:, t], pinv[:, t] = d * inv, inv
            for t in range(n - 2, -1, -1):
                xs[:, t] = xs[:, t] + pinv[:, t] * xs[:, t + 1]
        xs[0, :] = zero_line
        return xs

    def _poisson(self, rhs):
        """dct over y, x on the slab; along t either partitioned tridiagonal systems or slabs -> pencils, dct_t,
        ./kernel, idct_t, pencils -> slabs; idct x, y"""
        ny, nx, nt, w = self.ny, self.nx, self.nt, self.comm.world
        a = sfft.dct(sfft.dct(rhs, axis=0, norm="ortho"), axis=1, norm="ortho")
        flat = a.reshape((ny * nx, self.ntl), order="F")
        if self.tsolve == "tridiag" and w > 1:
            a = self._poisson_t_tridiag(flat).reshape((ny, nx, self.ntl), order="F")
            return sfft.idct(sfft.idct(a, axis=1, norm="ortho"), axis=0, norm="ortho")
        pieces = []
        for j in range(w):
            l0, l1 = pencil_range(ny * nx, w, j)
            tj0, tj1 = slab_range(nt, w, j)
            pieces.append((flat[l0:l1, :], (self.l1 - self.l0, tj1 - tj0)))
        pen = np.concatenate(self.comm.alltoall(pieces), axis=1)            # (columns, all t)
        pen = sfft.idct(sfft.dct(pen, axis=1, norm="ortho") / (self.D ** 2 * self.lam), axis=1, norm="ortho")
        pieces = []
        for j in range(w):
            l0, l1 = pencil_range(ny * nx, w, j)
            tj0, tj1 = slab_range(nt, w, j)
            pieces.append((pen[:, tj0:tj1], (l1 - l0, self.ntl)))
        flat = np.concatenate(self.comm.alltoall(pieces), axis=0)
        a = flat.reshape((ny, nx, self.ntl), order="F")
        return sfft.idct(sfft.idct(a, axis=1, norm="ortho"), axis=0, norm="ortho")

    # ------------------------------------------------------------------------------------
    def _sums(self, names_vals):
        return self.comm.allsum(np.array([np.sum(v) for v in names_vals]))

    def step(self):
        self.it += 1
        it = self.it
        # rescale block (:138-190): norms are all-reduced
        scaleYes = 0

        def norms():
            S = self._sums([self.phi ** 2, self.q0 ** 2, self.qbx ** 2, self.qby ** 2, self.z ** 2,
                            self.a0 ** 2, self.abx ** 2, self.aby ** 2, self.beta ** 2])
            sh = np.sqrt(self.h)
            nP = max(sh * np.sqrt(S[0]), sh * np.sqrt(S[1] + S[2] + S[3]), sh * np.sqrt(S[4]))
            nA = max(self.sigma * sh * np.sqrt(S[5] + S[6] + S[7]), self.sigma * sh * np.sqrt(S[8]))
            return nP, nA
        nP = nA = None
        if self.rescale >= 3 and it % 100 == 0:
            nP, nA = norms()
            if max(nA, nP) / min(nA, nP) > 1.2:
                scaleYes = 1
        if ((self.rescale == 1 and self.maxFeas < 2e-2 and it >= 10 and self.relGap < 5e-2)
                or (self.rescale == 2 and self.maxFeas < 5e-3 and it >= 50 and self.relGap < 1e-2) or scaleYes):
            if not scaleYes:
                nP, nA = norms()
            d2, c2 = nP, nA
            self.sigma *= c2 / d2
            self.c = self.c * d2 / c2 ** 2
            self.norm_c /= c2
            self.norm_d /= d2
            for n in ("a0", "abx", "aby", "beta"):
                setattr(self, n, getattr(self, n) * d2 / c2 ** 2)
            for n in ("q0", "qbx", "qby", "z"):
                setattr(self, n, getattr(self, n) / d2)
            if self.qbx_h is not None:
                self.qbx_h, self.qby_h = self.qbx_h / d2, self.qby_h / d2
            self.dScale *= d2
            self.cScale *= c2
            self.sigmaScale *= c2 / d2
            self._coef()
            self.rescale += 1
        # ---- phi step (:194): E1 = u0 tail -> right neighbour
        u0 = self.q0 - self.a0
        u0_prev = self.comm.shift(+1, u0[:, :, self.ncl - 1])
        rhs = self._AT(u0, self.qbx - self.abx, self.qby - self.aby, u0_prev) + self.c
        self.phi = self._poisson(rhs)
        # ---- z step (:199)
        z2 = self._z2()
        self.z = _proj(z2 - self.beta)
        # ---- q step (:204-206), alpha (:211,214)
        t0, tx, ty = self._A()
        g0, gx, gy = self._adjoint(self.z + self.beta)
        lay = self.t0 + np.arange(self.ntl)
        dinv = np.where((lay == 0) | (lay == self.nt - 1), 1.0 / self.c2, 1.0 / self.c1)[None, None, :]
        q0n = (t0 + self.a0 + g0) * (1.0 / self.c1)
        qbxn = (tx + self.abx + gx) * dinv
        qbyn = (ty + self.aby + gy) * dinv
        r0, rx, ry = t0 - q0n, tx - qbxn, ty - qbyn
        self.q0, self.qbx, self.qby = q0n, qbxn, qbyn
        self.a0, self.abx, self.aby = self.a0 + self.tau * r0, self.abx + self.tau * rx, self.aby + self.tau * ry
        self._exchange_q_halo()                                   # E2
        # ---- beta (:212-215)
        z2 = self._z2()
        resi_beta = self.z - z2
        self.beta = self.beta + self.tau * resi_beta
        # ---- KKT (:220-323)
        adjust = IfAdjustSigma(it, self.lastSigmaIt)
        if self.checkSByS or adjust or it == self.maxit:
            return self._kkt((t0, tx, ty), (r0, rx, ry), resi_beta, z2, adjust)
        return False

    def _kkt(self, Aphi, resa, resi_beta, z2, adjust):
        h, sg = self.h, self.sigma
        D, E, cS, dS = self.D, self.E, self.cScale, self.dScale
        # E5: alpha0 tail -> right neighbour (A' alpha at its first node layer, rho at its first nodes)
        a0_prev = self.comm.shift(+1, self.a0[:, :, self.ncl - 1])
        b0, bxg, byg = self._adjoint(self.beta)                  # includes the beta tails
        dual1 = self._AT(self.a0, self.abx, self.aby, a0_prev) - self.c
        comp = self.z - _proj(self.z - sg * self.beta)
        kappa = sg * cS * D
        rhoT = kappa * self.a0
        rhoFq = rhoT + (dS / D) * self.q0 + np.sum(((dS / E) * z2[..., 1:9]) ** 2, axis=-1) / 4.0
        rhoFq[rhoFq < 0] = 0.0
        prev = np.zeros((self.ny, self.nx)) if a0_prev is None else kappa * a0_prev
        pad = np.concatenate([prev[:, :, None], rhoT], axis=2)
        if self.last:
            pad = np.concatenate([pad, np.zeros((self.ny, self.nx, 1))], axis=2)
        rho = (pad[:, :, :-1] + pad[:, :, 1:]) / 2.0              # owned node layers
        rBx = (dS / D) * ((rho[:, :-1] + rho[:, 1:]) / 2.0 * self.qbx)
        rBy = (dS / D) * ((rho[:-1] + rho[1:]) / 2.0 * self.qby)
        mx, my = kappa * self.abx, kappa * self.aby
        sq = lambda *a: sum(np.sum(x ** 2) for x in a)
        S = self.comm.allsum(np.array([
            sq(self.q0, self.qbx, self.qby), sq(self.z), sq(*Aphi), sq(self.a0, self.abx, self.aby), sq(self.beta),
            sq(b0, bxg, byg), sq(*resa), sq(resi_beta), sq(dual1),
            sq(b0 + self.a0, bxg + self.abx, byg + self.aby), sq(comp), sq(rhoT - rhoFq), sq(rhoT), sq(rhoFq),
            sq(mx - rBx, my - rBy), sq(mx, my), sq(rBx, rBy),
            np.sum(self.q0 * self.a0) + np.sum(self.qbx * self.abx) + np.sum(self.qby * self.aby),
            np.sum(self.c * self.phi)]))
        n = lambda i: np.sqrt(h) * np.sqrt(S[i])
        norm_q, norm_z, norm_Aphi = n(0), n(1), n(2)
        norm_alpha, norm_beta, norm_FB = sg * n(3), sg * n(4), sg * n(5)
        p1, p2, d1, d2, cm = n(6), n(7), sg * n(8), sg * n(9), n(10)
        org = np.array([p1 / (D / dS + norm_Aphi + norm_q), p2 / (E / dS + self.norm_d),
                        d1 / (1 / cS + self.norm_c), cm / (E / dS + norm_z + norm_beta),
                        d2 / (1 / cS / D + norm_FB + norm_alpha), n(11) / (1 + n(12) + n(13)),
                        n(14) / (1 + n(15) + n(16))])
        res = np.array([p1 / (1 + norm_Aphi + norm_q), p2 / (1 + self.norm_d), d1 / (1 + self.norm_c),
                        cm / (1 + norm_z + norm_beta), d2 / (1 + norm_FB + norm_alpha)])
        pri, dual = (sg * cS * dS * h) * S[17], (sg * cS * dS * h) * S[18]
        gap = abs(pri - dual) / (1 + abs(pri) + abs(dual))
        self.hist.append((self.it, org, gap))
        if np.max(org[[0, 2, 5, 6]]) < self.tol:
            return True
        if np.max(res) < self.tol_feasOrg:
            self.use_feasOrg = 1
        if adjust:
            self.lastSigmaIt = self.it
            a, b = (org, org) if self.use_feasOrg else (res, res)
            xi = max(a[[0, 1]]) / max(b[[2, 4]])
            self.sigma, f = adjust_lagrangianParam(self.sigma, xi, UPDATE_RULE)
            if f != 1:
                for nme in ("a0", "abx", "aby", "beta", "c"):
                    setattr(self, nme, getattr(self, nme) / f)
        if self.rescale > 0:
            self.maxFeas, self.relGap = np.max(res), gap
        return False

    def run(self):
        while self.it < self.maxit:
            if self.step():
                break
        return self
