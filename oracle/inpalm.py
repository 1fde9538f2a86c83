"""The inPALM / ALG2 iteration loop of the reference, restated with numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, statement by statement,
  socp/dot2d/algorithms/solver_socp_inPALM.m   (2-D)
  socp/dot1d/algorithms/solver_socp_inPALM.m   (1-D; differs only in dimensionality)
  socp/wdot2d/algorithms/solver_wsocp_inPALM.m (weighted 2-D)
keeping the reference dataflow (sparse A / A', DCT Poisson solve, the three MEX
operators, the same temporaries) so that it can also serve as the timed CPU baseline.
"""
import time

import numpy as np

from . import mexops
from .model import (FnormL2, IfAdjustSigma, UPDATE_RULE, adjust_lagrangianParam,
                    compute_kkt_dot_complement, initialize_FFTkernel, normL2, oper_poisson,
                    oper_q)

TIME_NAMES = ['Step_1_1_FFT', 'Step_1_2_ProjSOC', 'Step_2_Q_Step', 'Step_3_Multiplier',
              'KKT', 'Total_Time', 'Iters']


def _get(opts, name, default=None):
    if isinstance(opts, dict):
        return opts.get(name, default)
    return getattr(opts, name, default)


class InPALMState:
    """All loop-carried state of solver_socp_inPALM.m:11-135, so that the loop can be
    advanced in pieces (bench warm-up / timed region) with identical results."""

    def __init__(self, var, opts, model, weighted=False):
        self.var, self.model = var, model
        self.weighted = weighted
        # :20-37 options (weighted default for checkPrimDualFeas is false: solver_wsocp_inPALM.m:25-29)
        cpdf = _get(opts, "checkPrimDualFeas", None)
        self.checkPrimDualFeas = (not weighted) if cpdf is None else bool(cpdf)
        self.time_limit = _get(opts, "time_limit", 3600)
        self.tau = _get(opts, "tau")
        self.sigma = float(_get(opts, "sigma"))
        self.maxit = int(_get(opts, "maxit"))
        self.tol = _get(opts, "tol")
        self.checkSByS = bool(_get(opts, "ifCheckStepByStep", False))
        self.lastSigmaIt = -np.inf
        # :53-61 scaling
        self.cScale, self.dScale = var.cScale, var.dScale
        self.D, self.E = var.D, var.E
        self.scaleBF = self.E / self.D
        self.scaleD = self.E / self.dScale
        self.use_feasOrg = 0
        self.tol_feasOrg = 5 * self.tol
        # :63-77 rescaling
        self.rescale = 1 if _get(opts, "scaling", False) else 0
        self.maxFeas, self.relGap = np.inf, np.inf
        # :79-97
        self.one_d = not hasattr(model, "ny")
        if self.one_d:
            self.dims = (model.nx, model.nt)
            self.h = 1.0 / (model.nx * model.nt)
        else:
            self.dims = (model.ny, model.nx, model.nt)
            self.h = 1.0 / (model.nx * model.ny * model.nt)
        self.A = model.grad
        self.AT = model.grad.T.tocsc()
        self.weight = model.weight if weighted else None
        self.phi, self.q, self.z = var.phi, var.q, var.z
        self.alpha, self.beta = var.alpha, var.beta
        if self.one_d:
            self.kernel = self.D ** 2 * initialize_FFTkernel(model.nt, model.nx)
        else:
            self.kernel = self.D ** 2 * initialize_FFTkernel(model.nt, model.nx, model.ny)
        self.diagQInv = 1.0 / oper_q(self.dims, self.D, self.E, self.weight)
        # :100-105
        self.norm_c = model.normc
        self.norm_d = model.normd
        self.alpha = self.alpha / self.sigma
        self.beta = np.asfortranarray(self.beta / self.sigma)
        self.c = model.c / self.sigma
        self.sigmaScale = 1.0
        # :107-121
        self.kkt_hist, self.time_hist, self.iter_hist, self.gap_hist = [], [], [], []
        self.stopCondition = [0, 2, 5, 6] if self.checkPrimDualFeas else [0, 2, 5]
        self.times = np.zeros(5)
        # :131-133
        self.z2 = np.zeros_like(self.z, order="F")
        self.q2 = np.zeros_like(self.q)
        self._bfd(self.z2, self.q)
        self.it = 0
        self.stopped = False
        self.elapsed = 0.0
        self.last_kkt = None

    # MEX dispatch (2-D vs 1-D names) -------------------------------------------
    def _bfd(self, z2, q):
        if self.one_d:
            mexops.mexBFd1d(z2, q, self.dims[1], self.dims[0], self.scaleBF, self.scaleD)
        else:
            mexops.mexBFd(z2, q, self.dims[2], self.dims[1], self.dims[0], self.scaleBF, self.scaleD)

    def _bfd_conj(self, q2, w):
        if self.one_d:
            mexops.mexBFdConj1d(q2, w, self.dims[1], self.dims[0], self.scaleBF)
        else:
            mexops.mexBFdConj(q2, w, self.dims[2], self.dims[1], self.dims[0], self.scaleBF)

    def _norms5(self):
        h = self.h
        return (normL2(self.phi, h), normL2(self.q, h), FnormL2(self.z, h),
                self.sigma * normL2(self.alpha, h), self.sigma * FnormL2(self.beta, h))

    # one iteration -----------------------------------------------------------------
    def step(self):
        """solver_socp_inPALM.m:136-325, one pass of the `for it` body.
        Returns True when the loop must break."""
        self.it += 1
        it = self.it
        h, w = self.h, self.weight
        t_start = time.perf_counter()
        # ---- rescaling :138-190 ----
        scaleYes = 0
        if self.rescale >= 3 and it % 100 == 0:
            normPhi, normQ, normZ, normAlpha, normBeta = self._norms5()
            normPhis = max(normPhi, normQ, normZ)
            normAlps = max(normAlpha, normBeta)
            ratio = max(normAlps, normPhis) / min(normAlps, normPhis)
            if ratio > 1.2:
                scaleYes = 1
        if ((self.rescale == 1 and self.maxFeas < 2e-2 and it >= 10 and self.relGap < 5e-2)
                or (self.rescale == 2 and self.maxFeas < 5e-3 and it >= 50 and self.relGap < 1e-2)
                or scaleYes):
            if not scaleYes:
                normPhi, normQ, normZ, normAlpha, normBeta = self._norms5()
                normPhis = max(normPhi, normQ, normZ)
                normAlps = max(normAlpha, normBeta)
            dScale2, cScale2 = normPhis, normAlps
            self.sigma = self.sigma * (cScale2 / dScale2)
            self.c = self.c * dScale2 / cScale2 ** 2
            self.norm_c = self.norm_c / cScale2
            if not self.weighted:                       # solver_wsocp_inPALM.m has no norm_d (:108,178)
                self.norm_d = self.norm_d / dScale2
            self.alpha = self.alpha * dScale2 / cScale2 ** 2
            self.beta = self.beta * dScale2 / cScale2 ** 2
            self.q = self.q / dScale2
            self.z = self.z / dScale2
            self.dScale = dScale2 * self.dScale
            self.cScale = cScale2 * self.cScale
            self.scaleD = self.E / self.dScale
            self.sigmaScale = self.sigmaScale * (cScale2 / dScale2)
            self._bfd(self.z2, self.q)
            self.rescale += 1
        # ---- step phi :192-195 ----
        t0 = time.perf_counter()
        u = (self.q - self.alpha) if w is None else (w * self.q - self.alpha)
        rhs = self.AT @ u + self.c
        self.phi = oper_poisson(self.kernel, rhs.reshape(self.dims, order="F")).ravel(order="F")
        t1 = time.perf_counter()
        self.times[0] += t1 - t0
        # ---- step z :197-200 ----
        mexops.mexProjSoc(self.z, np.asfortranarray(self.z2 - self.beta))
        t2 = time.perf_counter()
        self.times[1] += t2 - t1
        # ---- step q :202-207 ----
        tmp_q = self.A @ self.phi
        self._bfd_conj(self.q2, np.asfortranarray(self.z + self.beta))
        if w is None:
            self.q = (tmp_q + self.alpha + self.q2) * self.diagQInv
        else:
            self.q = (w * (tmp_q + self.alpha) + self.q2) * self.diagQInv
        t3 = time.perf_counter()
        self.times[2] += t3 - t2
        # ---- multipliers :209-216 ----
        resi_alpha = (tmp_q - self.q) if w is None else (tmp_q - w * self.q)
        self._bfd(self.z2, self.q)
        resi_beta = self.z - self.z2
        self.alpha = self.alpha + self.tau * resi_alpha
        self.beta = self.beta + self.tau * resi_beta
        t4 = time.perf_counter()
        self.times[3] += t4 - t3
        # ---- KKT :218-324 ----
        brk = False
        adjustSigmaYes = IfAdjustSigma(it, self.lastSigmaIt)
        timed_out = (self.elapsed + (t4 - t_start)) > self.time_limit
        if self.checkSByS or adjustSigmaYes or it == self.maxit or timed_out:
            brk = self._kkt(tmp_q, resi_alpha, resi_beta, adjustSigmaYes, timed_out)
        t5 = time.perf_counter()
        self.times[4] += t5 - t4
        self.elapsed += t5 - t_start
        return brk

    def _kkt(self, tmp_q, resi_alpha, resi_beta, adjustSigmaYes, timed_out):
        h, w, sigma = self.h, self.weight, self.sigma
        D, E, cScale, dScale = self.D, self.E, self.cScale, self.dScale
        self._bfd_conj(self.q2, self.beta)                                   # :225
        Dalpha = self.alpha if w is None else w * self.alpha                 # wsocp :232
        norm_q = normL2(self.q, h)                                           # :227-232
        norm_z = FnormL2(self.z, h)
        norm_Aphi = normL2(tmp_q, h)
        norm_alpha = sigma * normL2(self.alpha, h)
        norm_beta = sigma * FnormL2(self.beta, h)
        norm_FBbeta = sigma * normL2(self.q2, h)
        primFea1 = normL2(resi_alpha, h)                                     # :235-238
        primFea2 = FnormL2(resi_beta, h)
        dualFea1 = sigma * normL2(self.AT @ self.alpha - self.c, h)
        dualFea2 = sigma * normL2(self.q2 + Dalpha, h)
        mexops.mexProjSoc(self.z2, np.asfortranarray(self.z - sigma * self.beta))   # :240
        complem = FnormL2(self.z - self.z2, h)
        self._bfd(self.z2, self.q)                                           # :242
        dotcomplem, normRho, norm_rhoFq, mRhoB, normM, normRhoB = compute_kkt_dot_complement(
            self.q, self.alpha, self.z2, sigma, h, self.dims, self.var.qInd, cScale, dScale, D, E, w)
        kc = 1.0
        den2 = (self.norm_d if not self.weighted else (norm_q + norm_z))     # wsocp :256,265
        KKTResiOrg = np.array([                                              # :247-255
            primFea1 / (kc * D / dScale + norm_Aphi + norm_q),
            primFea2 / (kc * E / dScale + den2),
            dualFea1 / (kc / cScale + self.norm_c),
            complem / (kc * E / dScale + norm_z + norm_beta),
            dualFea2 / (kc / cScale / D + norm_FBbeta + norm_alpha),
            dotcomplem / (kc + normRho + norm_rhoFq),
            mRhoB / (kc + normM + normRhoB)])
        KKTResi = np.array([                                                 # :256-262
            primFea1 / (kc + norm_Aphi + norm_q),
            primFea2 / (kc + den2),
            dualFea1 / (kc + self.norm_c),
            complem / (kc + norm_z + norm_beta),
            dualFea2 / (kc + norm_FBbeta + norm_alpha)])
        qa = self.q if w is None else w * self.q                             # wsocp :272
        priVal = (sigma * cScale * dScale * h) * np.dot(qa, self.alpha)      # :265-267
        dualVal = (sigma * cScale * dScale * h) * np.dot(self.c, self.phi)
        pdGap = abs(priVal - dualVal) / (1 + abs(priVal) + abs(dualVal))
        self.kkt_hist.append(KKTResiOrg)                                     # :270-274
        self.time_hist.append(self.elapsed)
        self.iter_hist.append(self.it)
        self.gap_hist.append(pdGap)
        self.last_kkt = dict(KKTResiOrg=KKTResiOrg, KKTResi=KKTResi, priVal=priVal,
                             dualVal=dualVal, pdGap=pdGap, sigma=sigma)
        if np.max(KKTResiOrg[self.stopCondition]) < self.tol or timed_out:   # :287-290
            return True
        if np.max(KKTResi) < self.tol_feasOrg:                               # :293-295
            self.use_feasOrg = 1
        if adjustSigmaYes:                                                   # :298-316
            self.lastSigmaIt = self.it
            if self.use_feasOrg:
                resiPri, resiDual = max(KKTResiOrg[[0, 1]]), max(KKTResiOrg[[2, 4]])
            else:
                resiPri, resiDual = max(KKTResi[[0, 1]]), max(KKTResi[[2, 4]])
            self.sigma, factor = adjust_lagrangianParam(self.sigma, resiPri / resiDual, UPDATE_RULE)
            if factor != 1:
                self._apply_sigma_factor(factor)
        if self.rescale > 0:                                                # :319-322
            self.maxFeas = np.max(KKTResi)
            self.relGap = pdGap
        return False

    def _apply_sigma_factor(self, factor):
        """solver_socp_inPALM.m:311-315"""
        self.alpha = self.alpha / factor
        self.beta = self.beta / factor
        self.c = self.c / factor

    def run(self, n_iters=None):
        """Advance until break / maxit (or by n_iters iterations)."""
        done = 0
        while self.it < self.maxit and not self.stopped:
            if n_iters is not None and done >= n_iters:
                break
            if self.step():
                self.stopped = True
            done += 1
        return done

    def finish(self):
        """solver_socp_inPALM.m:328-357: write the iterates back and build outputs."""
        var = self.var
        var.name = 'Inexact Proximal ALM'
        var.phi, var.q, var.z = self.phi, self.q, self.z
        var.alpha = self.sigma * self.alpha
        var.beta = self.sigma * self.beta
        var.time = dict(zip(TIME_NAMES, list(self.times) + [self.elapsed, self.it]))
        var.cScale, var.dScale, var.D, var.E = self.cScale, self.dScale, self.D, self.E
        k = len(self.kkt_hist)
        runHist = dict(kkt=np.array(self.kkt_hist).reshape(k, 7), time=np.array(self.time_hist),
                       iter=np.array(self.iter_hist), pdGap=np.array(self.gap_hist), len=k)
        return runHist, self.sigma / self.sigmaScale


def solver_socp_inPALM(var, opts, model):
    """[runHist, sigma] = solver_socp_inPALM(var, opts, model)
    (socp/dot2d/algorithms/solver_socp_inPALM.m:1, socp/dot1d/... :1)."""
    st = InPALMState(var, opts, model, weighted=False)
    st.run()
    return st.finish()


def solver_wsocp_inPALM(var, opts, model):
    """socp/wdot2d/algorithms/solver_wsocp_inPALM.m:1"""
    st = InPALMState(var, opts, model, weighted=True)
    st.run()
    return st.finish()
