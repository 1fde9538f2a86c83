"""The proximal ALM loop (PALM) of the reference, restated with numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED (no fixture in the reference, no MATLAB).
Follows socp/dot2d/algorithms/solver_socp_PALM.m statement by statement; the file differs from
solver_socp_inPALM.m by the initial z (:136-138), an extra q-step in front of the phi-step (:196-200),
the z2 refresh in the z-step (:209) and a rescale block that scales tmp_q instead of q (:181-191).
The KKT block (:241-335) is the inPALM one and is inherited.
"""
import time

import numpy as np

from . import mexops
from .inpalm import InPALMState
from .model import IfAdjustSigma, oper_poisson

TIME_NAMES = ['Step_1_Q_Step', 'Step_2_1_FFT', 'Step_2_2_ProjSOC', 'Step_3_Q_Step', 'Step_4_Multiplier', 'KKT',
              'Total_Time', 'Iters']


class PALMState(InPALMState):
    def __init__(self, var, opts, model, weighted=False):
        if weighted or not hasattr(model, "ny"):
            raise ValueError("the reference has PALM for dot2d only (socp/dot2d/algorithms/solver_socp_PALM.m)")
        super().__init__(var, opts, model, weighted=False)
        self.times = np.zeros(6)
        # :136-138 initial var
        self.tmp_q = self.A @ self.phi
        self.z = np.asfortranarray(self.z)
        self._bfd(self.z, self.tmp_q)

    def step(self):
        """solver_socp_PALM.m:141-336"""
        self.it += 1
        it = self.it
        t_start = time.perf_counter()
        # ---- rescaling :142-194 ----
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
            self.norm_d = self.norm_d / dScale2
            self.alpha = self.alpha * dScale2 / cScale2 ** 2
            self.beta = self.beta * dScale2 / cScale2 ** 2
            self.z = self.z / dScale2                       # q is not scaled (:181): it is recomputed below
            self.dScale = dScale2 * self.dScale
            self.cScale = cScale2 * self.cScale
            self.scaleD = self.E / self.dScale
            self.sigmaScale = self.sigmaScale * (cScale2 / dScale2)
            self.tmp_q = self.tmp_q / dScale2               # :191
            self.rescale += 1
        # ---- first q-step :196-200 ----
        t0 = time.perf_counter()
        self._bfd_conj(self.q2, np.asfortranarray(self.z + self.beta))
        self.q = (self.tmp_q + self.alpha + self.q2) * self.diagQInv
        t1 = time.perf_counter()
        self.times[0] += t1 - t0
        # ---- step phi :202-205 ----
        rhs = self.AT @ (self.q - self.alpha) + self.c
        self.phi = oper_poisson(self.kernel, rhs.reshape(self.dims, order="F")).ravel(order="F")
        t2 = time.perf_counter()
        self.times[1] += t2 - t1
        # ---- step z :207-211 ----
        self._bfd(self.z2, self.q)
        mexops.mexProjSoc(self.z, np.asfortranarray(self.z2 - self.beta))
        t3 = time.perf_counter()
        self.times[2] += t3 - t2
        # ---- second q-step :213-218 ----
        self.tmp_q = self.A @ self.phi
        self._bfd_conj(self.q2, np.asfortranarray(self.z + self.beta))
        self.q = (self.tmp_q + self.alpha + self.q2) * self.diagQInv
        t4 = time.perf_counter()
        self.times[3] += t4 - t3
        # ---- multipliers :220-227 ----
        resi_alpha = self.tmp_q - self.q
        self._bfd(self.z2, self.q)
        resi_beta = self.z - self.z2
        self.alpha = self.alpha + self.tau * resi_alpha
        self.beta = self.beta + self.tau * resi_beta
        t5 = time.perf_counter()
        self.times[4] += t5 - t4
        # ---- KKT :229-336 ----
        brk = False
        adjustSigmaYes = IfAdjustSigma(it, self.lastSigmaIt)
        timed_out = (self.elapsed + (t5 - t_start)) > self.time_limit
        if self.checkSByS or adjustSigmaYes or it == self.maxit or timed_out:
            brk = self._kkt(self.tmp_q, resi_alpha, resi_beta, adjustSigmaYes, timed_out)
        t6 = time.perf_counter()
        self.times[5] += t6 - t5
        self.elapsed += t6 - t_start
        return brk

    def finish(self):
        """:339-369"""
        times = list(self.times)
        runHist, sigma = super().finish()
        self.var.name = 'Proximal ALM'
        self.var.time = dict(zip(TIME_NAMES, times + [self.elapsed, self.it]))
        return runHist, sigma


def solver_socp_PALM(var, opts, model):
    """[runHist, sigma] = solver_socp_PALM(var, opts, model)  (socp/dot2d/algorithms/solver_socp_PALM.m:1)"""
    st = PALMState(var, opts, model)
    st.run()
    return st.finish()
