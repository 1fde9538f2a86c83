"""The accelerated (Halpern / Nesterov-type) ADMM loop of the reference, restated with numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference holds no fixture for
this loop and MATLAB is not available, so this restatement is checked through the same invariants as
the inPALM oracle plus convergence to the same transport as inPALM.  Follows, statement by statement,
  socp/dot2d/algorithms/solver_socp_accADMM.m    (2-D)
  socp/wdot2d/algorithms/solver_wsocp_accADMM.m  (weighted 2-D)
The KKT block of both files is the one of the inPALM files evaluated at (phi, z, q, alpha, beta) after
the z-step (solver_socp_accADMM.m:251-366 == solver_socp_inPALM.m:218-323 up to the restart at :353-357),
so it is inherited from InPALMState.
"""
import time

import numpy as np

from . import mexops
from .inpalm import InPALMState, _get
from .model import IfAdjustSigma, oper_poisson

TIME_NAMES = ['Step_1_Q_Step', 'Step_2_Multiplier', 'Step_3_1_FFT', 'Step_3_2_ProjSOC', 'KKT', 'Interp',
              'Total_Time', 'Iters']
WTIME_NAMES = ['Step_1_Q_Step', 'Step_2_Multiplier', 'Step_3_1_FFT', 'Step_3_2_ProjSOC', 'Step_4_Interp', 'KKT',
               'Total_Time', 'Iters']      # solver_wsocp_accADMM.m:443-444
FIELDS = ("phi", "z", "q", "alpha", "beta")


class AccADMMState(InPALMState):
    def __init__(self, var, opts, model, weighted=False):
        if not hasattr(model, "ny"):
            raise ValueError("the reference has no 1-D acc-ADMM (socp/dot1d/algorithms holds inPALM only)")
        opts = dict(opts) if isinstance(opts, dict) else opts
        if isinstance(opts, dict):
            opts.setdefault("tau", 1.0)          # unused by this loop
        super().__init__(var, opts, model, weighted)
        # :12-34
        self.restart = int(_get(opts, "restart", 100))
        self.stepRho = float(_get(opts, "rho", 2))
        self.stepAlpha = float(_get(opts, "theta", 2))
        self.HalpernYes = (self.stepAlpha == 2)
        self.times = np.zeros(6)
        # :157-163
        self.k = 0
        self.old = self._copy()
        if self.HalpernYes:
            self.anchor = self._copy()
        self.hat_old = None

    def _copy(self):
        return {f: np.array(getattr(self, f), copy=True, order="F") for f in FIELDS}

    def step(self):
        """solver_socp_accADMM.m:166-424, one pass of the `for it` body."""
        self.it += 1
        it = self.it
        w = self.weight
        t_start = time.perf_counter()
        # ---- rescaling :167-225 ----
        scaleYes = 0
        if self.rescale >= 3 and it % 200 == 0:
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
            if not self.weighted:
                self.norm_d = self.norm_d / dScale2
            self.alpha = self.alpha * dScale2 / cScale2 ** 2
            self.beta = self.beta * dScale2 / cScale2 ** 2
            self.phi = self.phi / dScale2
            self.q = self.q / dScale2
            self.z = self.z / dScale2
            self.dScale = dScale2 * self.dScale
            self.cScale = cScale2 * self.cScale
            self.scaleD = self.E / self.dScale
            self.sigmaScale = self.sigmaScale * (cScale2 / dScale2)
            self.k = 0                                            # :217-222 restart
            self.old = self._copy()
            if self.HalpernYes:
                self.anchor = self._copy()
            self.rescale += 1
        # ---- step q :227-232 ----
        t0 = time.perf_counter()
        self._bfd_conj(self.q2, np.asfortranarray(self.z + self.beta))
        tmp_q = self.A @ self.phi
        if w is None:
            self.q = (tmp_q + self.alpha + self.q2) * self.diagQInv
        else:
            self.q = (w * (tmp_q + self.alpha) + self.q2) * self.diagQInv
        t1 = time.perf_counter()
        self.times[0] += t1 - t0
        # ---- multipliers :234-239 ----
        self._bfd(self.z2, self.q)
        self.alpha = (self.alpha + tmp_q - self.q) if w is None else (self.alpha + tmp_q - w * self.q)
        self.beta = self.beta + self.z - self.z2
        t2 = time.perf_counter()
        self.times[1] += t2 - t1
        # ---- step phi :241-244 ----
        u = (self.q - self.alpha) if w is None else (w * self.q - self.alpha)
        rhs = self.AT @ u + self.c
        self.phi = oper_poisson(self.kernel, rhs.reshape(self.dims, order="F")).ravel(order="F")
        t3 = time.perf_counter()
        self.times[2] += t3 - t2
        # ---- step z :246-249 ----
        mexops.mexProjSoc(self.z, np.asfortranarray(self.z2 - self.beta))
        t4 = time.perf_counter()
        self.times[3] += t4 - t3
        # ---- KKT :251-367 ----
        brk = False
        adjustSigmaYes = IfAdjustSigma(it, self.lastSigmaIt)
        timed_out = (self.elapsed + (t4 - t_start)) > self.time_limit
        if self.checkSByS or adjustSigmaYes or it == self.maxit or timed_out:
            tmp_q = self.A @ self.phi                                         # :259
            resi_alpha = (tmp_q - self.q) if w is None else (tmp_q - w * self.q)   # :273
            resi_beta = self.z - self.z2                                      # :271,274 (z2 = BF q + d)
            brk = self._kkt(tmp_q, resi_alpha, resi_beta, adjustSigmaYes, timed_out)
        t5 = time.perf_counter()
        self.times[4] += t5 - t4
        if brk:
            self.elapsed += t5 - t_start
            return True
        # ---- interpolation :369-423 ----
        self._interpolate()
        t6 = time.perf_counter()
        self.times[5] += t6 - t5
        self.elapsed += t6 - t_start
        return False

    def _apply_sigma_factor(self, factor):
        """:346-358"""
        self.alpha = self.alpha / factor
        self.old["alpha"] = self.old["alpha"] / factor
        self.beta = self.beta / factor
        self.old["beta"] = self.old["beta"] / factor
        self.c = self.c / factor
        self.k = 0
        if self.HalpernYes:
            self.anchor = self._copy()

    def _interpolate(self):
        rho, k = self.stepRho, self.k
        if self.HalpernYes:                                                    # :371-388
            c1 = 1 / (k + 2)
            c2 = (k + 1) / (k + 2)
            for f in FIELDS:
                x = getattr(self, f)
                setattr(self, f, c1 * self.anchor[f] + c2 * ((1 - rho) * self.old[f] + rho * x))
            self.k += 1
            self.old = self._copy()
            if self.k >= self.restart:
                self.k = 0
                self.anchor = self._copy()
        else:                                                                  # :389-422
            hat = {f: (1 - rho) * self.old[f] + rho * getattr(self, f) for f in FIELDS}
            c1 = self.stepAlpha / (2 * (k + self.stepAlpha))
            if k == 0:
                for f in FIELDS:
                    setattr(self, f, (1 - c1) * self.old[f] + c1 * hat[f])
            else:
                c2 = k / (k + self.stepAlpha)
                for f in FIELDS:
                    setattr(self, f, (1 - c1) * self.old[f] + (c1 + c2) * hat[f] - c2 * self.hat_old[f])
            self.k += 1
            self.old = self._copy()
            if self.k >= self.restart:
                self.k = 0
            else:
                self.hat_old = hat
        self.z = np.asfortranarray(self.z)
        self.beta = np.asfortranarray(self.beta)

    def finish(self):
        """:427-456"""
        runHist, sigma = super().finish()
        self.var.name = 'Accelerated ADMM'
        t = list(self.times)
        if self.weighted:
            self.var.time = dict(zip(WTIME_NAMES, t[:4] + [t[5], t[4], self.elapsed, self.it]))
        else:
            self.var.time = dict(zip(TIME_NAMES, t + [self.elapsed, self.it]))
        return runHist, sigma


def solver_socp_accADMM(var, opts, model):
    """[runHist, sigma] = solver_socp_accADMM(var, opts, model)  (socp/dot2d/algorithms/solver_socp_accADMM.m:1)"""
    st = AccADMMState(var, opts, model, weighted=False)
    st.run()
    return st.finish()


def solver_wsocp_accADMM(var, opts, model):
    """socp/wdot2d/algorithms/solver_wsocp_accADMM.m:1"""
    st = AccADMMState(var, opts, model, weighted=True)
    st.run()
    return st.finish()
