#!/usr/bin/env python3
"""SURVEY.md section 8c(iii) records a 2-D probe of the survey's own early proxy: example1 on 33 x 33 x 17, "fixed sigma = 1,
check every 10 -> stop at it = 80".  This script runs the restated loop (oracle/inpalm.py) in the four configurations that
description leaves open -- driver scaling (InitialScaling) on / off x in-loop rescale block on / off -- with sigma frozen
and a KKT check every 10 iterations, tol 1e-4, and prints where each stops (DESIGN.md section 5 keeps the table).
CPU only, a minute.  usage: python tools/survey_2d_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import driver as OD  # noqa: E402
from oracle import inpalm as OI  # noqa: E402
from oracle.examples import get_example_2d  # noqa: E402

OI.IfAdjustSigma = lambda it, last: (it - last) >= 10                      # a check every 10 iterations
OI.adjust_lagrangianParam = lambda sigma, xi, rule: (sigma, 1.0)           # sigma stays where it is

rho0, rho1 = get_example_2d("example1", 33, 33)
print("| driver scaling | rescale block | stops at iteration | final KKT (1, 3, 6, 7) |")
print("|---|---|---|---|")
for scaling in (True, False):
    for block in (True, False):
        var, model, o = OD.make_level(rho0, rho1, 17, dict(tol=1e-4, maxit=3000, scaling=scaling, sigma=1.0), "inPALM", None)
        st = OD.make_state(var, o, model, "inPALM")
        if not block:
            st.rescale = 0
        st.lastSigmaIt = 0
        st.run()
        hist, sigma = st.finish()
        k = hist["kkt"][-1]
        print(f"| {'on' if scaling else 'off'} | {'on' if block else 'off'} | {int(hist['iter'][-1])} | "
              f"{k[0]:.2e}, {k[2]:.2e}, {k[5]:.2e}, {k[6]:.2e} |")
# a fifth reading: no scaling and the reference's other initial sigma (0.1: solver_dotsocp2d.m:140-146)
var, model, o = OD.make_level(rho0, rho1, 17, dict(tol=1e-4, maxit=3000, scaling=False, sigma=0.1), "inPALM", None)
st = OD.make_state(var, o, model, "inPALM")
st.lastSigmaIt = 0
st.run()
hist, sigma = st.finish()
k = hist["kkt"][-1]
print(f"| off, sigma = 0.1 | off | {int(hist['iter'][-1])} | {k[0]:.2e}, {k[2]:.2e}, {k[5]:.2e}, {k[6]:.2e} |")
