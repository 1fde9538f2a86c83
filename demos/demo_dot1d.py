#!/usr/bin/env python3
"""Python twin of demo_dot1d.m: 1-D Gaussians, nx = 1025, nt = 33, 3 levels, tol 1e-5 (demo_dot1d.m:10-32)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dotsocp_amd as D  # noqa: E402

rho0, rho1 = D.get_example_1d("gaussian", 2 ** 10 + 1)
output, timeML, runHistML, runHist = D.solver_dotsocp1d(rho0, rho1, 2 ** 5 + 1, 3, dict(tol=1e-5, maxit=3000), "inPALM")
for lv, t in enumerate(timeML[:-1], 1):
    print(f"level {lv}: {int(t['Iters'])} iterations, {t['Total_Time']:.2f} s")
print("final KKT (1,3,6,7):", runHist["kkt"][-1][[0, 2, 5, 6]])
print("mass conservation within 1e-2:", D.check_massConservation(output["rho"], 1e-2))
