#!/usr/bin/env python3
"""Python twin of demo_dot2d.m: Gaussian-to-Gaussian (Example 5.1) on a 129 x 129 x 33 grid,
3 levels, inPALM, tol 1e-4 (demo_dot2d.m:10-18,55-63; Problem "example1" instead of the DOTmark
images, which need MATLAB's image toolbox)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dotsocp_amd as D  # noqa: E402

tol, nt, nx, levelN = 1e-4, 2 ** 5 + 1, 2 ** 7 + 1, 3
rho0, rho1 = D.get_example_2d("example1", nx, nx)
output, timeML, runHistML, runHist = D.solver_dotsocp2d(rho0, rho1, nt, levelN, dict(tol=tol, maxit=3000), "inPALM")
for lv, t in enumerate(timeML[:-1], 1):
    print(f"level {lv}: {int(t['Iters'])} iterations, {t['Total_Time']:.2f} s")
print("final KKT (1,3,6,7):", runHist["kkt"][-1][[0, 2, 5, 6]])
print("mass conservation within 1e-2:", D.check_massConservation(output["rho"], 1e-2))
