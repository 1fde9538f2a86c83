#!/usr/bin/env python3
"""Python twin of demo_wdot2d.m with the circle-pillar obstacle (demo_wdot2d.m:10-74): 129 x 129 x 129,
3 levels, tol 1e-3, weight 1e6 on the barrier edges."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dotsocp_amd as D  # noqa: E402

n = nt = 2 ** 7 + 1
rho0, rho1 = D.get_example_2d("example1", n, n)
barrier = D.gene_barrier_of_circle_pillar()
weight = D.get_weight_by_barrier(n, n, nt, barrier)
rho0, rho1, _ = D.ensure_barrier_validity(rho0, rho1, barrier)
opts = dict(tol=1e-3, weight=weight, maxit=10000)
output, timeML, runHistML, runHist = D.solver_wdotsocp2d(rho0, rho1, nt, 3, opts, "inPALM", barrier)
for lv, t in enumerate(timeML[:-1], 1):
    print(f"level {lv}: {int(t['Iters'])} iterations, {t['Total_Time']:.2f} s")
print("final KKT (1,3,6):", runHist["kkt"][-1][[0, 2, 5]])
print("mass conservation within 1e-2:", D.check_massConservation(output["rho"], 1e-2))
