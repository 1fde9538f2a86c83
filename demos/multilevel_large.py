#!/usr/bin/env python3
"""End-to-end multilevel solves at sizes beyond the reference demos (Gaussian -> Gaussian, Example 5.1, inPALM,
tol 1e-4): the whole level loop runs with the state resident on the GPU (dotsocp_jump_next_level between
levels, dotsocp_recover_outputs at the end).  usage: multilevel_large.py [n nt levelN] ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dotsocp_amd as D  # noqa: E402

cases = [(257, 65, 3), (513, 129, 4)]
if len(sys.argv) > 3:
    a = list(map(int, sys.argv[1:]))
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for n, nt, L in cases:
    rho0, rho1 = D.get_example_2d("example1", n, n)
    out = None                                    # the previous case's 1 GB output arrays are freed outside the timed call
    t = time.perf_counter()
    out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, nt, L, dict(tol=1e-4, maxit=3000), "inPALM")
    dt = time.perf_counter() - t
    its = [int(x["Iters"]) for x in timeML[:-1]]
    secs = [round(float(x["Total_Time"]), 2) for x in timeML[:-1]]
    print(f"{n}x{n}x{nt}, {L} levels: iterations {its}, loop seconds {secs}, wall {dt:.2f} s, "
          f"KKT(1,3,6,7) {hist['kkt'][-1][[0, 2, 5, 6]].max():.2e}, mass ok {D.check_massConservation(out['rho'], 1e-2)}",
          flush=True)
