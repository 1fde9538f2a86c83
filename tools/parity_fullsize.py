#!/usr/bin/env python3
"""One-off parity run at BASELINE's headline size: K iterations of inPALM on 1024 x 1024 x 128 on the GPU against
the CPU oracle (about half a minute per oracle iteration and ~100 GB of host memory, hence a tool and not a test).
usage: python tools/parity_fullsize.py [K] [n] [nt] [method = inPALM | ALG2 | PALM | acc-ADMM]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import dotsocp_amd as D  # noqa: E402
from oracle import driver as OD  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 128
method = sys.argv[4] if len(sys.argv) > 4 else "inPALM"
rho0, rho1 = D.get_example_2d("example1", n, n)
opts = dict(tol=0.0, maxit=K)
t = time.perf_counter()
ovar, omodel, oo = OD.make_level(rho0, rho1, nt, opts, method, None)
print(f"oracle set-up {time.perf_counter() - t:.1f} s", flush=True)
t = time.perf_counter()
st = OD.make_state(ovar, oo, omodel, method)
st.run()
o_hist, o_sigma = st.finish()
print(f"oracle {K} iterations {time.perf_counter() - t:.1f} s", flush=True)
gvar, gmodel = D.initialize(rho0, rho1, nt)
D.InitialScaling(gvar, gmodel, oo["scaling"], None, dim=2)
t = time.perf_counter()
solve = {"PALM": D.solver_socp_PALM, "acc-ADMM": D.solver_socp_accADMM}.get(method, D.solver_socp_inPALM)
g_hist, g_sigma = solve(gvar, oo, gmodel)
print(f"GPU incl. upload / download {time.perf_counter() - t:.1f} s", flush=True)
assert np.array_equal(g_hist["iter"], o_hist["iter"])
print("KKT max rel diff", float(np.max(np.abs(g_hist["kkt"] - o_hist["kkt"]) / (np.abs(o_hist["kkt"]) + 1e-10))))
print("sigma rel diff", abs(g_sigma - o_sigma) / abs(o_sigma))
for f in ("phi", "q", "z", "alpha", "beta"):
    a, b = getattr(gvar, f), getattr(ovar, f)
    print(f, "max rel err", float(np.max(np.abs(a - b)) / np.max(np.abs(b))), flush=True)
