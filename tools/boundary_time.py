#!/usr/bin/env python3
"""Wall time of the solver-level boundary B1 with HOST buffers, as a MATLAB / MEX caller pays it (include/dotsocp.h:
create + upload of phi, q, alpha, z, beta, c; K iterations; finish + download of the five iterates, alpha and beta
multiplied by sigma on the host): `python tools/boundary_time.py [ny nx nt [K]]`.  The PCIe-inclusive figures of DESIGN.md
section 7 come from here; bench.py's `value` has the state resident in HBM."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dotsocp_amd as D  # noqa: E402
from dotsocp_amd import capi  # noqa: E402

a = list(map(int, sys.argv[1:]))
ny, nx, nt = (a + [1024, 1024, 128])[:3] if len(a) >= 3 else (1024, 1024, 128)
K = a[3] if len(a) > 3 else 20
rho0, rho1 = D.get_example_2d("example1", ny, nx)
D.solver_dotsocp2d(*D.get_example_2d("example1", 33, 33), 9, 1, dict(tol=1e-2, maxit=20), "inPALM")   # runtime warm-up
var, model = D.initialize(rho0, rho1, nt)
D.InitialScaling(var, model, True, None, dim=2)
for x in (var.phi, var.q, var.alpha, var.z, var.beta, model.c):
    x += 0.0                                          # a caller's arrays exist in memory (numpy.zeros alone maps no page)
gb = sum(x.nbytes for x in (var.phi, var.q, var.alpha, var.z, var.beta)) / 1e9
opts = dict(tau=1.9, sigma=1.0, tol=0.0, maxit=K, scaling=True, ifCheckStepByStep=False, time_limit=1e9)
t0 = time.perf_counter()
ctx = D.InPALMContext(var, opts, model, z_unread=True)     # as solver_socp_inPALM / the MEX gateway do: z is overwritten before its first use
ctx.synchronize()
t1 = time.perf_counter()
assert ctx.run(-1) == K
ctx.synchronize()
t2 = time.perf_counter()
held = (var.phi, var.q, var.alpha, var.z, var.beta)   # finish() assigns new arrays: the caller's old ones are freed outside the timed region
hist, sigma = ctx.finish(download=True)
t3 = time.perf_counter()
ctx.close()
t4 = time.perf_counter()
del held
up = gb + model.c.nbytes / 1e9 - var.z.nbytes / 1e9
print(f"B1 with host buffers, {ny}x{nx}x{nt}, K = {K}: state {gb:.1f} GB (up: {up:.1f} GB, z stays home); create + upload {t1 - t0:.2f} s "
      f"({up / (t1 - t0):.1f} GB/s), loop {t2 - t1:.2f} s ({K / (t2 - t1):.1f} it/s), finish + download "
      f"{t3 - t2:.2f} s ({gb / (t3 - t2):.1f} GB/s), destroy {t4 - t3:.2f} s; whole call {t4 - t0:.2f} s = "
      f"{K / (t4 - t0):.1f} it/s including the transfers; host threads {os.environ.get('DOTSOCP_HOST_COPY_THREADS', 'default')}")
assert np.all(np.isfinite(hist["kkt"]))
