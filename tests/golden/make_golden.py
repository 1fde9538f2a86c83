#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/).

The reference has no fixtures of its own and cannot be executed here (no MATLAB; its prebuilt MEX
binaries are never loaded), so these vectors are produced by this repository's restatement and pin
it against regressions; they are what the GPU path is compared with at the boundary
(tests/test_golden.py).  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import driver as OD, mexops                      # noqa: E402
from oracle.examples import (ensure_barrier_validity, gene_barrier_of_circle_pillar, get_example_1d,  # noqa: E402
                             get_example_2d, get_weight_by_barrier)
from oracle.inpalm import InPALMState                         # noqa: E402


def operators():
    rng = np.random.default_rng(20260104)
    out = {}
    x = rng.standard_normal((40, 10)) * rng.choice([0.1, 1.0, 20.0], size=(40, 1))
    x[0] = 0.0                       # 0/0 -> NaN row
    x[1] = 0.0; x[1, 0] = 2.5        # n = 0, x1 > 0 -> unchanged
    x[2] = 0.0; x[2, 0] = -2.5       # n = 0, x1 < 0 -> zero row
    x[3, 1:] = 0.0; x[3, 1] = abs(x[3, 0])            # on the cone boundary
    x = np.asfortranarray(x)
    p = np.empty_like(x, order="F")
    mexops.mexProjSoc(p, x)
    out["proj_in"], out["proj_out"] = x, p
    x6 = np.asfortranarray(rng.standard_normal((17, 6)))
    p6 = np.empty_like(x6, order="F")
    mexops.mexProjSoc(p6, x6)
    out["proj6_in"], out["proj6_out"] = x6, p6
    nt, nx, ny = 4, 6, 5
    Nz = ny * nx * (nt - 1)
    Nq = Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt
    q = rng.standard_normal(Nq)
    z = np.zeros((Nz, 10), order="F")
    mexops.mexBFd(z, q, nt, nx, ny, 0.731, 1.37)
    w = np.asfortranarray(rng.standard_normal((Nz, 10)))
    qa = np.zeros(Nq)
    mexops.mexBFdConj(qa, w, nt, nx, ny, 0.731)
    out.update(bfd_dims=np.array([nt, nx, ny]), bfd_q=q, bfd_z=z, bfdc_w=w, bfdc_q=qa,
               bfd_scale=np.array([0.731, 1.37]))
    nt1, nx1 = 5, 9
    Nz1 = nx1 * (nt1 - 1)
    q1 = rng.standard_normal(Nz1 + (nx1 - 1) * nt1)
    z1 = np.zeros((Nz1, 6), order="F")
    mexops.mexBFd1d(z1, q1, nt1, nx1, 1.21, 0.6)
    w1 = np.asfortranarray(rng.standard_normal((Nz1, 6)))
    qa1 = np.zeros_like(q1)
    mexops.mexBFdConj1d(qa1, w1, nt1, nx1, 1.21)
    out.update(bfd1_dims=np.array([nt1, nx1]), bfd1_q=q1, bfd1_z=z1, bfdc1_w=w1, bfdc1_q=qa1,
               bfd1_scale=np.array([1.21, 0.6]))
    np.savez_compressed(os.path.join(HERE, "operators.npz"), **out)


def trajectory(name, rho0, rho1, nt, K, weight=None, method="inPALM", **extra_opts):
    var, model, o = OD.make_level(rho0, rho1, nt, dict(tol=0.0, maxit=K, **extra_opts), method, weight)
    st = OD.make_state(var, o, model, method, weighted=weight is not None)
    st.run()
    hist, sigma = st.finish()
    extra = {} if weight is None else {"weight": weight}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rho0=rho0, rho1=rho1, nt=nt, K=K,
                        phi=var.phi, q=var.q, alpha=var.alpha, z=var.z, beta=var.beta, sigma=sigma,
                        kkt=hist["kkt"], iters=hist["iter"], pdGap=hist["pdGap"], cScale=var.cScale,
                        dScale=var.dScale, method=method, **{"opt_" + k: v for k, v in extra_opts.items()}, **extra)


if __name__ == "__main__":
    mexops.build()
    operators()
    r0, r1 = get_example_2d("example1", 17, 17)
    trajectory("traj_dot2d_17x17x9", r0, r1, 9, 40)
    r0, r1 = get_example_2d("example1", 16, 12)
    trajectory("traj_dot2d_alg2_16x12x8", r0, r1, 8, 25, method="ALG2")
    r0, r1 = get_example_1d("gaussian", 33)
    trajectory("traj_dot1d_33x17", r0, r1, 17, 40)
    r0, r1 = get_example_2d("example1", 16, 16)
    b = gene_barrier_of_circle_pillar()
    w = get_weight_by_barrier(16, 16, 8, b)
    r0, r1, _ = ensure_barrier_validity(r0, r1, b)
    trajectory("traj_wdot2d_16x16x8", r0, r1, 8, 30, weight=w)
    # loop variants
    r0, r1 = get_example_2d("example1", 16, 12)
    trajectory("traj_palm_16x12x8", r0, r1, 8, 25, method="PALM")
    trajectory("traj_accadmm_16x12x8", r0, r1, 8, 30, method="acc-ADMM", restart=7)
    trajectory("traj_accadmm_theta3_16x12x8", r0, r1, 8, 20, method="acc-ADMM", theta=3.0, restart=6)
    r0, r1 = get_example_2d("example1", 16, 16)
    r0, r1, _ = ensure_barrier_validity(r0, r1, b)
    trajectory("traj_waccadmm_16x16x8", r0, r1, 8, 25, weight=w, method="acc-ADMM")
    print("golden vectors written to", HERE)
