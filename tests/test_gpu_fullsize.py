"""BASELINE.json's full single-GPU size (1024 x 1024 x 128) through size-independent properties -- the oracle
needs half a minute per iteration there, so parity at this size is asserted through what must hold at any size:
  * the cone projection is idempotent and lands in the cone (mexProjSoc semantics);
  * mexBFdConj is the exact adjoint of mexBFd: <BF q, w> = <q, F*B* w> (8a a2 / a3);
  * the loop is deterministic (two runs give identical iterates) and its time-slab decomposition reproduces the
    single-slab trajectory (8e: "1-GPU build must be bit-identical to the 1-slab case", slabs to rounding);
  * the state the loop carries stays consistent: z never leaves the cone, the KKT history is finite and the
    primal residuals fall.
The operator checks run on device-resident data through the *_dev entry points of the C ABI (torch only holds
the memory); the reference layout (column-major Nz x 10, q = [q0; bx; by]) is the same as at small sizes, where
the same kernels are compared with the oracle bit for bit (tests/test_gpu_operators.py)."""
import ctypes

import numpy as np
import pytest

import dotsocp_amd as D
from dotsocp_amd import capi

pytestmark = pytest.mark.gpu
NY, NX, NT = 1024, 1024, 128


def _sizes():
    Nz = NY * NX * (NT - 1)
    return Nz, Nz + NY * (NX - 1) * NT + (NY - 1) * NX * NT


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def test_projection_idempotent_and_in_cone_at_full_size():
    import torch
    Nz, _ = _sizes()
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((10, Nz), generator=g, device="cuda", dtype=torch.float64)       # column-major Nz x 10
    x[0] *= 3.0
    p1, p2 = torch.empty_like(x), torch.empty_like(x)
    L = capi.lib()
    capi.check(L.dotsocp_proj_soc_dev(_ptr(p1), _ptr(x), Nz, 10, None))
    capi.check(L.dotsocp_proj_soc_dev(_ptr(p2), _ptr(p1), Nz, 10, None))
    torch.cuda.synchronize()
    nrm = torch.linalg.vector_norm(p1[1:], dim=0)
    assert bool(torch.all(p1[0] >= nrm * (1 - 1e-14)))                               # in the cone
    apex = (p1[0] == 0) & (nrm == 0)                                                 # projected onto the apex: 0/0 -> NaN like the reference
    keep = ~apex
    assert int(apex.sum()) < Nz // 2
    assert bool(torch.all(torch.abs(p2[:, keep] - p1[:, keep]) <= 1e-15 * (1 + torch.abs(p1[:, keep]))))
    inside = x[0] >= torch.linalg.vector_norm(x[1:], dim=0)
    assert bool(torch.equal(p1[:, inside], x[:, inside]))                            # points of the cone are fixed
    del x, p1, p2
    torch.cuda.empty_cache()


def test_bfd_adjoint_identity_at_full_size():
    import torch
    Nz, Nq = _sizes()
    g = torch.Generator(device="cuda").manual_seed(12)
    q = torch.randn(Nq, generator=g, device="cuda", dtype=torch.float64)
    w = torch.randn((10, Nz), generator=g, device="cuda", dtype=torch.float64)
    z = torch.zeros((10, Nz), device="cuda", dtype=torch.float64)
    qa = torch.zeros(Nq, device="cuda", dtype=torch.float64)
    L = capi.lib()
    s = 0.83
    capi.check(L.dotsocp_bfd_dev(_ptr(z), _ptr(q), NT, NX, NY, s, 0.0, None))        # dF = 0: the linear part BF q
    capi.check(L.dotsocp_bfd_conj_dev(_ptr(qa), _ptr(w), NT, NX, NY, s, None))
    torch.cuda.synchronize()
    lhs, rhs = float(torch.sum(z * w)), float(torch.dot(q, qa))
    scale = float(torch.linalg.vector_norm(z)) * float(torch.linalg.vector_norm(w))
    assert abs(lhs - rhs) <= 1e-12 * scale
    del q, w, z, qa
    torch.cuda.empty_cache()


def _run(K, nslabs=1, keep=("phi", "q")):
    rho0, rho1 = D.get_example_2d("example1", NY, NX)
    var, model = D.initialize(rho0, rho1, NT, lazy_zeros=True)
    D.InitialScaling(var, model, True, None, dim=2)
    o = dict(tau=1.9, sigma=1.0, tol=0.0, maxit=K, scaling=True, ifCheckStepByStep=False, time_limit=1e9)
    ctx = D.InPALMContext(var, o, model, nslabs=nslabs)
    ctx.run(-1)
    hist, sigma = ctx.finish(download=False)
    Nz, Nq = _sizes()
    out = {}
    if "phi" in keep:
        out["phi"] = ctx.download(capi.F_PHI, np.empty(NY * NX * NT))
    if "q" in keep:
        out["q"] = ctx.download(capi.F_Q, np.empty(Nq))
    if "z" in keep:
        out["z"] = ctx.download(capi.F_Z, np.empty((Nz, 10), order="F"))
    ctx.close()
    return out, hist, sigma


def test_loop_deterministic_and_slab_invariant_at_full_size():
    K = 12                                              # several KKT checks and sigma updates on the way
    a, ha, sa = _run(K, keep=("phi", "q", "z"))
    b, hb, sb = _run(K)
    assert sa == sb and np.array_equal(ha["kkt"], hb["kkt"])
    assert np.array_equal(a["phi"], b["phi"]) and np.array_equal(a["q"], b["q"])     # run-to-run identical
    z = a.pop("z")
    nrm = np.sqrt(np.einsum("ij,ij->i", z[:, 1:], z[:, 1:]))
    assert np.all(z[:, 0] >= nrm * (1 - 1e-13))                                       # z = Pi_Q(...) stays in the cone
    del z, nrm
    assert np.all(np.isfinite(ha["kkt"])) and ha["kkt"].shape[1] == 7 and ha["iter"][-1] == K
    assert ha["kkt"][-1, 0] < ha["kkt"][0, 0] and ha["kkt"][-1, 1] < ha["kkt"][0, 1]  # primal residuals fall
    c, hc, sc = _run(K, nslabs=4)
    assert abs(sc - sa) <= 1e-12 * abs(sa)
    np.testing.assert_array_equal(hc["iter"], ha["iter"])
    np.testing.assert_allclose(hc["kkt"], ha["kkt"], rtol=1e-7, atol=1e-10)
    for f in ("phi", "q"):
        err = np.max(np.abs(c[f] - a[f])) / np.max(np.abs(a[f]))
        assert err <= 1e-9, (f, err)
