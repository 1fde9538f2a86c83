"""Guard bands around the device buffers (DOTSOCP_CANARY=1, csrc/guard.hip; SURVEY.md section 5 "out-of-bounds canaries").
The tile kernels run clamped lanes on every grid that is not a multiple of 64 x 4 -- exactly where a silent overrun
would hide -- so the odd-shape and unequal-slab cases of the suite are repeated here between NaN-pattern guard words:
finish() verifies every band (an out-of-bounds write fails the solve with a message naming the buffer), an
out-of-bounds read would pull NaNs into the iterates, and the results must be bit-identical to the unguarded run."""
import numpy as np
import pytest

import dotsocp_amd as D
from dotsocp_amd import capi
from oracle import driver as OD
from oracle.examples import (ensure_barrier_validity, gene_barrier_of_circle_pillar, get_example_1d,
                             get_example_2d, get_weight_by_barrier)

pytestmark = pytest.mark.gpu
FIELDS = ("phi", "q", "z", "alpha", "beta")


def _solve(rho0, rho1, nt, K, method="inPALM", weight=None, **kw):
    dim = 2 if np.ndim(rho0) == 2 else 1
    var, model = D.initialize(rho0, rho1, nt)
    if weight is not None:
        model.weight = np.asarray(weight, dtype=np.float64)
    o = OD.default_opts(dict(tol=0.0, maxit=K), method, weight is not None)
    D.InitialScaling(var, model, o["scaling"], None, dim=dim, weighted=weight is not None)
    ctx = D.InPALMContext(var, o, model, weighted=weight is not None, method=method, **kw)
    ctx.run(-1)
    hist, sigma = ctx.finish(download=True)        # raises DotsocpError when a guard band was overwritten
    outs = ctx.outputs()
    ctx.close()
    assert capi.lib().dotsocp_canary_check() == 0, capi.lib().dotsocp_last_error().decode()
    return var, hist, sigma, outs


def _guarded_equals_plain(monkeypatch, *args, **kw):
    monkeypatch.delenv("DOTSOCP_CANARY", raising=False)
    ref, h0, s0, o0 = _solve(*args, **kw)
    monkeypatch.setenv("DOTSOCP_CANARY", "1")
    got, h1, s1, o1 = _solve(*args, **kw)
    assert s0 == s1 and np.array_equal(h0["kkt"], h1["kkt"])
    for f in FIELDS:
        a = getattr(got, f)
        assert np.all(np.isfinite(a)), f
        assert np.array_equal(a, getattr(ref, f)), f
    for k in o0:
        assert np.array_equal(o0[k], o1[k]), k


@pytest.mark.parametrize("ny,nx,nt,K", [(100, 70, 20, 15), (50, 130, 9, 12), (65, 129, 33, 10), (2, 2, 2, 3), (3, 2, 2, 4),
                                         (63, 5, 7, 8), (129, 3, 5, 8)])
def test_odd_shapes_between_guard_bands(ny, nx, nt, K, monkeypatch):
    if ny * nx <= 6:
        rho0 = np.ones((ny, nx))
        rho1 = np.ones((ny, nx))
        rho1.flat[0] = 1.5
        rho1 /= rho1.mean()
    else:
        rho0, rho1 = get_example_2d("example1", ny, nx)
    _guarded_equals_plain(monkeypatch, rho0, rho1, nt, K)


@pytest.mark.parametrize("method", ["inPALM", "PALM", "acc-ADMM"])
@pytest.mark.parametrize("ny,nx,nt,nslabs", [(70, 50, 37, 5), (65, 130, 53, 6), (129, 31, 26, 2)])
def test_unequal_slabs_between_guard_bands(ny, nx, nt, nslabs, method, monkeypatch):
    rho0, rho1 = get_example_2d("example1", ny, nx)
    _guarded_equals_plain(monkeypatch, rho0, rho1, nt, 20, method=method, ngpu=nslabs)


def test_1d_weighted_and_dct_transposes_between_guard_bands(monkeypatch):
    r0, r1 = get_example_1d("gaussian", 129)
    _guarded_equals_plain(monkeypatch, r0, r1, 33, 25)
    _guarded_equals_plain(monkeypatch, r0, r1, 33, 25, ngpu=3)
    rho0, rho1 = get_example_2d("example1", 33, 47)
    barrier = gene_barrier_of_circle_pillar()
    weight = get_weight_by_barrier(47, 33, 13, barrier)
    rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    _guarded_equals_plain(monkeypatch, rho0, rho1, 13, 15, weight=weight)
    monkeypatch.setenv("DOTSOCP_TSOLVE", "dct")
    _guarded_equals_plain(monkeypatch, rho0, rho1, 13, 15, weight=weight, ngpu=3)


def test_an_overrun_is_caught(monkeypatch):
    """The library's test hook writes ONE double right behind model.c (as a kernel overrunning its last tile would);
    finish() must fail and name the damage."""
    monkeypatch.setenv("DOTSOCP_CANARY", "1")
    monkeypatch.setenv("DOTSOCP_CANARY_SELFTEST", "1")
    rho0, rho1 = get_example_2d("example1", 20, 12)
    var, model = D.initialize(rho0, rho1, 6)
    o = OD.default_opts(dict(tol=0.0, maxit=3), "inPALM", False)
    D.InitialScaling(var, model, True, None, dim=2)
    ctx = D.InPALMContext(var, o, model)
    try:
        ctx.run(-1)
        with pytest.raises(capi.DotsocpError) as ei:
            ctx.finish(download=False)
        assert "canary" in str(ei.value) and "behind the payload at word 0" in str(ei.value)
        assert capi.lib().dotsocp_canary_check() == 1          # the standalone check sees the same damage
    finally:
        ctx.close()
    monkeypatch.delenv("DOTSOCP_CANARY_SELFTEST")
    assert capi.lib().dotsocp_canary_check() == 0              # the damaged context is gone
