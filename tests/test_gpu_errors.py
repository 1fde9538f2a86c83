"""Error behaviour of the C ABI on a GPU box (include/dotsocp.h: status codes + dotsocp_last_error, no exceptions
across the boundary) and the 1-D MEX operators' argument checks (mexBFd1d.mexa64 raises mexBFd:invalidNumInputs /
invalidInput; the 2-D binaries validate nothing -- SURVEY.md 8b)."""
import ctypes

import numpy as np
import pytest

import dotsocp_amd as D
from dotsocp_amd import capi

pytestmark = pytest.mark.gpu


def _problem(ny=16, nx=16, nt=8, dim=2, weighted=0):
    p = capi.Problem()
    p.dim, p.weighted, p.ny, p.nx, p.nt = dim, weighted, ny, nx, nt
    p.D = p.E = p.cScale = p.dScale = 1.0
    p.normc, p.normd = 1.0, 1.0
    return p


def _opts(maxit=3):
    o = capi.Opts()
    o.tau, o.sigma, o.tol, o.maxit = 1.9, 1.0, 0.0, maxit
    o.ifCheckStepByStep, o.checkPrimDualFeas, o.scaling, o.time_limit = 0, -1, 1, 0.0
    return o


def _code(rc):
    with pytest.raises(capi.DotsocpError) as e:
        capi.check(rc)
    return e.value.code


def test_create_rejects_bad_problems():
    L = capi.lib()
    for bad in (_problem(dim=3), _problem(nt=1), _problem(nx=0)):
        assert not L.dotsocp_create(ctypes.byref(bad), 0, 1)
        assert L.dotsocp_last_error()
    assert not L.dotsocp_create(ctypes.byref(_problem()), 99, 1)          # device ordinal
    assert not L.dotsocp_create(ctypes.byref(_problem(nt=8)), 0, 5)        # more slabs than nt / 2


def test_call_sequence_is_enforced():
    L = capi.lib()
    p = _problem()
    ctx = L.dotsocp_create(ctypes.byref(p), 0, 1)
    assert ctx
    try:
        done = capi.i64()
        assert _code(L.dotsocp_run(ctx, 1, ctypes.byref(done))) == -4                 # run before begin
        res = capi.Result()
        assert _code(L.dotsocp_finish(ctx, ctypes.byref(res))) == -4                  # finish before begin
        assert _code(L.dotsocp_upload(ctx, 17, None)) == -1                           # unknown field / NULL
        assert _code(L.dotsocp_begin(ctx, None)) == -1
        o = _opts()
        o.sigma = 0.0
        assert _code(L.dotsocp_begin(ctx, ctypes.byref(o))) == -1                     # sigma must be positive
        o = _opts()
        assert _code(L.dotsocp_begin_method(ctx, ctypes.byref(o), 7, None)) == -1     # unknown method
        capi.check(L.dotsocp_begin(ctx, ctypes.byref(o)))
        assert _code(L.dotsocp_begin(ctx, ctypes.byref(o))) == -4                     # begin twice
        out = np.empty(8)
        assert _code(L.dotsocp_recover_outputs(ctx, None, None, None, None, None, capi.fptr(out), None, None)) == -4
        capi.check(L.dotsocp_run(ctx, -1, ctypes.byref(done)))
        assert done.value == 3
        capi.check(L.dotsocp_finish(ctx, ctypes.byref(res)))
        assert res.iters == 3 and res.hist_len >= 1
        assert _code(L.dotsocp_run(ctx, 1, ctypes.byref(done))) == -4                 # run after finish
        ms, n = capi.dbl(), capi.i64()
        assert _code(L.dotsocp_kernel_time(ctx, b"no_such_kernel", ctypes.byref(ms), ctypes.byref(n))) == -1
    finally:
        L.dotsocp_destroy(ctx)
    assert _code(L.dotsocp_run(None, 1, None)) == -1                                  # NULL context


@pytest.mark.parametrize("ny,nx,nt,nslabs", [(16, 12, 8, 1), (33, 17, 9, 1), (16, 12, 9, 3), (1, 24, 6, 1)])
def test_upload_layers_equals_the_full_upload(ny, nx, nt, nslabs):
    """dotsocp_upload_layers (driver extension: c's two non-zero layers instead of the whole vector): any layer range of
    phi / c lands where the full upload puts it -- pitched rows, time slabs, the 1-D engine layout (ny x 1) --, the other
    layers keep their contents; bad fields, ranges and call order are refused."""
    L = capi.lib()
    rng = np.random.default_rng(5)
    one_d = ny == 1
    p = _problem(ny=ny, nx=nx, nt=nt, dim=1 if one_d else 2)
    plane = nx if one_d else ny * nx
    full = rng.standard_normal(plane * nt)
    for field in (capi.F_C, capi.F_PHI):
        a = L.dotsocp_create(ctypes.byref(p), 0, nslabs)
        b = L.dotsocp_create(ctypes.byref(p), 0, nslabs)
        assert a and b
        try:
            capi.check(L.dotsocp_upload(a, field, capi.fptr(full)))
            want = full.copy()
            got0 = np.empty_like(full)
            capi.check(L.dotsocp_download(b, field, capi.fptr(got0)))
            assert not got0.any()                                                       # zeros after create
            expect = np.zeros_like(full)
            for t0, n in ((0, 1), (nt - 1, 1), (2, 3), (1, 0)):
                part = np.ascontiguousarray(full[plane * t0:plane * (t0 + n)]) if n else np.zeros(1)
                capi.check(L.dotsocp_upload_layers(b, field, capi.fptr(part), t0, n))
                expect[plane * t0:plane * (t0 + n)] = full[plane * t0:plane * (t0 + n)]
            got = np.empty_like(full)
            capi.check(L.dotsocp_download(b, field, capi.fptr(got)))
            assert np.array_equal(got, expect)
            capi.check(L.dotsocp_upload_layers(b, field, capi.fptr(full), 0, nt))       # the whole range
            capi.check(L.dotsocp_download(b, field, capi.fptr(got)))
            capi.check(L.dotsocp_download(a, field, capi.fptr(want)))
            assert np.array_equal(got, want) and np.array_equal(got, full)
            assert _code(L.dotsocp_upload_layers(b, capi.F_Q, capi.fptr(full), 0, 1)) == -1       # not a node field
            assert _code(L.dotsocp_upload_layers(b, field, capi.fptr(full), nt - 1, 2)) == -1     # past the end
            assert _code(L.dotsocp_upload_layers(b, field, capi.fptr(full), -1, 1)) == -1
            assert _code(L.dotsocp_upload_layers(b, field, None, 0, 1)) == -1
            o = _opts()
            capi.check(L.dotsocp_begin(b, ctypes.byref(o)))
            assert _code(L.dotsocp_upload_layers(b, field, capi.fptr(full), 0, 1)) == -4          # after begin
        finally:
            L.dotsocp_destroy(a)
            L.dotsocp_destroy(b)


def test_driver_start_with_lazy_fields_equals_the_eager_one():
    """initialize(lazy_zeros=True, phi=...) + InitialScaling + InPALMContext (c handed over as its two end layers, norms
    from those layers, z / beta / q / alpha left to the device default) runs the trajectory of the eager set-up."""
    from oracle import driver as OD
    from oracle.examples import get_example_2d
    rho0, rho1 = get_example_2d("example1", 24, 20)
    o = OD.default_opts(dict(tol=0.0, maxit=12), "inPALM")
    res = []
    for lazy in (False, True):
        var, model = D.initialize(rho0, rho1, 9, lazy_zeros=lazy)
        D.InitialScaling(var, model, True, None, dim=2)
        ctx = D.InPALMContext(var, o, model)
        assert ctx.run(-1) == 12
        hist, sigma = ctx.finish(download=False)
        res.append((ctx.download(capi.F_PHI, var.phi), ctx.download(capi.F_C, model.c), hist["kkt"].copy(), sigma,
                    var.cScale, model.normc, model.c.copy()))
        ctx.close()
    for x, y in zip(res[0], res[1]):     # ||c|| is summed over two layers instead of the whole vector: the last bit may differ
        np.testing.assert_allclose(np.asarray(y), np.asarray(x), rtol=1e-11, atol=1e-13 * np.abs(np.asarray(x)).max())


def test_device_buffers_of_destroyed_contexts_are_reused():
    """guard.hip's device block cache: the buffers of a destroyed context are handed out again (zero-filled) to the next
    context of that size -- same results, bit for bit, as with fresh memory --, dotsocp_release_cache() returns them to the
    driver, and a context of another size gets by without them."""
    import os
    if os.environ.get("DOTSOCP_CANARY", "0") != "0" or os.environ.get("DOTSOCP_DEVICE_CACHE", "1") == "0":
        pytest.skip("the cache is off under guard bands / DOTSOCP_DEVICE_CACHE=0")
    from oracle import driver as OD
    from oracle.examples import get_example_2d
    L = capi.lib()
    L.dotsocp_release_cache()
    o = OD.default_opts(dict(tol=0.0, maxit=25), "inPALM")

    def solve(n, nt):
        rho0, rho1 = get_example_2d("example1", n, n)
        var, model = D.initialize(rho0, rho1, nt)
        D.InitialScaling(var, model, True, None, dim=2)
        hist, sigma = D.solver_socp_inPALM(var, o, model)
        return var, hist, sigma

    a, ha, sa = solve(40, 12)                        # fresh memory; its buffers go into the cache on destroy
    b, hb, sb = solve(40, 12)                        # ... and serve this context
    held = L.dotsocp_release_cache()
    assert held > 8 * 40 * 40 * 12 * 27              # at least the state arrays came back
    assert L.dotsocp_release_cache() == 0
    c, hc, sc = solve(40, 12)                        # fresh again
    d, hd, sd = solve(24, 8)                         # another size next to cached blocks that do not fit
    assert sa == sb == sc
    for f in ("phi", "q", "z", "alpha", "beta"):
        assert np.array_equal(getattr(a, f), getattr(b, f)) and np.array_equal(getattr(a, f), getattr(c, f))
    assert np.array_equal(ha["kkt"], hb["kkt"]) and np.array_equal(ha["kkt"], hc["kkt"])
    assert np.all(np.isfinite(d.phi)) and hd["len"] == ha["len"]
    assert L.dotsocp_release_cache() > 0


def test_variant_restrictions():
    L = capi.lib()
    o = _opts()
    for prob, method in ((_problem(ny=1, nx=32, dim=1), capi.METHOD_ACCADMM), (_problem(ny=1, nx=32, dim=1), capi.METHOD_PALM),
                         (_problem(weighted=1), capi.METHOD_PALM)):
        ctx = L.dotsocp_create(ctypes.byref(prob), 0, 1)
        assert ctx
        try:
            assert _code(L.dotsocp_begin_method(ctx, ctypes.byref(o), method, None)) == -1
        finally:
            L.dotsocp_destroy(ctx)


def test_jump_next_level_checks_the_grids():
    L = capi.lib()
    coarse = L.dotsocp_create(ctypes.byref(_problem(9, 9, 5)), 0, 1)
    fine_ok = L.dotsocp_create(ctypes.byref(_problem(17, 17, 9)), 0, 1)
    fine_bad = L.dotsocp_create(ctypes.byref(_problem(16, 17, 9)), 0, 1)
    try:
        assert _code(L.dotsocp_jump_next_level(coarse, fine_ok)) == -4                # coarse not finished
        done, res = capi.i64(), capi.Result()
        capi.check(L.dotsocp_begin(coarse, ctypes.byref(_opts())))
        capi.check(L.dotsocp_run(coarse, -1, ctypes.byref(done)))
        capi.check(L.dotsocp_finish(coarse, ctypes.byref(res)))
        assert _code(L.dotsocp_jump_next_level(coarse, fine_bad)) == -1               # not 2 (n - 1) + 1
        capi.check(L.dotsocp_jump_next_level(coarse, fine_ok))
        assert _code(L.dotsocp_jump_next_level(None, fine_ok)) == -1
    finally:
        for c in (coarse, fine_ok, fine_bad):
            L.dotsocp_destroy(c)


def test_mex_1d_argument_errors():
    """mexBFd1d / mexBFdConj1d: too few inputs and a non-scalar scale are errors (mexBFd:invalidNumInputs /
    mexBFd:invalidInput in the reference binaries; a missing positional argument is a TypeError in Python)."""
    z = np.zeros((4 * 3, 6), order="F")
    q = np.zeros(4 * 3 + 3 * 4)
    with pytest.raises(TypeError):
        D.mexBFd1d(z, q, 4)
    with pytest.raises(ValueError, match="invalidInput"):
        D.mexBFd1d(z, q, 4, 4, np.ones(2))
    with pytest.raises(TypeError):
        D.mexBFdConj1d(q, z, 4)
    with pytest.raises(ValueError, match="invalidInput"):
        D.mexBFd1d(np.zeros((5, 6), order="F"), q, 4, 4)


@pytest.mark.parametrize("method", ["inPALM", "PALM", "acc-ADMM"])
def test_time_limit_stops_at_the_next_kkt_check(method):
    """opts.time_limit (solver_socp_inPALM.m:26-30,221,287-290): once exceeded, the current iteration ends with a KKT
    check and the loop breaks -- here in the very first iteration."""
    from oracle import driver as OD
    from oracle.examples import get_example_2d
    rho0, rho1 = get_example_2d("example1", 16, 16)
    var, model = D.initialize(rho0, rho1, 8)
    D.InitialScaling(var, model, True, None, dim=2)
    o = OD.default_opts(dict(tol=0.0, maxit=50, time_limit=1e-9), method)
    ctx = D.InPALMContext(var, o, model, method=method)
    assert ctx.run(-1) == 1
    hist, sigma = ctx.finish()
    ctx.close()
    assert hist["len"] == 1 and hist["iter"][0] == 1 and ctx.result.stopped == 1
    assert np.all(np.isfinite(var.phi)) and np.all(np.isfinite(hist["kkt"]))


def test_empty_and_degenerate_operator_inputs():
    """Zero rows are a no-op; grids too small for a cell are rejected with a message (the reference binaries would
    index out of bounds)."""
    L = capi.lib()
    x = np.zeros((0, 10), order="F")
    capi.check(L.dotsocp_proj_soc(None if x.size == 0 else capi.fptr(x), None, 0, 10))
    z, q = np.zeros((1, 10), order="F"), np.zeros(1)
    assert _code(L.dotsocp_bfd(capi.fptr(z), capi.fptr(q), 1, 1, 1, 1.0, 1.0)) == -1          # nt = 1: no cell
    one = np.array([[3.0, 0.0, 4.0, 0, 0, 0, 0, 0, 0, 0]], order="F")                           # a single row
    out = np.empty_like(one, order="F")
    D.mexProjSoc(out, one)
    c = (3.0 / 4.0 + 1.0) / 2.0
    np.testing.assert_allclose(out[0, :3], [c * 4.0, 0.0, c * 4.0], rtol=1e-15)
