"""The MEX gateways of dot-socp_amd/mex/ (boundary B1 / B2 as MATLAB would see it) EXECUTED on a box without
MATLAB: they are compiled against a small stand-in for libmx / libmex written for this purpose
(tests/fake_mx/fake_mx.c, test infrastructure: real double matrices, 1 x 1 structs, strings,
mexErrMsgIdAndTxt as a non-local exit) and their mexFunction is called through ctypes with the arguments the
reference's MATLAB code passes (solver_socp_inPALM.m:133,187,199,205; solver_dotsocp2d.m:208).  Checked: in-place
semantics of the first argument, scalar truncation, the error identifiers of the 1-D binaries, and the solver
gateway's output struct against the Python binding of the same C ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD, mexops as O
from oracle.examples import get_example_2d

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEX = os.path.join(ROOT, "dot-socp_amd", "mex")
OUT = os.path.join(ROOT, "tests", "fake_mx", "_build")
vp = ctypes.c_void_p


@pytest.fixture(scope="module")
def mx():
    os.makedirs(OUT, exist_ok=True)
    fake = os.path.join(OUT, "libfake_mx.so")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-shared", "-fPIC", "-o", fake,
                           os.path.join(ROOT, "tests", "fake_mx", "fake_mx.c")])
    L = ctypes.CDLL(fake, mode=ctypes.RTLD_GLOBAL)
    for name, res, args in [("fmx_wrap_double", vp, [ctypes.c_size_t, ctypes.c_size_t, vp]),
                            ("fmx_string", vp, [ctypes.c_char_p]),
                            ("fmx_struct", vp, [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]),
                            ("fmx_free", None, [vp]), ("fmx_clear_mex", None, []),
                            ("fmx_call", ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(vp)]),
                            ("fmx_error_id", ctypes.c_char_p, []), ("fmx_error_msg", ctypes.c_char_p, []),
                            ("mxCreateDoubleScalar", vp, [ctypes.c_double]),
                            ("mxSetField", None, [vp, ctypes.c_size_t, ctypes.c_char_p, vp]),
                            ("mxGetField", vp, [vp, ctypes.c_size_t, ctypes.c_char_p]),
                            ("mxGetPr", ctypes.POINTER(ctypes.c_double), [vp]),
                            ("mxGetM", ctypes.c_size_t, [vp]), ("mxGetN", ctypes.c_size_t, [vp])]:
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    gates = {}
    libdir = os.path.join(ROOT, "dot-socp_amd", "lib")
    for src in ("mexProjSoc", "mexBFd", "mexBFdConj", "mexBFd1d", "mexBFdConj1d", "dotsocp_inpalm_mex"):
        so = os.path.join(OUT, src + ".so")
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-shared", "-fPIC", "-I" + os.path.join(MEX, "compile_check"),
                               "-I" + os.path.join(ROOT, "include"), "-I" + MEX, os.path.join(MEX, src + ".c"), "-o", so,
                               "-L" + libdir, "-ldotsocp", "-L" + OUT, "-lfake_mx",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath," + OUT])
        gates[src] = ctypes.CDLL(so).mexFunction
    D.capi.lib()                      # same libdotsocp instance as the Python binding
    return L, gates


class Mx:
    """An mxArray returned by a gateway, passed on to another call as it is (e.g. a level handle)."""

    def __init__(self, h):
        self.h = h


class Call:
    """Builds prhs from numpy arrays (aliased, so in-place writes are visible), scalars and dicts, runs the gateway."""

    def __init__(self, L):
        self.L, self.keep, self.made = L, [], []

    def arg(self, v):
        L = self.L
        if isinstance(v, Mx):
            return v.h
        if isinstance(v, np.ndarray):
            a = v if v.flags.f_contiguous else np.asfortranarray(v)
            assert a is v, "pass Fortran-ordered arrays so that in-place results are visible"
            m, n = (a.shape[0], a.shape[1]) if a.ndim == 2 else (a.size, 1)
            h = L.fmx_wrap_double(m, n, a.ctypes.data)
        elif isinstance(v, str):
            h = L.fmx_string(v.encode())
        elif isinstance(v, dict):
            names = (ctypes.c_char_p * len(v))(*[k.encode() for k in v])
            h = L.fmx_struct(len(v), names)
            for k, x in v.items():
                L.mxSetField(h, 0, k.encode(), self.arg(x))      # the struct owns its fields
            self.made.append(h)
            return h
        else:
            buf = np.array([float(v)])
            self.keep.append(buf)
            h = L.fmx_wrap_double(1, 1, buf.ctypes.data)
        self.keep.append(v)
        return h

    def run(self, fn, args, nlhs=0):
        hs = []
        for v in args:
            h = self.arg(v)
            hs.append(h)
            if not isinstance(v, (dict, Mx)):
                self.made.append(h)
        prhs = (vp * max(len(hs), 1))(*hs)
        plhs = (vp * max(nlhs, 1))()
        rc = self.L.fmx_call(ctypes.cast(fn, vp), nlhs, plhs, len(hs), prhs)
        err = (self.L.fmx_error_id().decode(), self.L.fmx_error_msg().decode()) if rc else None
        return err, [plhs[i] for i in range(nlhs)]


def _field(L, s, name):
    h = L.mxGetField(s, 0, name.encode())
    assert h, name
    m, n = L.mxGetM(h), L.mxGetN(h)
    return np.ctypeslib.as_array(L.mxGetPr(h), shape=(m * n,)).reshape((m, n), order="F").copy()


def test_operator_gateways_in_place(mx):
    L, g = mx
    rng = np.random.default_rng(5)
    nt, nx, ny = 4, 6, 5
    Nz = ny * nx * (nt - 1)
    Nq = Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt
    x = np.asfortranarray(rng.standard_normal((Nz, 10)))
    out = np.full((Nz, 10), 7.0, order="F")
    ref = np.empty_like(out, order="F")
    O.mexProjSoc(ref, x)
    err, _ = Call(L).run(g["mexProjSoc"], [out, x])
    assert err is None and np.array_equal(out, ref)
    q = rng.standard_normal(Nq)
    z, zr = np.zeros((Nz, 10), order="F"), np.zeros((Nz, 10), order="F")
    O.mexBFd(zr, q, nt, nx, ny, 0.7, 1.3)
    err, _ = Call(L).run(g["mexBFd"], [z, q, nt + 0.9, nx, ny, 0.7, 1.3])       # doubles are truncated (cvttsd2si)
    assert err is None and np.array_equal(z, zr)
    qa, qr = np.zeros(Nq), np.zeros(Nq)
    O.mexBFdConj(qr, x, nt, nx, ny, 0.7)
    err, _ = Call(L).run(g["mexBFdConj"], [qa, x, nt, nx, ny, 0.7])
    assert err is None and np.array_equal(qa, qr)
    z5, z5r = np.zeros((Nz, 10), order="F"), np.zeros((Nz, 10), order="F")       # defaults scale = dF = 1
    O.mexBFd(z5r, q, nt, nx, ny, 1.0, 1.0)
    err, _ = Call(L).run(g["mexBFd"], [z5, q, nt, nx, ny])
    assert err is None and np.array_equal(z5, z5r)


def test_1d_gateways_and_their_error_identifiers(mx):
    L, g = mx
    rng = np.random.default_rng(6)
    nt, nx = 5, 9
    Nz, Nq = nx * (nt - 1), nx * (nt - 1) + (nx - 1) * nt
    q = rng.standard_normal(Nq)
    z, zr = np.zeros((Nz, 6), order="F"), np.zeros((Nz, 6), order="F")
    O.mexBFd1d(zr, q, nt, nx, 1.2, 0.6)
    err, _ = Call(L).run(g["mexBFd1d"], [z, q, nt, nx, 1.2, 0.6])
    assert err is None and np.array_equal(z, zr)
    w = np.asfortranarray(rng.standard_normal((Nz, 6)))
    qa, qr = np.zeros(Nq), np.zeros(Nq)
    O.mexBFdConj1d(qr, w, nt, nx, 1.2)
    err, _ = Call(L).run(g["mexBFdConj1d"], [qa, w, nt, nx, 1.2])
    assert err is None and np.array_equal(qa, qr)
    # the reference binaries' identifiers
    err, _ = Call(L).run(g["mexBFd1d"], [z, q, nt])
    assert err and err[0] == "mexBFd:invalidNumInputs"
    err, _ = Call(L).run(g["mexBFd1d"], [z, q, nt, nx], nlhs=1)
    assert err and err[0] == "mexBFd:invalidNumOutputs"
    err, _ = Call(L).run(g["mexBFd1d"], [z, q, nt, nx, np.ones(2)])
    assert err and err[0] == "mexBFd:invalidInput"
    err, _ = Call(L).run(g["mexBFd1d"], [np.zeros((Nz + 1, 6), order="F"), q, nt, nx])
    assert err and err[0] == "mexBFd:invalidInput"


@pytest.mark.parametrize("method,weighted", [("inPALM", False), ("accADMM", False), ("PALM", False), ("inPALM", True),
                                             ("accADMM", True)])
def test_solver_gateway_matches_the_python_binding(mx, method, weighted):
    """dotsocp_inpalm_mex(S, opts) as solver_socp_inPALM.m / solver_wsocp_inPALM.m / solver_socp_accADMM.m /
    solver_wsocp_accADMM.m / solver_socp_PALM.m call it (dotsocp_run_inpalm.m)."""
    from oracle.examples import ensure_barrier_validity, gene_barrier_of_circle_pillar, get_weight_by_barrier
    L, g = mx
    rho0, rho1 = get_example_2d("example1", 24, 16)
    nt, K = 8, 14
    pym = {"inPALM": "inPALM", "accADMM": "acc-ADMM", "PALM": "PALM"}[method]
    o = OD.default_opts(dict(tol=0.0, maxit=K), pym, weighted)
    weight = None
    if weighted:
        barrier = gene_barrier_of_circle_pillar()
        weight = get_weight_by_barrier(16, 24, nt, barrier)
        rho0, rho1, _ = ensure_barrier_validity(rho0, rho1, barrier)
    var, model = D.initialize(rho0, rho1, nt)
    if weighted:
        model.weight = weight
    D.InitialScaling(var, model, True, None, dim=2, weighted=weighted)
    S = dict(phi=var.phi.copy(), q=var.q.copy(), alpha=var.alpha.copy(), z=var.z.copy(order="F"), beta=var.beta.copy(order="F"),
             c=model.c.copy(), nx=model.nx, ny=model.ny, nt=model.nt, D=var.D, E=var.E, cScale=var.cScale,
             dScale=var.dScale, normc=model.normc, normd=model.normd)
    if weighted:
        S["weight"] = weight.copy()
    opts = dict(sigma=o["sigma"], maxit=o["maxit"], tol=o["tol"], ifCheckStepByStep=0.0, scaling=1.0, method=method)
    if method != "accADMM":
        opts["tau"] = o["tau"]
    call = Call(L)
    err, outs = call.run(g["dotsocp_inpalm_mex"], [S, opts], nlhs=1)
    assert err is None, err
    out = outs[0]
    solve = {("inPALM", False): D.solver_socp_inPALM, ("accADMM", False): D.solver_socp_accADMM,
             ("PALM", False): D.solver_socp_PALM, ("inPALM", True): D.solver_wsocp_inPALM,
             ("accADMM", True): D.solver_wsocp_accADMM}[(method, weighted)]
    hist, sigma = solve(var, o, model)
    for f in ("phi", "q", "alpha"):
        np.testing.assert_array_equal(_field(L, out, f).ravel(order="F"), getattr(var, f))
    for f in ("z", "beta"):
        np.testing.assert_array_equal(_field(L, out, f), getattr(var, f))
    assert float(_field(L, out, "sigma")[0, 0]) == sigma
    np.testing.assert_array_equal(_field(L, out, "kkt"), hist["kkt"])
    np.testing.assert_array_equal(_field(L, out, "iter").ravel(), hist["iter"])
    assert float(_field(L, out, "cScale")[0, 0]) == var.cScale and float(_field(L, out, "dScale")[0, 0]) == var.dScale
    # var.time (solver_socp_inPALM.m:339-341): the five step columns are device times, filled and consistent
    tm = _field(L, out, "times")
    assert tm.shape == (1, 7) and tm[0, 6] == K
    steps = tm[0, :5]
    nonzero = steps > 0
    assert nonzero[0] and nonzero[2] and nonzero[4], steps            # FFT, q-step, KKT ran in every loop variant
    assert nonzero[1] or method == "accADMM"                          # acc-ADMM books its cone pass as Step_3_2
    assert 0 < steps.sum() <= tm[0, 5] * 1.05 + 1e-3, (steps, tm[0, 5])   # device time inside the host wall time
    L.fmx_free(out)
    # a missing required field is a MATLAB error, not a crash
    bad = dict(opts)
    del bad["sigma"]
    err, _ = Call(L).run(g["dotsocp_inpalm_mex"], [S, bad], nlhs=1)
    assert err and err[0] == "dotsocp:inPALM" and "sigma" in err[1]
    # an array whose length disagrees with nx, ny, nt is refused before any pointer reaches the library
    for name in ("q", "beta", "c"):
        short = dict(S)
        short[name] = np.asfortranarray(np.asarray(S[name]).ravel(order="F")[:-3].copy())
        err, _ = Call(L).run(g["dotsocp_inpalm_mex"], [short, opts], nlhs=1)
        assert err and err[0] == "dotsocp:inPALM:size" and name in err[1], err
    if weighted:
        short = dict(S)
        short["weight"] = S["weight"][:-1].copy()
        err, _ = Call(L).run(g["dotsocp_inpalm_mex"], [short, opts], nlhs=1)
        assert err and err[0] == "dotsocp:inPALM:size" and "weight" in err[1], err


def test_solver_gateway_ngpu_is_the_single_process_multi_device_mode(mx):
    """opts.ngpu = N: N time slabs, slab r on device r mod #devices of this ONE process (dotsocp_create_multi), as a
    MATLAB host would drive several GPUs; on the one-GPU test box all slabs share the device.  The result must agree
    with ngpu = 1 to the slab tolerance and carry the same history."""
    L, g = mx
    rho0, rho1 = get_example_2d("example1", 24, 16)
    nt, K = 16, 20
    o = OD.default_opts(dict(tol=0.0, maxit=K), "inPALM", False)
    var, model = D.initialize(rho0, rho1, nt)
    D.InitialScaling(var, model, True, None, dim=2)

    def run(ngpu):
        S = dict(phi=var.phi.copy(), q=var.q.copy(), alpha=var.alpha.copy(), z=var.z.copy(order="F"),
                 beta=var.beta.copy(order="F"), c=model.c.copy(), nx=model.nx, ny=model.ny, nt=model.nt, D=var.D, E=var.E,
                 cScale=var.cScale, dScale=var.dScale, normc=model.normc, normd=model.normd)
        opts = dict(sigma=o["sigma"], maxit=K, tol=0.0, ifCheckStepByStep=0.0, scaling=1.0, tau=o["tau"], ngpu=float(ngpu))
        err, outs = Call(L).run(g["dotsocp_inpalm_mex"], [S, opts], nlhs=1)
        assert err is None, err
        res = {f: _field(L, outs[0], f) for f in ("phi", "q", "z", "alpha", "beta", "kkt", "iter", "times", "sigma")}
        L.fmx_free(outs[0])
        return res

    one, four = run(1), run(4)
    np.testing.assert_array_equal(four["iter"], one["iter"])
    np.testing.assert_allclose(four["kkt"], one["kkt"], rtol=1e-7, atol=1e-10)
    assert abs(four["sigma"][0, 0] - one["sigma"][0, 0]) <= 1e-12 * one["sigma"][0, 0]
    for f in ("phi", "q", "z", "alpha", "beta"):
        err = np.max(np.abs(four[f] - one[f])) / np.max(np.abs(one[f]))
        assert err <= 1e-10, (f, err)
    assert four["times"][0, 0] > 0 and four["times"][0, 6] == K
    # the multilevel driver passes one opts struct to every level (solver_dotsocp2d.m:208): a request for more slabs than
    # nt / 2 is served with nt / 2 of them (here 8) instead of an error, like the level gateway does
    many = run(64)
    np.testing.assert_array_equal(many["iter"], one["iter"])
    for f in ("phi", "q", "z", "alpha", "beta"):
        err = np.max(np.abs(many[f] - one[f])) / np.max(np.abs(one[f]))
        assert err <= 1e-10, (f, err)
    bad = dict(sigma=1.0, maxit=K, tol=0.0, ifCheckStepByStep=0.0, scaling=1.0, tau=1.9, ngpu=0.0)
    S = dict(phi=var.phi.copy(), q=var.q.copy(), alpha=var.alpha.copy(), z=var.z.copy(order="F"), beta=var.beta.copy(order="F"),
             c=model.c.copy(), nx=model.nx, ny=model.ny, nt=model.nt, D=var.D, E=var.E, cScale=var.cScale, dScale=var.dScale,
             normc=model.normc, normd=model.normd)
    err, _ = Call(L).run(g["dotsocp_inpalm_mex"], [S, bad], nlhs=1)
    assert err and "ngpu" in err[1]


def test_level_gateway_runs_the_multilevel_driver_on_the_device(mx):
    """dotsocp_level_mex: the call sequence a MATLAB solver_dotsocp2d.m would issue for levelN = 2 (create, solve,
    create-from-coarse, destroy, solve, outputs), with the host-side level logic of solver_dotsocp2d.m:154-250 done
    here in Python; must reproduce D.solver_dotsocp2d (same library underneath) exactly."""
    from dotsocp_amd import multilevel as ML
    L, g = mx
    so = os.path.join(OUT, "dotsocp_level_mex.so")
    libdir = os.path.join(ROOT, "dot-socp_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-shared", "-fPIC", "-I" + os.path.join(MEX, "compile_check"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + MEX, os.path.join(MEX, "dotsocp_level_mex.c"),
                           "-o", so, "-L" + libdir, "-ldotsocp", "-L" + OUT, "-lfake_mx",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + OUT])
    gate = ctypes.CDLL(so).mexFunction

    def call(*args, nlhs=1):
        err, outs = Call(L).run(gate, list(args), nlhs=nlhs)
        assert err is None, err
        return outs[0] if nlhs else None

    n, nt, tol = 33, 17, 1e-4
    rho0, rho1 = get_example_2d("example1", n, n)
    ref_out, ref_time, ref_histML, ref_hist = D.solver_dotsocp2d(rho0, rho1, nt, 2, dict(tol=tol), "inPALM")
    # ---- the driver's host side (solver_dotsocp2d.m:154-250) ----
    o = OD.default_opts(dict(tol=tol), "inPALM")
    r0c, r1c = ML.downSample_phi(rho0), ML.downSample_phi(rho1)
    r0c, r1c = r0c / (r0c.sum() / r0c.size), r1c / (r1c.sum() / r1c.size)
    ntc = (nt - 1) // 2 + 1
    tolc = max(tol * 2 ** (-0.5), 1e-4)

    def level_struct(var, model):
        S = dict(c=model.c.copy(), nx=model.nx, ny=model.ny, nt=model.nt, D=var.D, E=var.E, cScale=var.cScale,
                 dScale=var.dScale, normc=model.normc, normd=model.normd)
        if var.phi is not None:
            S["phi"] = var.phi.copy()
        return S

    def mopts(o, tol_):
        return dict(tau=o["tau"], sigma=o["sigma"], maxit=o["maxit"], tol=tol_, ifCheckStepByStep=0.0, scaling=1.0,
                    time_limit=o["time_limit"])

    var, model = D.initialize(r0c, r1c, ntc, lazy_zeros=True)
    D.InitialScaling(var, model, True, None, dim=2)
    h1 = Mx(call("create", level_struct(var, model), mopts(o, tolc)))
    out1 = call("solve", h1)
    kkt1, sigma1 = _field(L, out1, "kkt"), float(_field(L, out1, "sigma")[0, 0])
    iters1 = int(_field(L, out1, "iter")[-1, 0])
    o["sigma"] = 10 ** (np.log10(o["sigma"] * sigma1) / 2)
    E2 = var.E2
    var, model = D.initialize(rho0, rho1, nt, lazy_zeros=True)
    var.phi, var.E2 = None, E2
    model.n_global = model.c.size
    D.InitialScaling(var, model, True, kkt1[-1], dim=2)
    h2 = Mx(call("create", level_struct(var, model), mopts(o, tol), h1))
    call("destroy", h1, nlhs=0)
    out2 = call("solve", h2)
    iters2 = int(_field(L, out2, "iter")[-1, 0])
    outs = call("outputs", h2, np.asfortranarray(rho0), np.asfortranarray(rho1))
    st = call("fields", h2)
    call("destroy", h2, nlhs=0)
    assert [iters1, iters2] == [int(t["Iters"]) for t in ref_time[:2]]
    np.testing.assert_array_equal(_field(L, out2, "kkt"), ref_hist["kkt"])
    for k in ("rho", "Ex", "Ey", "q0", "bx", "by"):
        got = _field(L, outs, k).reshape(ref_out[k].shape, order="F")
        np.testing.assert_array_equal(got, ref_out[k], err_msg=k)
    assert _field(L, st, "beta").shape == (n * n * (nt - 1), 10) and np.all(np.isfinite(_field(L, st, "phi")))
    # a stale handle is an error, not a crash
    err, _ = Call(L).run(gate, ["solve", h2], nlhs=1)
    assert err and err[0] == "dotsocp:level"
    # a length that disagrees with the grid is refused before the level exists
    var, model = D.initialize(r0c, r1c, ntc, lazy_zeros=True)
    D.InitialScaling(var, model, True, None, dim=2)
    S = level_struct(var, model)
    S["c"] = S["c"][:-1].copy()
    err, _ = Call(L).run(gate, ["create", S, mopts(o, tolc)], nlhs=1)
    assert err and err[0] == "dotsocp:level:size" and "'c'" in err[1], err
    # `clear mex` with a level still open: the exit handler releases its device memory, the handle dies with it
    S = level_struct(var, model)
    h3 = Mx(call("create", S, mopts(o, tolc)))
    L.fmx_clear_mex()
    err, _ = Call(L).run(gate, ["solve", h3], nlhs=1)
    assert err and err[0] == "dotsocp:level" and "invalid level handle" in err[1]
