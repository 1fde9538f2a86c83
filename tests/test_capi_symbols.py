"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/dotsocp.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import dotsocp_amd
from dotsocp_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "dotsocp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dotsocp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    names = _declared_functions()
    assert len(names) >= 25
    L = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/dotsocp.h but not exported: {missing}"
    # and the ctypes table binds exactly the declared set
    assert sorted(capi.SYMBOLS) == names


def test_version_and_slab_range():
    L = capi.lib()
    assert b"gfx950" in L.dotsocp_version()
    for nt, world in [(128, 8), (129, 8), (33, 4), (64, 1), (17, 3)]:
        edges = [capi.slab_range(nt, world, r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == nt
        for a, b in zip(edges[:-1], edges[1:]):
            assert a[1] == b[0]
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 2
    with pytest.raises(capi.DotsocpError):
        capi.slab_range(5, 4, 0)          # fewer than two time nodes per slab


def test_argument_validation_without_gpu():
    z = np.zeros((4, 10), order="F")
    with pytest.raises(ValueError):
        dotsocp_amd.mexBFd(z, np.zeros(3), 3, 2, 2)
    with pytest.raises(ValueError):
        dotsocp_amd.mexBFd1d(np.zeros((4, 6), order="F"), np.zeros(4 + 3), 3, 2, scale=np.ones(2))


@pytest.mark.skipif(capi.lib().dotsocp_device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    x = np.ones((3, 10), order="F")
    with pytest.raises(capi.DotsocpError) as e:
        dotsocp_amd.mexProjSoc(np.zeros_like(x, order="F"), x)
    assert e.value.code == -2
    var, model = dotsocp_amd.initialize(np.ones((4, 4)), np.ones((4, 4)), 4)
    dotsocp_amd.InitialScaling(var, model, True)
    with pytest.raises(capi.DotsocpError):
        dotsocp_amd.solver_socp_inPALM(var, dict(tau=1.9, sigma=1.0, tol=1e-3, maxit=3), model)


def test_mex_gateways_build_and_reject_bad_calls_without_a_gpu():
    """The MEX gateways link against libdotsocp and a stand-in libmx (tests/fake_mx), export mexFunction, and their
    argument checks fire before any device work (so this runs on the CPU-only box; the compute paths are in
    tests/test_gpu_mex_gateways.py)."""
    import ctypes
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mexdir = os.path.join(root, "dot-socp_amd", "mex")
    out = os.path.join(root, "tests", "fake_mx", "_build")
    os.makedirs(out, exist_ok=True)
    fake = os.path.join(out, "libfake_mx.so")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-shared", "-fPIC", "-o", fake,
                           os.path.join(root, "tests", "fake_mx", "fake_mx.c")])
    L = ctypes.CDLL(fake, mode=ctypes.RTLD_GLOBAL)
    vp = ctypes.c_void_p
    L.fmx_wrap_double.restype, L.fmx_wrap_double.argtypes = vp, [ctypes.c_size_t, ctypes.c_size_t, vp]
    L.fmx_call.restype = ctypes.c_int
    L.fmx_call.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(vp)]
    L.fmx_error_id.restype = ctypes.c_char_p
    libdir = os.path.join(root, "dot-socp_amd", "lib")
    for src in ("mexProjSoc", "mexBFd", "mexBFdConj", "mexBFd1d", "mexBFdConj1d", "dotsocp_inpalm_mex",
                "dotsocp_level_mex"):
        so = os.path.join(out, src + ".so")
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-shared", "-fPIC", "-I" + os.path.join(mexdir, "compile_check"),
                               "-I" + os.path.join(root, "include"), "-I" + mexdir, os.path.join(mexdir, src + ".c"),
                               "-o", so, "-L" + libdir, "-ldotsocp", "-L" + out, "-lfake_mx",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath," + out])
        fn = ctypes.CDLL(so).mexFunction
        assert fn
        if src == "mexBFd1d":
            z, q, s = np.zeros(12), np.zeros(20), np.array([4.0])
            hs = [L.fmx_wrap_double(a.size, 1, a.ctypes.data) for a in (z, q, s)]
            prhs, plhs = (vp * 3)(*hs), (vp * 1)()
            assert L.fmx_call(ctypes.cast(fn, vp), 0, plhs, 3, prhs) == 1
            assert L.fmx_error_id() == b"mexBFd:invalidNumInputs"


def test_field_len_is_pure_host_arithmetic():
    """dotsocp_field_len (what the MEX gateways check their arrays against) needs no device."""
    from dotsocp_amd import capi
    L = capi.lib()
    p = capi.Problem()
    p.dim, p.weighted, p.ny, p.nx, p.nt = 2, 0, 5, 6, 4
    Nphi, Nz = 5 * 6 * 4, 5 * 6 * 3
    Nq = Nz + 5 * 5 * 4 + 4 * 6 * 4
    want = {capi.F_PHI: Nphi, capi.F_C: Nphi, capi.F_Q: Nq, capi.F_ALPHA: Nq, capi.F_WEIGHT: Nq, capi.F_Z: 10 * Nz,
            capi.F_BETA: 10 * Nz}
    for f, n in want.items():
        assert L.dotsocp_field_len(p, f) == n
    assert L.dotsocp_field_len(p, 99) == -1
    p.dim, p.ny, p.nx, p.nt = 1, 0, 9, 5                      # 1-D: grid nx x nt, cone Nz x 6
    assert L.dotsocp_field_len(p, capi.F_Q) == 9 * 4 + 8 * 5 and L.dotsocp_field_len(p, capi.F_BETA) == 6 * 9 * 4
    p.nt = 1
    assert L.dotsocp_field_len(p, capi.F_PHI) == -1
    assert L.dotsocp_canary_check() == 0                      # nothing guarded, nothing damaged
