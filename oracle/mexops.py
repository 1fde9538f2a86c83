"""ctypes front-end of oracle/mex_kernels.c with the reference's MEX names.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function mutates its first
argument in place, like the reference MEX files do with prhs[0]
(call sites socp/dot2d/algorithms/solver_socp_inPALM.m:
133,187,199,205,212,225,240,242).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "mex_kernels.c")
_SO = os.path.join(_HERE, "_build", "liboracle_mex.so")
_lib = None


def build(force=False):
    """Compile oracle/mex_kernels.c with gcc (no FMA contraction, strict IEEE)."""
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
             "-o", _SO, _SRC, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        dp, i64, d = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_double
        lib.oracle_proj_soc.argtypes = [dp, dp, i64, i64, dp]
        lib.oracle_bfd.argtypes = [dp, dp, i64, i64, i64, d, d]
        lib.oracle_bfd_conj.argtypes = [dp, dp, i64, i64, i64, d]
        lib.oracle_bfd1d.argtypes = [dp, dp, i64, i64, d, d]
        lib.oracle_bfd_conj1d.argtypes = [dp, dp, i64, i64, d]
        for f in (lib.oracle_proj_soc, lib.oracle_bfd, lib.oracle_bfd_conj,
                  lib.oracle_bfd1d, lib.oracle_bfd_conj1d):
            f.restype = None
        _lib = lib
    return _lib


def _colmajor(a, name):
    if a.dtype != np.float64 or not a.flags.f_contiguous:
        raise ValueError(f"{name} must be a Fortran-contiguous float64 array (MATLAB layout)")
    return a.ctypes.data


def mexProjSoc(out, inp):
    """mexProjSoc(out, in): rows of the M x K matrix `inp` projected onto the SOC."""
    M, K = inp.shape
    assert out.shape == inp.shape
    tmp = np.empty(2 * M)
    _load().oracle_proj_soc(_colmajor(out, "out"), _colmajor(inp, "in"), M, K, tmp.ctypes.data)


def mexBFd(z, q, nt, nx, ny, scale=1.0, dF=1.0):
    assert z.shape == (ny * nx * (nt - 1), 10)
    assert q.size == ny * nx * (nt - 1) + ny * (nx - 1) * nt + (ny - 1) * nx * nt
    _load().oracle_bfd(_colmajor(z, "z"), _colmajor(q, "q"), int(nt), int(nx), int(ny), scale, dF)


def mexBFdConj(q, z, nt, nx, ny, scale=1.0):
    assert z.shape == (ny * nx * (nt - 1), 10)
    assert q.size == ny * nx * (nt - 1) + ny * (nx - 1) * nt + (ny - 1) * nx * nt
    _load().oracle_bfd_conj(_colmajor(q, "q"), _colmajor(z, "z"), int(nt), int(nx), int(ny), scale)


def mexBFd1d(z, q, nt, nx, scale=1.0, dF=1.0):
    assert z.shape == (nx * (nt - 1), 6)
    assert q.size == nx * (nt - 1) + (nx - 1) * nt
    _load().oracle_bfd1d(_colmajor(z, "z"), _colmajor(q, "q"), int(nt), int(nx), scale, dF)


def mexBFdConj1d(q, z, nt, nx, scale=1.0):
    assert z.shape == (nx * (nt - 1), 6)
    assert q.size == nx * (nt - 1) + (nx - 1) * nt
    _load().oracle_bfd_conj1d(_colmajor(q, "q"), _colmajor(z, "z"), int(nt), int(nx), scale)
