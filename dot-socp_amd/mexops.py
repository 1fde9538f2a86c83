"""Operator-level boundary B2: the reference's MEX operators and Poisson helper, same names,
argument order and in-place behaviour, executed by the HIP kernels of libdotsocp.

Reference call sites: socp/dot2d/algorithms/solver_socp_inPALM.m:133,187,194,199,205,212,225,240,242;
argument conventions: SURVEY.md section 8b (B2).  Arrays are float64 numpy arrays in MATLAB
(Fortran) order; like the MEX files, the first argument is overwritten in place.
"""
import numpy as np

from . import capi


def _check_scalar(x, name):
    if np.ndim(x) != 0:
        # mexBFd1d raises mexBFd:invalidInput for a non-scalar scale
        raise ValueError(f"mexBFd:invalidInput: {name} must be a scalar")
    return float(x)


def mexProjSoc(out, inp):
    """mexProjSoc(out, in): row-wise projection of the M x K matrix onto the second-order cone."""
    if out.shape != inp.shape or inp.ndim != 2:
        raise ValueError("mexProjSoc: out and in must be M x K matrices of equal size")
    M, K = inp.shape
    capi.check(capi.lib().dotsocp_proj_soc(capi.fptr(out), capi.fptr(np.asfortranarray(inp)), M, K))


def _sizes2d(nt, nx, ny):
    Nz = ny * nx * (nt - 1)
    return Nz, Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt


def mexBFd(z, q, nt, nx, ny, scale=1.0, dF=1.0):
    """mexBFd(z, q, nt, nx, ny[, scale=1[, dF=1]]):  z <- B F q + d."""
    nt, nx, ny = int(nt), int(nx), int(ny)          # doubles are truncated to int like cvttsd2si
    Nz, Nq = _sizes2d(nt, nx, ny)
    if z.shape != (Nz, 10) or q.size != Nq or not z.flags.f_contiguous:
        raise ValueError("mexBFd: z must be Nz x 10 (column-major) and q of length Nq")
    capi.check(capi.lib().dotsocp_bfd(capi.fptr(z), capi.fptr(q), nt, nx, ny,
                                      _check_scalar(scale, "scale"), _check_scalar(dF, "dF")))


def mexBFdConj(q, z, nt, nx, ny, scale=1.0):
    """mexBFdConj(q, z, nt, nx, ny[, scale=1]):  q <- F* B* z."""
    nt, nx, ny = int(nt), int(nx), int(ny)
    Nz, Nq = _sizes2d(nt, nx, ny)
    if z.shape != (Nz, 10) or q.size != Nq or not z.flags.f_contiguous:
        raise ValueError("mexBFdConj: z must be Nz x 10 (column-major) and q of length Nq")
    capi.check(capi.lib().dotsocp_bfd_conj(capi.fptr(q), capi.fptr(z), nt, nx, ny, _check_scalar(scale, "scale")))


def mexBFd1d(z, q, nt, nx, scale=1.0, dF=1.0):
    """mexBFd1d(z, q, nt, nx[, scale[, dF]]) -- 1-D grid, z is Nz x 6."""
    nt, nx = int(nt), int(nx)
    Nz = nx * (nt - 1)
    if z.shape != (Nz, 6) or q.size != Nz + (nx - 1) * nt or not z.flags.f_contiguous:
        raise ValueError("mexBFd:invalidInput: z must be Nz x 6 (column-major) and q of length Nq")
    capi.check(capi.lib().dotsocp_bfd1d(capi.fptr(z), capi.fptr(q), nt, nx,
                                        _check_scalar(scale, "scale"), _check_scalar(dF, "dF")))


def mexBFdConj1d(q, z, nt, nx, scale=1.0):
    """mexBFdConj1d(q, z, nt, nx[, scale])."""
    nt, nx = int(nt), int(nx)
    Nz = nx * (nt - 1)
    if z.shape != (Nz, 6) or q.size != Nz + (nx - 1) * nt or not z.flags.f_contiguous:
        raise ValueError("mexBFd:invalidInput: z must be Nz x 6 (column-major) and q of length Nq")
    capi.check(capi.lib().dotsocp_bfd_conj1d(capi.fptr(q), capi.fptr(z), nt, nx, _check_scalar(scale, "scale")))


def _dims3(a):
    shp = a.shape + (1,) * (3 - a.ndim)
    if a.ndim == 2:                 # (nx, nt) 1-D problem -> ny = nx1d, nx = 1
        shp = (a.shape[0], 1, a.shape[1])
    return shp


def mirt_dctn(a):
    """Orthonormal DCT-II along every axis (socp/dot2d/utils/mirt_dctn.m); returns a new array."""
    out = np.array(a, dtype=np.float64, order="F", copy=True)
    ny, nx, nt = _dims3(out)
    capi.check(capi.lib().dotsocp_dctn(capi.fptr(out), ny, nx, nt, 0))
    return out


def mirt_idctn(a):
    """Orthonormal DCT-III along every axis (socp/dot2d/utils/mirt_idctn.m)."""
    out = np.array(a, dtype=np.float64, order="F", copy=True)
    ny, nx, nt = _dims3(out)
    capi.check(capi.lib().dotsocp_dctn(capi.fptr(out), ny, nx, nt, 1))
    return out


def oper_poisson3dim(kernelScale, rhs):
    """res = oper_poisson3dim(kernelScale * initialize_FFTkernel(nt,nx,ny), rhs)
    (socp/dot2d/utils/oper_poisson3dim.m:4; 1-D: socp/dot1d/utils/oper_poisson.m:4 with a 2-D rhs).
    The spectral kernel is generated on the device from the grid sizes, so only its scalar
    factor (D^2 in solver_socp_inPALM.m:96) is passed."""
    rhs = np.asfortranarray(rhs, dtype=np.float64)
    ny, nx, nt = _dims3(rhs)
    res = np.empty(rhs.size)
    capi.check(capi.lib().dotsocp_oper_poisson(capi.fptr(res), capi.fptr(rhs), ny, nx, nt, float(kernelScale)))
    return res


oper_poisson = oper_poisson3dim
