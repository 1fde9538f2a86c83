"""ctypes binding of lib/libdotsocp.so (include/dotsocp.h).

There is no CPU fallback: if the shared library is missing this module raises at import of
the first symbol, and every compute entry point returns DOTSOCP_ENODEVICE without a GPU.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdotsocp.so")

i64 = ctypes.c_longlong
dbl = ctypes.c_double
vp = ctypes.c_void_p


class DotsocpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdotsocp error {code}: {msg}")
        self.code = code


class Problem(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int), ("weighted", ctypes.c_int),
                ("ny", i64), ("nx", i64), ("nt", i64),
                ("D", dbl), ("E", dbl), ("cScale", dbl), ("dScale", dbl),
                ("normc", dbl), ("normd", dbl)]


class Opts(ctypes.Structure):
    _fields_ = [("tau", dbl), ("sigma", dbl), ("tol", dbl), ("maxit", i64),
                ("ifCheckStepByStep", ctypes.c_int), ("checkPrimDualFeas", ctypes.c_int),
                ("scaling", ctypes.c_int), ("time_limit", dbl)]


class Result(ctypes.Structure):
    _fields_ = [("sigma", dbl), ("sigma_internal", dbl), ("cScale", dbl), ("dScale", dbl),
                ("times", dbl * 7), ("iters", i64), ("hist_len", i64), ("stopped", ctypes.c_int),
                ("time_extra", dbl)]


class AccOpts(ctypes.Structure):
    _fields_ = [("restart", i64), ("rho", dbl), ("theta", dbl)]


METHOD_INPALM, METHOD_PALM, METHOD_ACCADMM = 0, 1, 2


F_PHI, F_Q, F_ALPHA, F_Z, F_BETA, F_C, F_WEIGHT = range(7)

# every symbol include/dotsocp.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dotsocp_last_error": (ctypes.c_char_p, []),
    "dotsocp_version": (ctypes.c_char_p, []),
    "dotsocp_device_count": (ctypes.c_int, []),
    "dotsocp_proj_soc": (ctypes.c_int, [vp, vp, i64, i64]),
    "dotsocp_bfd": (ctypes.c_int, [vp, vp, i64, i64, i64, dbl, dbl]),
    "dotsocp_bfd_conj": (ctypes.c_int, [vp, vp, i64, i64, i64, dbl]),
    "dotsocp_bfd1d": (ctypes.c_int, [vp, vp, i64, i64, dbl, dbl]),
    "dotsocp_bfd_conj1d": (ctypes.c_int, [vp, vp, i64, i64, dbl]),
    "dotsocp_oper_poisson": (ctypes.c_int, [vp, vp, i64, i64, i64, dbl]),
    "dotsocp_dctn": (ctypes.c_int, [vp, i64, i64, i64, ctypes.c_int]),
    "dotsocp_proj_soc_dev": (ctypes.c_int, [vp, vp, i64, i64, vp]),
    "dotsocp_bfd_dev": (ctypes.c_int, [vp, vp, i64, i64, i64, dbl, dbl, vp]),
    "dotsocp_bfd_conj_dev": (ctypes.c_int, [vp, vp, i64, i64, i64, dbl, vp]),
    "dotsocp_create": (vp, [ctypes.POINTER(Problem), ctypes.c_int, ctypes.c_int]),
    "dotsocp_create_multi": (vp, [ctypes.POINTER(Problem), ctypes.c_int, ctypes.c_int]),
    "dotsocp_destroy": (None, [vp]),
    "dotsocp_rccl_unique_id": (ctypes.c_int, [vp]),
    "dotsocp_attach_rccl": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int]),
    "dotsocp_slab_range": (ctypes.c_int, [i64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(i64), ctypes.POINTER(i64)]),
    "dotsocp_field_len": (i64, [ctypes.POINTER(Problem), ctypes.c_int]),
    "dotsocp_release_cache": (i64, []),
    "dotsocp_upload": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "dotsocp_upload_layers": (ctypes.c_int, [vp, ctypes.c_int, vp, i64, i64]),
    "dotsocp_download": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "dotsocp_begin": (ctypes.c_int, [vp, ctypes.POINTER(Opts)]),
    "dotsocp_begin_method": (ctypes.c_int, [vp, ctypes.POINTER(Opts), ctypes.c_int, ctypes.POINTER(AccOpts)]),
    "dotsocp_run": (ctypes.c_int, [vp, i64, ctypes.POINTER(i64)]),
    "dotsocp_recover_outputs": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dotsocp_jump_next_level": (ctypes.c_int, [vp, vp]),
    "dotsocp_finish": (ctypes.c_int, [vp, ctypes.POINTER(Result)]),
    "dotsocp_get_history": (ctypes.c_int, [vp, vp, vp, vp, vp]),
    "dotsocp_set_profiling": (ctypes.c_int, [vp, ctypes.c_int]),
    "dotsocp_kernel_time": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.POINTER(dbl), ctypes.POINTER(i64)]),
    "dotsocp_canary_check": (ctypes.c_int, []),
    "dotsocp_synchronize": (ctypes.c_int, [vp]),
}

_lib = None


def _share_torchs_hip_runtime():
    """The PyTorch-ROCm wheel ships its own copy of the HIP runtime (same SONAME as the system one), and whichever copy
    a process loads first serves everything loaded later.  If libdotsocp came first, with the system runtime, a later
    `import torch` would bring a second runtime and find "no ROCm-capable device".  So when torch is installed but not
    yet imported, its copy is loaded here first: libdotsocp then binds to it (same SONAME) and a later `import torch`
    finds its own runtime already in place -- the import order no longer matters.  DOTSOCP_SYSTEM_HIP=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("DOTSOCP_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if not spec or not spec.submodule_search_locations:
        return
    for name in ("libamdhip64.so", "libamdhip64.so.7"):
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
            return


def lib():
    """Load libdotsocp.so (built by `make -C dot-socp_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        _share_torchs_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != 0:
        raise DotsocpError(code, lib().dotsocp_last_error().decode())


def fptr(a):
    """Pointer to a Fortran/C-contiguous float64 numpy array (no copy)."""
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not (a.flags.f_contiguous or a.flags.c_contiguous):
        raise ValueError("expected a contiguous float64 numpy array")
    return a.ctypes.data


def slab_range(nt, world, rank):
    t0, t1 = i64(), i64()
    check(lib().dotsocp_slab_range(nt, world, rank, ctypes.byref(t0), ctypes.byref(t1)))
    return t0.value, t1.value


def rccl_unique_id():
    """128-byte ncclUniqueId for dotsocp_attach_rccl(); call on rank 0 and broadcast to the others."""
    buf = (ctypes.c_ubyte * 128)()
    check(lib().dotsocp_rccl_unique_id(buf))
    return bytes(buf)
