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
