"""The prime-factor DCT of csrc/pfa.hip, as its numpy prototype (tools/pfa_proto.py: Good-Thomas index maps, folded
small DFTs on real cos / sin matrices, Makhoul pre / post-processing, the in-place second transform of the fused
t-axis solve through the swapped index map) against scipy's orthonormal DCT-II / DCT-III -- for exactly the
factorisations the kernels are instantiated with.  CPU only: the kernel follows this prototype step by step and is
held against scipy and against the dense product on the GPU (tests/test_gpu_operators.py)."""
import math
import os
import re
import sys

import numpy as np
import scipy.fft as sf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pfa_proto import Pfa  # noqa: E402


def _kernel_factorisations():
    src = open(os.path.join(ROOT, "dot-socp_amd", "csrc", "pfa.hip")).read()
    pairs = re.findall(r"case (\d+): n1 = (\d+); n2 = (\d+); return true;", src)
    return [(int(n), int(a), int(b)) for n, a, b in pairs]


def test_factor_table_of_the_kernels():
    tab = _kernel_factorisations()
    assert {n for n, _, _ in tab} == {1025, 513, 129, 65, 33, 17, 9, 5, 3}
    for n, n1, n2 in tab:
        assert n1 * n2 == n and math.gcd(n1, n2) == 1 and n % 2 == 1
        # every length the multilevel driver produces from these by halving (solver_dotsocp2d.m:166-178:
        # n -> (n - 1) / 2 + 1) is again in the table -- except 257, which is prime (one factor of 257 does not fit a
        # thread's registers) and keeps the dense matrix-core product
        if n > 3:
            assert (n - 1) // 2 + 1 in {m for m, _, _ in tab} | {257}


def test_prime_factor_dct_matches_scipy():
    rng = np.random.default_rng(5)
    for n, n1, n2 in _kernel_factorisations():
        P = Pfa(n1, n2)
        xa, xb = rng.standard_normal(n), rng.standard_normal(n)
        fa, fb = P.dct2(xa, xb)
        np.testing.assert_allclose(fa, sf.dct(xa, norm="ortho"), atol=1e-13)
        np.testing.assert_allclose(fb, sf.dct(xb, norm="ortho"), atol=1e-13)
        ia, ib = P.dct3(xa, xb)
        np.testing.assert_allclose(ia, sf.idct(xa, norm="ortho"), atol=1e-13)
        np.testing.assert_allclose(ib, sf.idct(xb, norm="ortho"), atol=1e-13)
        la, lb = 1.0 + rng.random(n), 1.0 + rng.random(n)
        ta, tb = P.tsolve(xa, xb, la, lb)                  # second transform in place through the swapped map
        np.testing.assert_allclose(ta, sf.idct(sf.dct(xa, norm="ortho") / la, norm="ortho"), atol=1e-13)
        np.testing.assert_allclose(tb, sf.idct(sf.dct(xb, norm="ortho") / lb, norm="ortho"), atol=1e-13)


def test_index_maps_are_bijections_and_swap_roles():
    for n, n1, n2 in _kernel_factorisations():
        P = Pfa(n1, n2)
        for m in (P.mapA, P.mapB):
            pos = np.asarray(m[0]) * n2 + np.asarray(m[1])
            assert sorted(pos.tolist()) == list(range(n))
        # a transform fed through map A comes out through map B and vice versa: a unit impulse at p = 1 gives
        # exp(-2 pi i k / n) at the position map B assigns to k
        v = np.zeros(n, complex)
        v[1] = 1.0
        X = P.dft(v, "A")
        k = np.arange(n)
        np.testing.assert_allclose(X[P.mapB[0][k], P.mapB[1][k]], np.exp(-2j * np.pi * k / n), atol=1e-13)
        Y = P.dft(v, "B")
        np.testing.assert_allclose(Y[P.mapA[0][k], P.mapA[1][k]], np.exp(-2j * np.pi * k / n), atol=1e-13)
