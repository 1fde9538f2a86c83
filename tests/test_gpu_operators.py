"""Operator-level parity (boundary B2): the HIP kernels behind mexProjSoc / mexBFd / mexBFdConj /
mexBFd1d / mexBFdConj1d / oper_poisson3dim against the CPU oracle on identical seeded inputs.
All calls go through the C ABI (ctypes -> lib/libdotsocp.so)."""
import numpy as np
import pytest
import scipy.fft as sfft

import dotsocp_amd as D
from oracle import mexops as O
from oracle.model import initialize_FFTkernel, oper_poisson

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(2024)


def _sizes(nt, nx, ny):
    Nz = ny * nx * (nt - 1)
    return Nz, Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt


@pytest.mark.parametrize("M,K", [(1, 10), (777, 10), (100000, 10), (513, 6), (64, 2), (300, 13)])
def test_proj_soc_bit_exact(M, K):
    x = np.asfortranarray(rng.standard_normal((M, K)) * rng.choice([0.1, 1, 30], size=(M, 1)))
    ref = np.empty_like(x, order="F")
    got = np.full_like(x, -7.0, order="F")
    O.mexProjSoc(ref, x)
    D.mexProjSoc(got, x)
    # same operation order, no FMA contraction on either side: results are identical
    assert np.array_equal(got, ref)


def test_proj_soc_edge_rows():
    """zero row -> NaN, n = 0 with x1 > 0 -> unchanged, x1 < 0 -> zero, boundary of the cone, +-inf"""
    x = np.zeros((7, 10), order="F")
    x[1, 0] = 3.0
    x[2, 0] = -3.0
    x[3, :2] = [1.0, 1.0]
    x[4, :2] = [-1.0, 1.0]
    x[5] = 1e-200
    x[6] = 1e150
    ref, got = np.empty_like(x, order="F"), np.empty_like(x, order="F")
    O.mexProjSoc(ref, x)
    D.mexProjSoc(got, x)
    assert np.all(np.isnan(got[0])) and np.array_equal(got[1], x[1]) and np.all(got[2] == 0)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("nt,nx,ny", [(2, 1, 1), (2, 2, 2), (4, 6, 5), (3, 70, 130), (9, 17, 64), (5, 1, 100), (5, 100, 1)])
def test_bfd_and_conj(nt, nx, ny):
    Nz, Nq = _sizes(nt, nx, ny)
    s, dF = 0.731, 1.37
    q = rng.standard_normal(Nq)
    z0 = np.asfortranarray(rng.standard_normal((Nz, 10)))        # sentinel values in unwritten slots
    ref, got = z0.copy(order="F"), z0.copy(order="F")
    O.mexBFd(ref, q, nt, nx, ny, s, dF)
    D.mexBFd(got, q, nt, nx, ny, s, dF)
    assert np.array_equal(got, ref)
    w = np.asfortranarray(rng.standard_normal((Nz, 10)))
    qr, qg = np.zeros(Nq), np.full(Nq, 5.0)
    O.mexBFdConj(qr, w, nt, nx, ny, s)
    D.mexBFdConj(qg, w, nt, nx, ny, s)
    assert np.array_equal(qg, qr)
    # adjoint identity on the device results themselves
    z = np.zeros((Nz, 10), order="F")
    D.mexBFd(z, q, nt, nx, ny, s, 0.0)
    assert abs(np.vdot(z, w) - np.vdot(q, qg)) <= 1e-12 * (np.linalg.norm(z) * np.linalg.norm(w) + 1)


def test_bfd_defaults():
    nt, nx, ny = 3, 4, 5
    Nz, Nq = _sizes(nt, nx, ny)
    q = rng.standard_normal(Nq)
    a, b = np.zeros((Nz, 10), order="F"), np.zeros((Nz, 10), order="F")
    D.mexBFd(a, q, float(nt), float(nx), float(ny))              # doubles are truncated, scale = dF = 1
    O.mexBFd(b, q, nt, nx, ny, 1.0, 1.0)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("nt,nx", [(2, 2), (4, 9), (33, 129), (5, 300)])
def test_bfd1d_and_conj1d(nt, nx):
    Nz = nx * (nt - 1)
    Nq = Nz + (nx - 1) * nt
    s, dF = 1.21, 0.6
    q = rng.standard_normal(Nq)
    z0 = np.asfortranarray(rng.standard_normal((Nz, 6)))
    ref, got = z0.copy(order="F"), z0.copy(order="F")
    O.mexBFd1d(ref, q, nt, nx, s, dF)
    D.mexBFd1d(got, q, nt, nx, s, dF)
    assert np.array_equal(got, ref)
    w = np.asfortranarray(rng.standard_normal((Nz, 6)))
    qr, qg = np.zeros(Nq), np.zeros(Nq)
    O.mexBFdConj1d(qr, w, nt, nx, s)
    D.mexBFdConj1d(qg, w, nt, nx, s)
    assert np.array_equal(qg, qr)


@pytest.mark.parametrize("shape", [(8, 4, 2), (64, 32, 16), (256, 8, 4), (16, 256, 8), (4, 16, 128), (1024, 2, 2),
                                   (5, 6, 7), (33, 17, 9), (129, 3, 2), (16, 1, 8), (129, 1, 33),
                                   (129, 65, 33), (65, 129, 40), (257, 257, 5),
                                   # prime-factor lengths along every axis, odd line counts, partial tiles
                                   (1025, 3, 2), (7, 1025, 3), (3, 5, 1025), (513, 11, 3), (20, 513, 2), (17, 9, 5),
                                   (5, 3, 9), (3, 3, 3), (65, 33, 17), (1025, 1, 1), (1, 513, 1), (1, 1, 129),
                                   (96, 100, 48), (50, 70, 66), (192, 3, 80)])   # even lengths that are no powers of two
def test_dctn_matches_scipy(shape):
    a = np.asfortranarray(rng.standard_normal(shape))
    tol = 2e-13 * np.sqrt(np.prod(shape))
    np.testing.assert_allclose(D.mirt_dctn(a), sfft.dctn(a, norm="ortho"), atol=tol)
    np.testing.assert_allclose(D.mirt_idctn(a), sfft.idctn(a, norm="ortho"), atol=tol)
    np.testing.assert_allclose(D.mirt_idctn(D.mirt_dctn(a)), a, atol=tol)


def test_prime_factor_dct_against_the_dense_product():
    """The 2^k+1 lengths (1025 = 25 x 41, 513 = 27 x 19, 129 = 3 x 43, 65, 33, 17, 9, 5, 3) take the prime-factor
    transform of csrc/pfa.hip; DOTSOCP_PFA=0 (read once per process, hence the subprocess) sends them through the dense
    DCT-matrix product instead.  Two different algorithms for the same transform: results agree to rounding, for the
    transforms along every axis and for the Poisson solve (fused t-axis pass against three separate ones)."""
    import os
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, numpy as np, dotsocp_amd as D\n"
        "rng = np.random.default_rng(12)\n"
        "out = {}\n"
        "for i, shape in enumerate([(1025, 6, 3), (10, 1025, 3), (513, 9, 17), (18, 513, 5), (33, 65, 129), (129, 17, 9), (5, 3, 1025)]):\n"
        "    a = np.asfortranarray(rng.standard_normal(shape))\n"
        "    out['f%d' % i] = D.mirt_dctn(a); out['i%d' % i] = D.mirt_idctn(a)\n"
        "    out['p%d' % i] = D.oper_poisson3dim(0.37 ** 2, a)\n"
        "np.savez(sys.argv[1], **out)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for flag in ("0", "1"):
            path = os.path.join(tmp, f"pfa{flag}.npz")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, DOTSOCP_PFA=flag), cwd=root,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-2000:]
            with np.load(path) as z:
                res[flag] = {k: z[k].copy() for k in z.files}
    for k in res["0"]:
        ref = res["0"][k]
        np.testing.assert_allclose(res["1"][k], ref, rtol=0, atol=2e-13 * max(1.0, np.abs(ref).max()) * np.sqrt(ref.size), err_msg=k)


@pytest.mark.parametrize("shape", [(1024, 1024, 16), (512, 2048, 8), (256, 4096, 8), (128, 8192, 8), (1024, 1023, 9),
                                   # every length of the pipelined (LDS-DMA) kernels along every axis
                                   (512, 512, 64), (256, 256, 256), (128, 128, 1024), (1024, 128, 128),
                                   # tile counts that do not divide by the persistent grid: workgroups with two and with
                                   # three tiles (first / steady / last wait counts of the LDS-DMA pipeline)
                                   (1024, 1024, 5), (512, 1024, 9),
                                   # 2048-point lines in the pipelined kernels (quarter / half tables in LDS, tiles of two pairs)
                                   (2048, 512, 4), (512, 2048, 5), (2048, 2048, 3)])
def test_dctn_many_lines(shape):
    """Many lines per axis (every workgroup of the chip busy several times over), odd line counts, dense x / t axes."""
    a = np.asfortranarray(rng.standard_normal(shape))
    np.testing.assert_allclose(D.mirt_dctn(a), sfft.dctn(a, norm="ortho"), atol=2e-12)
    np.testing.assert_allclose(D.mirt_idctn(a), sfft.idctn(a, norm="ortho"), atol=2e-12)


@pytest.mark.parametrize("ny,nx,nt", [(16, 16, 8), (64, 32, 16), (9, 5, 3), (33, 33, 17), (128, 1, 32), (65, 65, 33),
                                      (129, 64, 17), (129, 129, 33), (1025, 1, 33), (513, 40, 129), (40, 1025, 9),
                                      (64, 64, 65), (7, 3, 513),
                                      # large enough for the pipelined kernels (fused t-axis solve of length 128 .. 1024)
                                      (256, 128, 128), (128, 512, 256), (64, 512, 512), (32, 1024, 1024),
                                      (256, 160, 128), (64, 1000, 128),        # tile counts that leave a remainder
                                      # short time axes in the pipelined t pass (tiles of 64 / 32 pairs of columns)
                                      (512, 512, 32), (256, 1024, 64), (1024, 300, 32),
                                      # the t axis as tridiagonal systems (tri.hip): one tile per workgroup (any nt <= 512 that
                                      # is no power of two) and the persistent LDS-DMA flavour (64 < nt <= 136, >= 4096 tiles;
                                      # there also nt = 128)
                                      (48, 40, 49), (64, 64, 97), (33, 31, 257), (40, 24, 300), (512, 512, 72), (514, 520, 129),
                                      (1024, 512, 128), (513, 511, 66)])
def test_oper_poisson(ny, nx, nt):
    Dsc = 0.37
    rhs = np.asfortranarray(rng.standard_normal((ny, nx, nt)))
    if nx == 1:
        kernel = Dsc ** 2 * initialize_FFTkernel(nt, ny)          # 1-D problem: grid nx1d x nt
        ref = oper_poisson(kernel, rhs.reshape((ny, nt), order="F")).ravel(order="F")
    else:
        kernel = Dsc ** 2 * initialize_FFTkernel(nt, nx, ny)
        ref = oper_poisson(kernel, rhs).ravel(order="F")
    got = D.oper_poisson3dim(Dsc ** 2, rhs)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("ny,nx,nt", [(512, 512, 72), (1024, 512, 128), (514, 520, 129)])
def test_tridiagonal_t_solve_flavours_agree(ny, nx, nt, monkeypatch):
    """The t axis of the Poisson solve as tridiagonal systems: the persistent LDS-DMA kernel (k_tsolve_pipe), the
    one-tile-per-workgroup kernel (DOTSOCP_TS_PIPE=0) -- the same arithmetic on the same registers, bit for bit -- and the
    transform passes along t (DOTSOCP_TSOLVE=dct), another algorithm for the same linear systems: rounding."""
    rhs = np.asfortranarray(np.random.default_rng(5).standard_normal((ny, nx, nt)))
    a = D.oper_poisson3dim(0.37 ** 2, rhs)
    monkeypatch.setenv("DOTSOCP_TS_PIPE", "0")
    b = D.oper_poisson3dim(0.37 ** 2, rhs)
    monkeypatch.setenv("DOTSOCP_TSOLVE", "dct")
    c = D.oper_poisson3dim(0.37 ** 2, rhs)
    if nt & (nt - 1):
        np.testing.assert_array_equal(a, b)
    else:       # a power of two without the pipelined flavour goes back to the transform pass (tsolve_tri_preferred)
        np.testing.assert_array_equal(b, c)
    np.testing.assert_allclose(a, c, rtol=0, atol=2e-12 * np.abs(c).max())


def test_pipelined_dct_kernels_against_the_workgroup_wide_ones():
    """DOTSOCP_DCT_PIPE=0 (read once per process, hence the subprocesses) switches the persistent LDS-DMA kernels off; both
    families run the same butterflies on the same operands, so transforms and Poisson solves agree to rounding of the
    few places where the order of operations differs or where the compiler contracts a * b + c differently in the two
    families (dct.hip is built with -ffp-contract=fast since round 3: the FFT has no operation-by-operation counterpart
    in the reference)."""
    import os
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, numpy as np, dotsocp_amd as D\n"
        "rng = np.random.default_rng(11)\n"
        "a = np.asfortranarray(rng.standard_normal((1024, 512, 16)))\n"
        "b = np.asfortranarray(rng.standard_normal((256, 256, 128)))\n"
        "np.savez(sys.argv[1], f=D.mirt_dctn(a), i=D.mirt_idctn(a), p=D.oper_poisson3dim(0.37 ** 2, b))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for flag in ("0", "1"):
            path = os.path.join(tmp, f"dct{flag}.npz")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, DOTSOCP_DCT_PIPE=flag), cwd=root,
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            with np.load(path) as z:
                out[flag] = {k: z[k].copy() for k in z.files}
    np.testing.assert_allclose(out["1"]["f"], out["0"]["f"], rtol=0, atol=1e-13 * np.abs(out["0"]["f"]).max())
    np.testing.assert_allclose(out["1"]["i"], out["0"]["i"], rtol=0, atol=1e-13 * np.abs(out["0"]["i"]).max())
    np.testing.assert_allclose(out["1"]["p"], out["0"]["p"], rtol=0, atol=1e-13 * np.abs(out["0"]["p"]).max())
