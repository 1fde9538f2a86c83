"""numpy prototype of the prime-factor DCT used by csrc/pfa.hip for the 2^k+1 grid lengths: index maps, folded
small DFTs, Makhoul pre/post-processing -- checked against scipy.  Development aid; the kernel follows it step by step."""
import numpy as np
import scipy.fft as sf


def egcd_inv(a, m):
    return pow(a % m, -1, m) if m > 1 else 0


class Pfa:
    def __init__(self, N1, N2):
        self.N1, self.N2, self.N = N1, N2, N1 * N2
        self.i2 = egcd_inv(N2, N1)      # N2^-1 mod N1
        self.i1 = egcd_inv(N1, N2)      # N1^-1 mod N2
        p = np.arange(self.N)
        # map A: p -> (p mod N1, p mod N2); map B: p -> (p i2 mod N1, p i1 mod N2)
        self.mapA = (p % N1, p % N2)
        self.mapB = ((p * self.i2) % N1 if N1 > 1 else 0 * p, (p * self.i1) % N2 if N2 > 1 else 0 * p)
        k = np.arange(self.N)
        sc = 2.0 / np.sqrt(2.0 * self.N)
        self.ww = sc * np.exp(-1j * np.pi * k / (2.0 * self.N))
        self.ww[0] /= np.sqrt(2.0)

    @staticmethod
    def folded_dft(z, axis):
        """DFT of odd length M along `axis` with real cos / sin matrices of size H x H, H = (M-1)/2"""
        z = np.moveaxis(z, axis, -1)
        M = z.shape[-1]
        if M == 1:
            return np.moveaxis(z.copy(), -1, axis)
        H = (M - 1) // 2
        j = np.arange(1, H + 1)
        th = 2.0 * np.pi * ((j[:, None] * j[None, :]) % M) / M
        C, S = np.cos(th), np.sin(th)            # [k][j]
        out = np.empty_like(z)
        for part in (0, 1):
            P = z.real if part == 0 else z.imag
            Q = z.imag if part == 0 else z.real
            sgn = 1.0 if part == 0 else -1.0
            e = P[..., 1:H + 1] + P[..., :H:-1]                 # P[j] + P[M-j]
            o = sgn * (Q[..., 1:H + 1] - Q[..., :H:-1])
            u0 = P[..., 0]
            CE = e @ C.T
            SO = o @ S.T
            res = np.empty(P.shape)
            res[..., 0] = u0 + e.sum(-1)
            res[..., 1:H + 1] = u0[..., None] + CE + SO
            res[..., :H:-1] = u0[..., None] + CE - SO
            if part == 0:
                out.real = res
            else:
                out.imag = res
        return np.moveaxis(out, -1, axis)

    def dft(self, v, in_map):
        """forward DFT of v (length N); in_map 'A' -> output at map B positions, and vice versa.  Returns the
        [N1][N2] array after both stages (the caller un-maps)."""
        N1, N2 = self.N1, self.N2
        m = self.mapA if in_map == "A" else self.mapB
        X = np.zeros((N1, N2), complex)
        X[m[0], m[1]] = v
        X = self.folded_dft(X, 1)
        X = self.folded_dft(X, 0)
        return X

    def makhoul(self, n):
        k = np.arange(n)
        return np.where(k % 2 == 0, k // 2, n - 1 - k // 2)

    def dct2(self, xa, xb):
        N = self.N
        pos = self.makhoul(N)
        v = np.zeros(N, complex)
        v[pos] = xa + 1j * xb
        X = self.dft(v, "A")
        mo = self.mapB
        k = np.arange(N)
        Vk = X[mo[0][k], mo[1][k]]
        Vm = X[mo[0][(N - k) % N], mo[1][(N - k) % N]]
        Va = 0.5 * (Vk + np.conj(Vm))
        Vb = (Vk - np.conj(Vm)) / 2j
        return (self.ww * Va).real, (self.ww * Vb).real

    def dct3(self, Xa, Xb):
        N = self.N
        ww = self.ww
        k = np.arange(N)
        X = Xa + 1j * Xb        # componentwise real lines
        G = np.empty(N, complex)
        # G[k] = (ww[k] X[k] + conj(ww[N-k]) X[N-k]) / 2 for each real line; packed a + i b
        def g(Xr):
            out = np.empty(N, complex)
            out[0] = ww[0] * Xr[0]
            out[1:] = 0.5 * (ww[1:] * Xr[1:] + np.conj(ww[:0:-1]) * Xr[:0:-1])
            return out
        G = g(Xa) + 1j * g(Xb)
        Y = self.dft(G, "A")
        mo = self.mapB
        pos = self.makhoul(N)
        y = Y[mo[0][pos], mo[1][pos]]
        return y.real, y.imag

    def tsolve(self, xa, xb, lam_a, lam_b):
        """idct(dct(x) / lam) with the second transform in place (input at map B positions, output at map A)"""
        N = self.N
        ww = self.ww
        pos = self.makhoul(N)
        v = np.zeros(N, complex)
        v[pos] = xa + 1j * xb
        X = self.dft(v, "A")
        mo = self.mapB
        k = np.arange(N)
        Vk = X[mo[0][k], mo[1][k]]
        Vm = X[mo[0][(N - k) % N], mo[1][(N - k) % N]]
        Xa = (ww * 0.5 * (Vk + np.conj(Vm))).real / lam_a
        Xb = (ww * (Vk - np.conj(Vm)) / 2j).real / lam_b
        def g(Xr):
            out = np.empty(N, complex)
            out[0] = ww[0] * Xr[0]
            out[1:] = 0.5 * (ww[1:] * Xr[1:] + np.conj(ww[:0:-1]) * Xr[:0:-1])
            return out
        G = g(Xa) + 1j * g(Xb)
        X2 = np.zeros((self.N1, self.N2), complex)
        X2[mo[0], mo[1]] = G                       # in place: same positions the spectrum was read from
        X2 = self.folded_dft(X2, 1)
        X2 = self.folded_dft(X2, 0)
        ma = self.mapA
        y = X2[ma[0][pos], ma[1][pos]]
        return y.real, y.imag


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for N1, N2 in [(25, 41), (27, 19), (3, 43), (5, 13), (3, 11), (1, 17), (9, 1), (1, 257), (5, 1), (3, 1)]:
        P = Pfa(N1, N2)
        N = P.N
        xa, xb = rng.standard_normal(N), rng.standard_normal(N)
        fa, fb = P.dct2(xa, xb)
        e1 = max(abs(fa - sf.dct(xa, norm="ortho")).max(), abs(fb - sf.dct(xb, norm="ortho")).max())
        ia, ib = P.dct3(xa, xb)
        e2 = max(abs(ia - sf.idct(xa, norm="ortho")).max(), abs(ib - sf.idct(xb, norm="ortho")).max())
        la, lb = 1.0 + rng.random(N), 1.0 + rng.random(N)
        ta, tb = P.tsolve(xa, xb, la, lb)
        e3 = max(abs(ta - sf.idct(sf.dct(xa, norm="ortho") / la, norm="ortho")).max(),
                 abs(tb - sf.idct(sf.dct(xb, norm="ortho") / lb, norm="ortho")).max())
        print(N1, N2, N, "fwd %.2e inv %.2e tsolve %.2e" % (e1, e2, e3))
