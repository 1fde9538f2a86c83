// O(n log n)-type DCT-II / DCT-III for the grid lengths the reference's multilevel driver actually runs:
// n = 2^k + 1 (solver_dotsocp2d.m:166-178; demo_dot2d.m:12-14: 129 x 129 x 33; mirt_dctn.m:100-141 and
// mirt_idctn.m:98-128 are FFT-based for ANY length).  Such an n is odd and factors into two coprime parts,
//      1025 = 25 . 41,  513 = 27 . 19,  129 = 3 . 43,  65 = 5 . 13,  33 = 3 . 11   (17, 9, 5, 3: one factor),
// so the length-n complex DFT behind Makhoul's DCT is a prime-factor (Good-Thomas) transform: the line, laid out as
// an N1 x N2 array through the index map p -> (p mod N1, p mod N2), needs N2-point DFTs along the rows, N1-point DFTs
// along the columns, NO twiddle factors in between, and comes out at k -> (k N2^-1 mod N1, k N1^-1 mod N2).  Fed
// through that second map it comes out through the first, which is what lets the fused t-axis solve run both of
// its transforms in place.
// A small DFT of odd length M is done densely but folded: with e_j = x_j + x_{M-j}, o_j = x_j - x_{M-j} it is
//      Y_k, Y_{M-k} = x_0 + sum_j cos(2 pi jk / M) e_j  -/+  i sum_j sin(2 pi jk / M) o_j,     j = 1 .. (M-1)/2,
// i.e. (M-1)/2 x (M-1)/2 REAL cos / sin matrices on the real and imaginary parts separately: one thread owns one
// (row, part), keeps its M folded inputs in registers, streams the matrix through the scalar cache (wave-uniform
// s_load: no LDS traffic, one v_fma_f64 per multiply-add with an SGPR operand) and writes the outputs in place.
// 1025-point line: 25 x 800 + 41 x 288 = 31.8k multiply-adds per real line instead of the 525k of the dense DCT
// product (k_dct_mfma_split), all of them on the vector ALU -- the fp64 matrix cores have the same peak rate as the
// vector fp64 pipe on this chip, so 16x fewer operations beat the contraction.  The passes run at 2.5 - 3.8 TB/s, bound by
// their own vector-ALU work (see pfa_launch_axis0_n), 6x faster than the dense product.
#include "device_utils.h"
#include "kernels.h"
#include "pfa.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace dotsocp {

typedef const double __attribute__((address_space(4))) *ctab_t;      // constant address space: uniform loads become s_load

__device__ __forceinline__ ctab_t as_ctab(const double *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (ctab_t)p;
#pragma clang diagnostic pop
}

constexpr int pfa_modinv(int a, int m) {
    if (m <= 1) return 0;
    a %= m;
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 0;
}

template <int N1_, int N2_>
struct PfaDims {
    static constexpr int N1 = N1_, N2 = N2_, N = N1_ * N2_, NH = (N1_ * N2_ - 1) / 2;
    static constexpr int I2 = pfa_modinv(N2_, N1_);     // N2^-1 mod N1
    static constexpr int I1 = pfa_modinv(N1_, N2_);     // N1^-1 mod N2
    // Position of element p inside the [N1][N2] image of a line: map A: (p mod N1) N2 + (p mod N2), map B:
    // ((p I2) mod N1) N2 + ((p I1) mod N2) -- a transform that reads through one writes through the other.  The kernels
    // look the positions up (PfaPlan::pos, built on the host together with Makhoul's reordering v[j] = x[2j],
    // v[n-1-j] = x[2j+1], mirt_dctn.m:71).
};

// In-place folded DFT of the M values p[0], p[ES], ..., p[(M-1) ES] (doubles; the own component of complex elements)
// with the other component one double beside each of them.  part = 0: own = real part, 1: own = imaginary part.
// tab: rows k = 1 .. H of [cos(2 pi jk/M), j = 1..H | sin(2 pi jk/M), j = 1..H].
template <int M, int ES>
__device__ __forceinline__ void pfa_small_dft(double *__restrict__ p, const int part, ctab_t tab) {
    if constexpr (M > 1) {
        constexpr int H = (M - 1) / 2;
        const double *q = part ? p - 1 : p + 1;
        double e[H], o[H];
        const double u0 = p[0];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const double p1 = p[j * ES], p2 = p[(M - j) * ES], q1 = q[j * ES], q2 = q[(M - j) * ES];
            e[j - 1] = p1 + p2;
            o[j - 1] = part ? q2 - q1 : q1 - q2;      // Im Y = b0 + sum c eb - sum s oa: the sign rides on o
        }
        {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < H; ++j) {
                if (j & 1) s1 += e[j]; else s0 += e[j];
            }
            p[0] = u0 + (s0 + s1);
        }
        if constexpr (H <= 4) {
#pragma unroll
            for (int k = 1; k <= H; ++k) {
                double ce = 0.0, so = 0.0;
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    ce = __builtin_fma(tab[(k - 1) * 2 * H + j], e[j], ce);
                    so = __builtin_fma(tab[(k - 1) * 2 * H + H + j], o[j], so);
                }
                const double c = u0 + ce;
                p[k * ES] = c + so;
                p[(M - k) * ES] = c - so;
            }
        } else {
#pragma unroll 2
            for (int k = 1; k <= H; ++k) {
                ctab_t row = tab + (k - 1) * 2 * H;
                double ce0 = 0.0, ce1 = 0.0, so0 = 0.0, so1 = 0.0;
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    if (j & 1) {
                        ce1 = __builtin_fma(row[j], e[j], ce1);
                        so1 = __builtin_fma(row[H + j], o[j], so1);
                    } else {
                        ce0 = __builtin_fma(row[j], e[j], ce0);
                        so0 = __builtin_fma(row[H + j], o[j], so0);
                    }
                }
                const double c = u0 + (ce0 + ce1), s = so0 + so1;
                p[k * ES] = c + s;
                p[(M - k) * ES] = c - s;
            }
        }
    }
}

// both stages on a tile of P line pairs ([P][N1][N2] complex values), ending with a workgroup barrier
template <class PF, int P, int T>
__device__ __forceinline__ void pfa_tile_dft(double2 *tile, int tid, ctab_t tab1, ctab_t tab2) {
    constexpr int N1 = PF::N1, N2 = PF::N2, N = PF::N;
    if constexpr (N2 > 1) {
        for (int item = tid; item < P * N1 * 2; item += T) {
            const int part = item & 1, row = item >> 1;
            pfa_small_dft<N2, 2>((double *)(tile + row * N2) + part, part, tab2);
        }
        __syncthreads();
    }
    if constexpr (N1 > 1) {
        for (int item = tid; item < P * N2 * 2; item += T) {
            const int part = item & 1, c = item >> 1;
            const int r = c / N2, k2 = c - r * N2;
            pfa_small_dft<N1, 2 * N2>((double *)(tile + r * N + k2) + part, part, tab1);
        }
        __syncthreads();
    }
}

// DCT post-processing of the pair (k, m = N - k) of one packed spectrum: (Xa[k], Xb[k], Xa[m], Xb[m])
// V_a = (V[k] + conj(V[m])) / 2, V_b = (V[k] - conj(V[m])) / (2i), X = real(ww .* V)   (mirt_dctn.m:130)
struct Pfa4 {
    double ak, bk, am, bm;
};
__device__ __forceinline__ Pfa4 pfa_post(double2 vk, double2 vm, double2 wk, double2 wm) {
    const double ar = 0.5 * (vk.x + vm.x), ai = 0.5 * (vk.y - vm.y);
    const double br = 0.5 * (vk.y + vm.y), bi = -0.5 * (vk.x - vm.x);
    Pfa4 r;
    r.ak = wk.x * ar - wk.y * ai;
    r.bk = wk.x * br - wk.y * bi;
    r.am = wm.x * ar + wm.y * ai;
    r.bm = wm.x * br + wm.y * bi;
    return r;
}

// inverse pre-processing of the pair (k, m): G[k] = (ww[k] X[k] + conj(ww[m]) X[m]) / 2 for both lines, packed
// Ga + i Gb   (mirt_idctn.m:109,119-120: fft(G) = real(fft(ww .* X)))
__device__ __forceinline__ void pfa_pre(const Pfa4 &x, double2 wk, double2 wm, double2 &gk, double2 &gm) {
    const double gar = 0.5 * (wk.x * x.ak + wm.x * x.am), gai = 0.5 * (wk.y * x.ak - wm.y * x.am);
    const double gbr = 0.5 * (wk.x * x.bk + wm.x * x.bm), gbi = 0.5 * (wk.y * x.bk - wm.y * x.bm);
    gk = make_double2(gar - gbi, gai + gbr);
    gm = make_double2(gar + gbi, gbr - gai);
}

// Strided axes.  Lines are addressed as (row, y): first element at row * rowStride + y, element k a further
// k * elStride on; the 2 P lines of a tile are consecutive in y (coalesced accesses), a tile never straddles rows.
//   x axis:  row = t, rowStride = pitch * nx, elStride = pitch;    t axis: row = x, rowStride = pitch, elStride = pitch * nx
// (pitch = allocated row length of the array, >= ny).
struct PfaGeom {
    i64 ny, nrows;
    i64 srow, sel, drow, del;
    int tilesPerRow;
    int xcd;
};
// MODE 2 (fused t-axis solve): lambda of line (row, y), mode k = cy[G % nyE] + cx[G / nyE] + ct[k], G = line0 + row * gRow + y
struct PfaSolve {
    double kscale;
    const double *cy, *cx, *ct;
    i64 nyE, line0, gRow;
};

template <class PF, int P, int T, int MODE /*0 forward, 1 inverse, 2 t-axis solve*/, bool VEC>
__global__ void __launch_bounds__(T, (PF::N > 1000 ? 4 : 1)) k_pfa_strided(const double *__restrict__ src, double *__restrict__ dst, PfaGeom g,
                                                    PfaSolve sa, const double2 *__restrict__ ww,
                                                    const double *__restrict__ t1, const double *__restrict__ t2,
                                                    const unsigned short *__restrict__ pos) {
    extern __shared__ double2 tile[];
    constexpr int N = PF::N, NH = PF::NH;
    // position tables (PfaPlan::pos): the two index maps composed with Makhoul's reordering, looked up instead of
    // recomputed (the modular arithmetic was a third of these kernels' vector instructions)
    const unsigned short *tA = pos, *tB = pos + N, *tAk = pos + 2 * N, *tBm = pos + 3 * N;
    constexpr int KS = T / P;                      // element step of the load / store loops
    static_assert(T % P == 0, "threads per pair");
    const ctab_t tab1 = as_ctab(t1), tab2 = as_ctab(t2);
    const int tid = threadIdx.x;
    const i64 tl = g.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (i64)blockIdx.x;
    const i64 row = tl / g.tilesPerRow;
    const i64 y0 = (tl - row * g.tilesPerRow) * (2 * P);
    const int r = tid % P, e0 = tid / P;
    const bool okA = y0 + 2 * r < g.ny, okB = y0 + 2 * r + 1 < g.ny;
    const double *sp = src + row * g.srow + y0 + 2 * r;
    double *dp = dst + row * g.drow + y0 + 2 * r;
    double2 *mine = tile + r * N;
    // VEC: every pair starts on an even element and -- when the row has an odd number of lines -- its pitch leaves room
    // for one more element, so the last pair of a row is fetched as 16 bytes too; what comes in from the pad is replaced by
    // zero, and only the valid half of that pair is stored (pad entries are zero and are never written: common.h)
    auto ld2 = [&](i64 off) -> double2 {
        if (VEC) {
            double2 v = okA ? *(const double2 *)(sp + off) : make_double2(0.0, 0.0);
            if (!okB) v.y = 0.0;
            return v;
        }
        return make_double2(okA ? sp[off] : 0.0, okB ? sp[off + 1] : 0.0);
    };
    auto st2 = [&](i64 off, double a, double b) {
        if (VEC) {
            if (okB) *(double2 *)(dp + off) = make_double2(a, b);
            else if (okA) dp[off] = a;
        } else {
            if (okA) dp[off] = a;
            if (okB) dp[off + 1] = b;
        }
    };
    // ---- load ----
    if (MODE != 1) {
        constexpr int NIT = (N + KS - 1) / KS;
        constexpr int NB = NIT < 8 ? NIT : 8;
        for (int b0 = 0; b0 < NIT; b0 += NB) {
            double2 v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = e0 + (b0 + u) * KS;
                v[u] = ld2((i64)(e < N ? e : 0) * g.sel);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = e0 + (b0 + u) * KS;
                if (e < N) mine[tA[e]] = v[u];
            }
        }
    } else {
        // spectrum in natural order; the pair (k, N - k) of both lines becomes G[k], G[N - k] on the way in
        constexpr int NIT = (NH + 1 + KS - 1) / KS;
        constexpr int NB = NIT < 4 ? NIT : 4;
        for (int b0 = 0; b0 < NIT; b0 += NB) {
            double2 vk[NB], vm[NB], wk[NB], wm[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int k = e0 + (b0 + u) * KS;
                const int kc = k <= NH ? k : 0, mc = kc ? N - kc : 0;
                vk[u] = ld2((i64)kc * g.sel);
                vm[u] = ld2((i64)mc * g.sel);
                wk[u] = ww[kc];
                wm[u] = ww[mc];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int k = e0 + (b0 + u) * KS;
                if (k > NH) continue;
                if (k == 0) {
                    mine[0] = make_double2(wk[u].x * vk[u].x, wk[u].x * vk[u].y);
                } else {
                    double2 gk, gm;
                    pfa_pre(Pfa4{vk[u].x, vk[u].y, vm[u].x, vm[u].y}, wk[u], wm[u], gk, gm);
                    mine[tAk[k]] = gk;
                    mine[tAk[N - k]] = gm;
                }
            }
        }
    }
    __syncthreads();
    pfa_tile_dft<PF, P, T>(tile, tid, tab1, tab2);
    if (MODE == 2) {
        // spectrum at map-B positions: post-processing, division by the eigenvalues, inverse pre-processing, in place
        const i64 Ga = sa.line0 + row * sa.gRow + y0 + 2 * r;
        const i64 Gac = okA ? Ga : sa.line0, Gbc = okB ? Ga + 1 : Gac;
        const double ea = sa.cy[Gac % sa.nyE] + sa.cx[Gac / sa.nyE];
        const double eb = sa.cy[Gbc % sa.nyE] + sa.cx[Gbc / sa.nyE];
        for (int k = e0; k <= NH; k += KS) {
            if (k == 0) {
                double la = ea + sa.ct[0], lb = eb + sa.ct[0];
                if (la == 0.0) la = 1.0;
                if (lb == 0.0) lb = 1.0;
                const double w0 = ww[0].x;
                const double2 v0 = mine[0];
                mine[0] = make_double2(w0 * ((w0 * v0.x) / (sa.kscale * la)), w0 * ((w0 * v0.y) / (sa.kscale * lb)));
                continue;
            }
            const int m = N - k;
            const int ik = tB[k], im = tB[m];
            const double2 wk = ww[k], wm = ww[m];
            Pfa4 x = pfa_post(mine[ik], mine[im], wk, wm);
            const double ctk = sa.ct[k], ctm = sa.ct[m];
            double lak = ea + ctk, lbk = eb + ctk, lam = ea + ctm, lbm = eb + ctm;
            if (lak == 0.0) lak = 1.0;
            if (lbk == 0.0) lbk = 1.0;
            if (lam == 0.0) lam = 1.0;
            if (lbm == 0.0) lbm = 1.0;
            x.ak = x.ak / (sa.kscale * lak);       // (a reciprocal seed + Newton steps instead of the four IEEE divisions:
            x.bk = x.bk / (sa.kscale * lbk);       //  measured, no change in the pass's time -- not kept)
            x.am = x.am / (sa.kscale * lam);
            x.bm = x.bm / (sa.kscale * lbm);
            double2 gk, gm;
            pfa_pre(x, wk, wm, gk, gm);
            mine[ik] = gk;
            mine[im] = gm;
        }
        __syncthreads();
        pfa_tile_dft<PF, P, T>(tile, tid, tab1, tab2);      // reads through map B, writes through map A
    }
    // ---- store ----
    if (MODE == 0) {
        for (int k = e0; k <= NH; k += KS) {
            if (k == 0) {
                const double w0 = ww[0].x;
                const double2 v0 = mine[0];
                st2(0, w0 * v0.x, w0 * v0.y);
                continue;
            }
            const int m = N - k;
            const Pfa4 x = pfa_post(mine[tB[k]], mine[tB[m]], ww[k], ww[m]);
            st2((i64)k * g.del, x.ak, x.bk);
            st2((i64)m * g.del, x.am, x.bm);
        }
    } else {
        for (int e = e0; e < N; e += KS) {
            const double2 v = mine[MODE == 1 ? tBm[e] : tA[e]];
            st2((i64)e * g.del, v.x, v.y);
        }
    }
}

// Axis 0: lines contiguous in memory (line L starts at L * lineStride), a tile = 2 P consecutive lines.  The T threads
// form P groups of GS = T / P; group r owns pair r (lines 2 r, 2 r + 1) and its lanes walk the line: consecutive lanes
// touch consecutive elements (coalesced), no index is ever divided, and both lines of a pair travel together so that
// the LDS image is written in whole 16-byte values.
template <class PF, int P, int T, bool INV>
__global__ void __launch_bounds__(T, (PF::N > 1000 ? 4 : 1)) k_pfa_axis0(const double *__restrict__ src, double *__restrict__ dst, i64 nLines,
                                                  i64 sline, i64 dline, const double2 *__restrict__ ww,
                                                  const double *__restrict__ t1, const double *__restrict__ t2,
                                                  const unsigned short *__restrict__ pos) {
    extern __shared__ double2 tile[];
    constexpr int N = PF::N, NH = PF::NH;
    constexpr int GS = T / P;
    static_assert(T % P == 0 && (GS & (GS - 1)) == 0, "threads per pair: a power of two");
    const unsigned short *tA = pos, *tB = pos + N, *tAk = pos + 2 * N, *tBm = pos + 3 * N;
    const ctab_t tab1 = as_ctab(t1), tab2 = as_ctab(t2);
    const int tid = threadIdx.x;
    const int r = tid / GS, j = tid % GS;
    const i64 La = (i64)blockIdx.x * (2 * P) + 2 * r;
    const bool okA = La < nLines, okB = La + 1 < nLines;
    const double *sa = src + (okA ? La : 0) * sline, *sb = src + (okB ? La + 1 : 0) * sline;
    double *da = dst + La * dline, *db = da + dline;
    double2 *mine = tile + r * N;
    // ---- load ----
    if (!INV) {
        constexpr int NIT = (N + GS - 1) / GS;
        constexpr int NB = NIT < 9 ? NIT : 9;
        for (int b0 = 0; b0 < NIT; b0 += NB) {
            double2 v[NB];
            int p[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = j + (b0 + u) * GS;
                const int ec = e < N ? e : 0;
                v[u] = make_double2(okA ? sa[ec] : 0.0, okB ? sb[ec] : 0.0);
                p[u] = tA[ec];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = j + (b0 + u) * GS;
                if (e < N && (b0 + u) < NIT) mine[p[u]] = v[u];
            }
        }
    } else {
        constexpr int NIT = (NH + 1 + GS - 1) / GS;
        constexpr int NB = NIT < 5 ? NIT : 5;
        for (int b0 = 0; b0 < NIT; b0 += NB) {
            Pfa4 x[NB];
            double2 wk[NB], wm[NB];
            int pk[NB], pm[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int k = j + (b0 + u) * GS;
                const int kc = k <= NH ? k : 0, mc = kc ? N - kc : 0;
                x[u].ak = okA ? sa[kc] : 0.0;
                x[u].am = okA ? sa[mc] : 0.0;
                x[u].bk = okB ? sb[kc] : 0.0;
                x[u].bm = okB ? sb[mc] : 0.0;
                wk[u] = ww[kc];
                wm[u] = ww[mc];
                pk[u] = tAk[kc];
                pm[u] = tAk[mc];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int k = j + (b0 + u) * GS;
                if (k > NH || (b0 + u) >= NIT) continue;
                if (k == 0) {
                    mine[0] = make_double2(wk[u].x * x[u].ak, wk[u].x * x[u].bk);
                } else {
                    double2 gk, gm;
                    pfa_pre(x[u], wk[u], wm[u], gk, gm);
                    mine[pk[u]] = gk;
                    mine[pm[u]] = gm;
                }
            }
        }
    }
    __syncthreads();
    pfa_tile_dft<PF, P, T>(tile, tid, tab1, tab2);
    // ---- store ----
    if (!INV) {
        for (int k = j; k <= NH; k += GS) {
            if (k == 0) {
                const double w0 = ww[0].x;
                const double2 v0 = mine[0];
                if (okA) da[0] = w0 * v0.x;
                if (okB) db[0] = w0 * v0.y;
                continue;
            }
            const int m = N - k;
            const Pfa4 x = pfa_post(mine[tB[k]], mine[tB[m]], ww[k], ww[m]);
            if (okA) { da[k] = x.ak; da[m] = x.am; }
            if (okB) { db[k] = x.bk; db[m] = x.bm; }
        }
    } else {
        for (int e = j; e < N; e += GS) {
            const double2 v = mine[tBm[e]];
            if (okA) da[e] = v.x;
            if (okB) db[e] = v.y;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct PfaPlan {
    int n, n1, n2;
    double2 *ww;          // [n]  2 exp(-i pi k / 2n) / sqrt(2n), ww[0] /= sqrt(2)   (mirt_dctn.m:69-70)
    double *tab1, *tab2;  // folded cos / sin matrices of the N1- and N2-point DFTs
    unsigned short *pos;  // [4][n] positions inside the [N1][N2] image: A(makhoul(e)), B(k), A(k), B(makhoul(e))
};

static bool pfa_factors(i64 n, int &n1, int &n2) {
    switch (n) {
        case 1025: n1 = 25; n2 = 41; return true;
        case 513: n1 = 27; n2 = 19; return true;
        case 129: n1 = 3; n2 = 43; return true;
        case 65: n1 = 5; n2 = 13; return true;
        case 33: n1 = 3; n2 = 11; return true;
        case 17: n1 = 1; n2 = 17; return true;
        case 9: n1 = 9; n2 = 1; return true;
        case 5: n1 = 5; n2 = 1; return true;
        case 3: n1 = 3; n2 = 1; return true;
        default: return false;
    }
}

bool pfa_supported(i64 n) {
    int a, b;
    return pfa_factors(n, a, b);
}

static int pfa_make_tab(double **out, int M) {
    *out = nullptr;
    if (M <= 1) return 0;
    const int H = (M - 1) / 2;
    const long double PI = 3.141592653589793238462643383279502884L;
    std::vector<double> t((size_t)H * 2 * H);
    for (int k = 1; k <= H; ++k)
        for (int j = 1; j <= H; ++j) {
            const long double a = 2.0L * PI * (long double)((j * k) % M) / (long double)M;
            t[(size_t)(k - 1) * 2 * H + (j - 1)] = (double)cosl(a);
            t[(size_t)(k - 1) * 2 * H + H + (j - 1)] = (double)sinl(a);
        }
    // a few spare doubles behind the table: the scalar loads of a row are issued in blocks of up to 16 dwords
    if (hipMalloc(out, sizeof(double) * (t.size() + 16)) != hipSuccess) return DOTSOCP_EHIP;
    if (hipMemset(*out, 0, sizeof(double) * (t.size() + 16)) != hipSuccess) return DOTSOCP_EHIP;
    if (hipMemcpy(*out, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice) != hipSuccess) return DOTSOCP_EHIP;
    return 0;
}

PfaPlan *pfa_plan_create(i64 n) {
    int n1, n2;
    if (!pfa_factors(n, n1, n2)) return nullptr;
    PfaPlan *p = new PfaPlan();
    p->n = (int)n; p->n1 = n1; p->n2 = n2;
    p->ww = nullptr; p->tab1 = p->tab2 = nullptr; p->pos = nullptr;
    const long double PI = 3.141592653589793238462643383279502884L;
    std::vector<double2> ww((size_t)n);
    for (i64 k = 0; k < n; ++k) {
        long double a = -PI * (long double)k / (2.0L * (long double)n);
        long double sc = 2.0L / sqrtl(2.0L * (long double)n);
        if (k == 0) sc /= sqrtl(2.0L);
        ww[k] = make_double2((double)(sc * cosl(a)), (double)(sc * sinl(a)));
    }
    // the index maps of pfa.hip's header: A: p -> (p mod N1, p mod N2), B: p -> (p N2^-1 mod N1, p N1^-1 mod N2)
    std::vector<unsigned short> pos((size_t)4 * n);
    {
        auto inv = [](int a, int m) { a %= (m > 0 ? m : 1); for (int x = 1; x < m; ++x) if ((a * x) % m == 1) return x; return 0; };
        const int i2 = inv(n2, n1), i1 = inv(n1, n2);
        auto A = [&](int q) { return (q % n1) * n2 + (q % n2); };
        auto B = [&](int q) { return ((q * i2) % n1) * n2 + ((q * i1) % n2); };
        auto mk = [&](int k) { return (k & 1) ? ((int)n - 1 - (k >> 1)) : (k >> 1); };
        for (int e = 0; e < (int)n; ++e) {
            pos[e] = (unsigned short)A(mk(e));
            pos[(size_t)n + e] = (unsigned short)B(e);
            pos[(size_t)2 * n + e] = (unsigned short)A(e);
            pos[(size_t)3 * n + e] = (unsigned short)B(mk(e));
        }
    }
    if (hipMalloc(&p->ww, sizeof(double2) * n) != hipSuccess ||
        hipMemcpy(p->ww, ww.data(), sizeof(double2) * n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&p->pos, sizeof(unsigned short) * pos.size()) != hipSuccess ||
        hipMemcpy(p->pos, pos.data(), sizeof(unsigned short) * pos.size(), hipMemcpyHostToDevice) != hipSuccess ||
        pfa_make_tab(&p->tab1, n1) != 0 || pfa_make_tab(&p->tab2, n2) != 0) {
        pfa_plan_destroy(p);
        return nullptr;
    }
    return p;
}

void pfa_plan_destroy(PfaPlan *p) {
    if (!p) return;
    if (p->ww) (void)hipFree(p->ww);
    if (p->tab1) (void)hipFree(p->tab1);
    if (p->tab2) (void)hipFree(p->tab2);
    if (p->pos) (void)hipFree(p->pos);
    delete p;
}

// tile shape per length: P pairs of lines per workgroup of T threads (LDS = P * n * 16 bytes)
#ifndef PFA_P0_1025
#define PFA_P0_1025 4
#define PFA_T0_1025 512
#define PFA_P0_513 8
#define PFA_T0_513 512
#endif
template <int N> struct PfaShape;
// (P0, T0: the same for the axis-0 kernels, whose lines are contiguous in memory whatever the tile holds)
template <> struct PfaShape<1025> { typedef PfaDims<25, 41> D; static constexpr int P = 4, T = 512, P0 = PFA_P0_1025, T0 = PFA_T0_1025; };
template <> struct PfaShape<513> { typedef PfaDims<27, 19> D; static constexpr int P = 8, T = 512, P0 = PFA_P0_513, T0 = PFA_T0_513; };
template <> struct PfaShape<129> { typedef PfaDims<3, 43> D; static constexpr int P = 16, T = 256, P0 = 16, T0 = 256; };
template <> struct PfaShape<65> { typedef PfaDims<5, 13> D; static constexpr int P = 32, T = 256, P0 = 32, T0 = 256; };
template <> struct PfaShape<33> { typedef PfaDims<3, 11> D; static constexpr int P = 32, T = 256, P0 = 32, T0 = 256; };
template <> struct PfaShape<17> { typedef PfaDims<1, 17> D; static constexpr int P = 32, T = 256, P0 = 32, T0 = 256; };
template <> struct PfaShape<9> { typedef PfaDims<9, 1> D; static constexpr int P = 32, T = 256, P0 = 32, T0 = 256; };
template <> struct PfaShape<5> { typedef PfaDims<5, 1> D; static constexpr int P = 32, T = 256, P0 = 32, T0 = 256; };
template <> struct PfaShape<3> { typedef PfaDims<3, 1> D; static constexpr int P = 32, T = 256, P0 = 32, T0 = 256; };

#define PFA_FOR_LENGTHS(X) X(1025) X(513) X(129) X(65) X(33) X(17) X(9) X(5) X(3)

// the dynamic-LDS limit is a (function, device) attribute: raised once per device under a lock (contexts on different
// host threads may reach this at the same time)
static std::mutex pfa_attr_mutex;
static unsigned long long pfa_attr_done = 0;

template <int N>
static void pfa_raise_lds() {
    typedef PfaShape<N> S;
    typedef typename S::D D;
    constexpr int P = S::P, T = S::T;
    const int lim = 160 * 1024;
#define PFA_RAISE(K) (void)hipFuncSetAttribute((const void *)(K), hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    PFA_RAISE((k_pfa_strided<D, P, T, 0, true>)); PFA_RAISE((k_pfa_strided<D, P, T, 0, false>));
    PFA_RAISE((k_pfa_strided<D, P, T, 1, true>)); PFA_RAISE((k_pfa_strided<D, P, T, 1, false>));
    PFA_RAISE((k_pfa_strided<D, P, T, 2, true>)); PFA_RAISE((k_pfa_strided<D, P, T, 2, false>));
    PFA_RAISE((k_pfa_axis0<D, S::P0, S::T0, false>)); PFA_RAISE((k_pfa_axis0<D, S::P0, S::T0, true>));
#undef PFA_RAISE
}

static void pfa_prepare_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lock(pfa_attr_mutex);
    if (pfa_attr_done & (1ull << dev)) return;
    pfa_raise_lds<1025>();
    pfa_raise_lds<513>();
    pfa_attr_done |= 1ull << dev;      // only after the attributes are in place
}

template <int N>
static void pfa_launch_strided_n(const PfaPlan *p, const double *src, double *dst, const PfaGeom &g, const PfaSolve &sa,
                                 int mode, bool vec, unsigned blocks, hipStream_t st) {
    typedef PfaShape<N> S;
    typedef typename S::D D;
    constexpr int P = S::P, T = S::T;
    const size_t lds = (size_t)P * N * sizeof(double2);
#define PFA_GO(M, V) DS_KLAUNCH((k_pfa_strided<D, P, T, M, V>), dim3(blocks), dim3(T), lds, st, src, dst, g, sa, p->ww, p->tab1, p->tab2, p->pos)
    if (mode == 0) { if (vec) PFA_GO(0, true); else PFA_GO(0, false); }
    else if (mode == 1) { if (vec) PFA_GO(1, true); else PFA_GO(1, false); }
    else { if (vec) PFA_GO(2, true); else PFA_GO(2, false); }
#undef PFA_GO
}

template <class D, int P, int T>
static void pfa_launch_axis0_pt(const PfaPlan *p, const double *src, double *dst, i64 nLines, i64 sline, i64 dline,
                                int inverse, hipStream_t st) {
    const size_t lds = (size_t)P * D::N * sizeof(double2);
    const unsigned blocks = (unsigned)((nLines + 2 * P - 1) / (2 * P));
    if (inverse)
        DS_KLAUNCH((k_pfa_axis0<D, P, T, true>), dim3(blocks), dim3(T), lds, st, src, dst, nLines, sline, dline, p->ww, p->tab1, p->tab2, p->pos);
    else
        DS_KLAUNCH((k_pfa_axis0<D, P, T, false>), dim3(blocks), dim3(T), lds, st, src, dst, nLines, sline, dline, p->ww, p->tab1, p->tab2, p->pos);
}

template <int N>
static void pfa_launch_axis0_n(const PfaPlan *p, const double *src, double *dst, i64 nLines, i64 sline, i64 dline,
                               int inverse, hipStream_t st) {
    typedef PfaShape<N> S;
    typedef typename S::D D;
    // Measured and dropped (1025 x 1025 x 129 / 513 x 513 x 129, Poisson solve): half the tile = twice the workgroups per
    // CU along axis 0: 3.26 vs 3.23 ms / 0.796 vs 0.805 ms; a persistent LDS-DMA pipeline like k_dct_axis0_pipe's (one
    // workgroup per CU, two 72 KB buffers, counted vmcnt waits; correct, 66 / 66 operator tests): 3.43 vs 3.22 ms / 0.868
    // vs 0.807 ms -- SLOWER.  These kernels are bound by their own vector-ALU work (folded DFT stages at 78 % / 64 % lane
    // use, index arithmetic of the two maps), about 4.4 us per 65.6 KB tile against 1.8 us of multiply-adds at full lanes,
    // not by bytes in flight: eight waves per CU hide less of it than sixteen.
    pfa_launch_axis0_pt<D, S::P0, S::T0>(p, src, dst, nLines, sline, dline, inverse, st);
}

template <int N>
static int pfa_pairs() { return PfaShape<N>::P; }

// mode 0 / 1: DCT-II / DCT-III along a strided axis; mode 2: t-axis solve (sa set).  nyLines lines per row.
int pfa_launch_strided(const PfaPlan *p, const double *src, double *dst, i64 nyLines, i64 nrows, i64 srow, i64 sel,
                       i64 drow, i64 del, int mode, const PfaSolveArgs *sargs, hipStream_t st) {
    if (nyLines <= 0 || nrows <= 0) return 0;
    pfa_prepare_device();
    int P = 0;
    switch (p->n) {
#define X(NV) case NV: P = pfa_pairs<NV>(); break;
        PFA_FOR_LENGTHS(X)
#undef X
        default: set_error("no prime-factor DCT for length %d", p->n); return DOTSOCP_EINVAL;
    }
    PfaGeom g;
    g.ny = nyLines; g.nrows = nrows; g.srow = srow; g.sel = sel; g.drow = drow; g.del = del;
    g.tilesPerRow = (int)((nyLines + 2 * P - 1) / (2 * P));
    const i64 tiles = (i64)g.tilesPerRow * nrows;
    if (tiles >= (1ll << 31)) { set_error("too many DCT tiles"); return DOTSOCP_EINVAL; }
    g.xcd = tiles >= 64 ? 1 : 0;
    PfaSolve sa{};
    if (mode == 2) {
        sa.kscale = sargs->kscale; sa.cy = sargs->cy; sa.cx = sargs->cx; sa.ct = sargs->ct;
        sa.nyE = sargs->nyE; sa.line0 = sargs->line0; sa.gRow = sargs->gRow;
    }
    // one 16-byte access carries both lines of a pair when every pair starts on an even element; a row with an odd
    // number of lines needs one pad element behind its last line (pitched rows: common.h)
    const bool roomy = (nyLines % 2 == 0) ||
                       ((nrows == 1 || (srow > nyLines && drow > nyLines)) && sel > nyLines && del > nyLines);
    const bool vec = roomy && (srow % 2 == 0) && (sel % 2 == 0) && (drow % 2 == 0) && (del % 2 == 0) &&
                     (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    // (the fused t pass of length 129: 32 pairs x 512 threads and 8 pairs x 128 threads measured against the 16 x 256 of
    // PfaShape<129>: Poisson solve 3.14 / 3.10 vs 3.04 ms at 1025 x 1025 x 129 -- not better)
    switch (p->n) {
#define X(NV) case NV: pfa_launch_strided_n<NV>(p, src, dst, g, sa, mode, vec, (unsigned)tiles, st); break;
        PFA_FOR_LENGTHS(X)
#undef X
    }
    DS_HIP(hipGetLastError());
    return 0;
}

int pfa_launch_axis0(const PfaPlan *p, const double *src, double *dst, i64 nLines, i64 sline, i64 dline, int inverse,
                     hipStream_t st) {
    if (nLines <= 0) return 0;
    pfa_prepare_device();
    switch (p->n) {
#define X(NV) case NV: pfa_launch_axis0_n<NV>(p, src, dst, nLines, sline, dline, inverse, st); break;
        PFA_FOR_LENGTHS(X)
#undef X
        default: set_error("no prime-factor DCT for length %d", p->n); return DOTSOCP_EINVAL;
    }
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
