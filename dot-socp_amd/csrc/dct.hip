// Batched orthonormal DCT-II / DCT-III along one axis of an [n0][n1][n2] fp64 array (n0
// fastest) -- the transforms behind the Neumann-Poisson solve phi = idctn(dctn(rhs) ./ kernel)
// (socp/dot2d/utils/oper_poisson3dim.m:4; mirt_dctn.m:64-141, mirt_idctn.m:59-128).
//
// Power-of-two lengths: Makhoul's reordering + one complex FFT per PAIR of real lines (line a
// in the real part, line b in the imaginary part), entirely in LDS: one HBM read and one HBM
// write per element and axis.  A workgroup stages TL lines; for the strided axes (x, t) the TL
// lines are consecutive in y so that global accesses stay coalesced.
// Other lengths (the 2^k+1 grids of the multilevel driver): dense DCT matrix applied from an
// LDS-staged tile (exact, O(n^2) per line; fallback path).
#include "kernels.h"

#include <cmath>
#include <vector>

namespace dotsocp {

struct DctPlan {
    i64 n;
    int log2n;      // -1 when n is not a power of two
    double2 *tw;    // [n/2]  exp(-2 pi i k / n)
    double2 *ww;    // [n]    2 exp(-i pi k / 2n) / sqrt(2n), ww[0] /= sqrt(2)   (mirt_dctn.m:69-70)
    double *Cfwd;   // dense: Cfwd[j*n + k] = C[k][j]   (forward,  out_k = sum_j C[k][j] in_j)
    double *Cinv;   // dense: Cinv[j*n + k] = C[j][k]   (inverse)
};

DctPlan *dct_plan_create(i64 n) {
    DctPlan *p = new DctPlan();
    p->n = n;
    p->log2n = -1;
    p->tw = nullptr;
    p->ww = nullptr;
    p->Cfwd = p->Cinv = nullptr;
    if (n <= 1) return p;
    const long double PI = 3.141592653589793238462643383279502884L;
    if ((n & (n - 1)) == 0) {
        int lg = 0;
        while (((i64)1 << lg) < n) ++lg;
        p->log2n = lg;
        std::vector<double2> tw(n / 2), ww(n);
        for (i64 k = 0; k < n / 2; ++k) {
            long double a = -2.0L * PI * (long double)k / (long double)n;
            tw[k] = make_double2((double)cosl(a), (double)sinl(a));
        }
        for (i64 k = 0; k < n; ++k) {
            long double a = -PI * (long double)k / (2.0L * (long double)n);
            long double sc = 2.0L / sqrtl(2.0L * (long double)n);
            if (k == 0) sc /= sqrtl(2.0L);
            ww[k] = make_double2((double)(sc * cosl(a)), (double)(sc * sinl(a)));
        }
        if (hipMalloc(&p->tw, sizeof(double2) * (n / 2)) != hipSuccess ||
            hipMalloc(&p->ww, sizeof(double2) * n) != hipSuccess) {
            dct_plan_destroy(p);
            return nullptr;
        }
        (void)hipMemcpy(p->tw, tw.data(), sizeof(double2) * (n / 2), hipMemcpyHostToDevice);
        (void)hipMemcpy(p->ww, ww.data(), sizeof(double2) * n, hipMemcpyHostToDevice);
    } else {
        std::vector<double> cf((size_t)n * n), ci((size_t)n * n);
        for (i64 k = 0; k < n; ++k) {
            long double sc = sqrtl(2.0L / (long double)n);
            if (k == 0) sc /= sqrtl(2.0L);
            for (i64 j = 0; j < n; ++j) {
                // reduce the argument exactly: cos(pi * m / (2n)) with m = (2j+1) k mod 4n
                i64 m = ((2 * j + 1) * k) % (4 * n);
                double v = (double)(sc * cosl(PI * (long double)m / (2.0L * (long double)n)));
                cf[(size_t)j * n + k] = v;   // C[k][j] stored with k contiguous
                ci[(size_t)k * n + j] = v;   // C[k][j] stored with j contiguous: inverse out_j = sum_k C[k][j] X_k
            }
        }
        if (hipMalloc(&p->Cfwd, sizeof(double) * n * n) != hipSuccess ||
            hipMalloc(&p->Cinv, sizeof(double) * n * n) != hipSuccess) {
            dct_plan_destroy(p);
            return nullptr;
        }
        (void)hipMemcpy(p->Cfwd, cf.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
        (void)hipMemcpy(p->Cinv, ci.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    }
    return p;
}

void dct_plan_destroy(DctPlan *p) {
    if (!p) return;
    if (p->tw) (void)hipFree(p->tw);
    if (p->ww) (void)hipFree(p->ww);
    if (p->Cfwd) (void)hipFree(p->Cfwd);
    if (p->Cinv) (void)hipFree(p->Cinv);
    delete p;
}

// Line addressing shared by all axes: line L, element k lives at
//   (L % nin) + (L / nin) * outerStride + k * nin
// axis 0: nin = 1, outerStride = n;  axis 1: nin = n0, outerStride = n0*n1;  axis 2: nin = n0*n1.
struct LineMap {
    i64 nin, outerStride, nLines;
    __device__ __forceinline__ i64 addr(i64 L, i64 k) const { return (L % nin) + (L / nin) * outerStride + k * nin; }
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ unsigned bitrev(unsigned k, int lg) { return __brev(k) >> (32 - lg); }

#define DCT_THREADS 256
#define DCT_PAD 1   // complex elements of padding per LDS row

// In-LDS radix-2 decimation-in-frequency FFT of `npairs` rows of length n (natural order in,
// bit-reversed order out).
__device__ __forceinline__ void fft_dif_lds(double2 *buf, int npairs, int n, int lg, int rowStride,
                                            const double2 *__restrict__ tw) {
    const int halfn = n >> 1;
    const int total = npairs * halfn;
    for (int s = lg - 1; s >= 0; --s) {
        const int half = 1 << s;
        const int twStride = halfn >> s;   // n / (2*half)
        for (int b = threadIdx.x; b < total; b += DCT_THREADS) {
            const int row = b / halfn;
            const int ii = b - row * halfn;
            const int j = ii & (half - 1);
            const int i = ((ii - j) << 1) + j;
            double2 *r = buf + row * rowStride;
            const double2 a = r[i], bb = r[i + half];
            const double2 w = tw[j * twStride];
            r[i] = make_double2(a.x + bb.x, a.y + bb.y);
            r[i + half] = cmul(make_double2(a.x - bb.x, a.y - bb.y), w);
        }
        __syncthreads();
    }
}

// LINE_FAST: consecutive threads walk consecutive lines (strided axes); otherwise consecutive k.
template <bool INVERSE, bool LINE_FAST>
__global__ void __launch_bounds__(DCT_THREADS) k_dct_pow2(const double *__restrict__ src, double *__restrict__ dst,
                                                           LineMap map, int n, int lg, int TL,
                                                           const double2 *__restrict__ tw,
                                                           const double2 *__restrict__ ww) {
    extern __shared__ double2 buf[];
    const int rowStride = n + DCT_PAD;
    const int npairs = TL >> 1;
    const i64 L0 = (i64)blockIdx.x * TL;
    const int total = TL * n;
    // ---- load ----
    for (int e = threadIdx.x; e < total; e += DCT_THREADS) {
        int l, k;
        if (LINE_FAST) { l = e % TL; k = e / TL; } else { k = e % n; l = e / n; }
        const i64 L = L0 + l;
        const double v = (L < map.nLines) ? src[map.addr(L, k)] : 0.0;
        // forward: Makhoul reordering v[j] = x[2j], v[n-1-j] = x[2j+1] (mirt_dctn.m:71);
        // inverse: natural order first, combined below
        const int pos = INVERSE ? k : ((k & 1) ? (n - 1 - (k >> 1)) : (k >> 1));
        ((double *)&buf[(l >> 1) * rowStride + pos])[l & 1] = v;
    }
    __syncthreads();
    if (INVERSE) {
        // G[k] = (ww[k] X[k] + conj(ww[n-k]) X[n-k]) / 2, so that fft(G) = real(fft(ww .* X))
        // (mirt_idctn.m:109,119-120); rows hold Xa + i Xb elementwise.
        const int halfn = n >> 1;
        for (int b = threadIdx.x; b < npairs * (halfn + 1); b += DCT_THREADS) {
            const int row = b / (halfn + 1);
            const int k = b - row * (halfn + 1);
            double2 *r = buf + row * rowStride;
            if (k == 0) {
                const double w0 = ww[0].x;
                r[0] = make_double2(w0 * r[0].x, w0 * r[0].y);
            } else {
                const int m = n - k;
                const double2 xk = r[k], xm = r[m];          // (Xa[k], Xb[k]), (Xa[m], Xb[m])
                const double2 wk = ww[k], wm = ww[m];
                // Ga[k] = (wk*Xa[k] + conj(wm)*Xa[m])/2 ; Gb likewise ; G = Ga + i Gb
                const double gar = 0.5 * (wk.x * xk.x + wm.x * xm.x), gai = 0.5 * (wk.y * xk.x - wm.y * xm.x);
                const double gbr = 0.5 * (wk.x * xk.y + wm.x * xm.y), gbi = 0.5 * (wk.y * xk.y - wm.y * xm.y);
                r[k] = make_double2(gar - gbi, gai + gbr);
                if (m != k) {
                    const double har = 0.5 * (wm.x * xm.x + wk.x * xk.x), hai = 0.5 * (wm.y * xm.x - wk.y * xk.x);
                    const double hbr = 0.5 * (wm.x * xm.y + wk.x * xk.y), hbi = 0.5 * (wm.y * xm.y - wk.y * xk.y);
                    r[m] = make_double2(har - hbi, hai + hbr);
                }
            }
        }
        __syncthreads();
    }
    fft_dif_lds(buf, npairs, n, lg, rowStride, tw);
    // ---- store ----
    for (int e = threadIdx.x; e < total; e += DCT_THREADS) {
        int l, k;
        if (LINE_FAST) { l = e % TL; k = e / TL; } else { k = e % n; l = e / n; }
        const i64 L = L0 + l;
        if (L >= map.nLines) continue;
        const double2 *r = buf + (l >> 1) * rowStride;
        double out;
        if (!INVERSE) {
            // X[k] = real(ww[k] * V[k]) per line; V_a = (V[k] + conj(V[n-k]))/2, V_b = (V[k] - conj(V[n-k]))/(2i)
            const double2 vk = r[bitrev((unsigned)k, lg)];
            const double2 vm = r[bitrev((unsigned)((n - k) & (n - 1)), lg)];
            const double2 w = ww[k];
            double vr, vi;
            if ((l & 1) == 0) { vr = 0.5 * (vk.x + vm.x); vi = 0.5 * (vk.y - vm.y); }
            else              { vr = 0.5 * (vk.y + vm.y); vi = -0.5 * (vk.x - vm.x); }
            out = w.x * vr - w.y * vi;
        } else {
            // x[2j] = v[j], x[2j+1] = v[n-1-j]   (mirt_idctn.m:71-73,120)
            const int srcp = (k & 1) ? (n - 1 - (k >> 1)) : (k >> 1);
            const double2 v = r[bitrev((unsigned)srcp, lg)];
            out = (l & 1) ? v.y : v.x;
        }
        dst[map.addr(L, k)] = out;
    }
}

// Dense fallback: out_k = sum_j M[j*n + k] in_j from an LDS-staged tile of TL lines.
template <bool LINE_FAST>
__global__ void __launch_bounds__(DCT_THREADS) k_dct_dense(const double *__restrict__ src, double *__restrict__ dst,
                                                            LineMap map, int n, int TL,
                                                            const double *__restrict__ M) {
    extern __shared__ double2 buf[];
    double *tile = (double *)buf;   // [TL][n]
    const i64 L0 = (i64)blockIdx.x * TL;
    const int total = TL * n;
    for (int e = threadIdx.x; e < total; e += DCT_THREADS) {
        int l, k;
        if (LINE_FAST) { l = e % TL; k = e / TL; } else { k = e % n; l = e / n; }
        const i64 L = L0 + l;
        tile[l * n + k] = (L < map.nLines) ? src[map.addr(L, k)] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += DCT_THREADS) {
        int l, k;
        if (LINE_FAST) { l = e % TL; k = e / TL; } else { k = e % n; l = e / n; }
        const i64 L = L0 + l;
        if (L >= map.nLines) continue;
        const double *in = tile + l * n;
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += M[(i64)j * n + k] * in[j];
        dst[map.addr(L, k)] = acc;
    }
}

__global__ void __launch_bounds__(256) k_copy(const double *__restrict__ src, double *__restrict__ dst, i64 n) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) dst[i] = src[i];
}

int launch_dct_axis(const DctPlan *p, const double *src, double *dst, i64 n0, i64 n1, i64 n2, int axis, int inverse,
                    hipStream_t st) {
    const i64 dims[3] = {n0, n1, n2};
    const i64 n = dims[axis];
    const i64 total = n0 * n1 * n2;
    if (total <= 0) return 0;
    if (n != p->n) {
        set_error("dct plan length mismatch (%lld vs %lld)", (long long)n, (long long)p->n);
        return DOTSOCP_EINVAL;
    }
    if (n == 1) {
        if (src != dst)
            hipLaunchKernelGGL(k_copy, dim3(launch_blocks(total, 256, 1 << 14)), dim3(256), 0, st, src, dst, total);
        DS_HIP(hipGetLastError());
        return 0;
    }
    LineMap map;
    if (axis == 0) { map.nin = 1; map.outerStride = n; }
    else if (axis == 1) { map.nin = n0; map.outerStride = n0 * n1; }
    else { map.nin = n0 * n1; map.outerStride = 0; }
    map.nLines = total / n;
    const bool lineFast = (axis != 0);
    if (p->log2n > 0) {
        // LDS budget ~64 KiB per workgroup: TL lines of n doubles (as TL/2 complex rows)
        int TL = (int)(65536 / (n * 8));
        if (TL > 16) TL = 16;
        if (TL < 2) TL = 2;
        TL &= ~1;
        if ((i64)TL > ((map.nLines + 1) & ~(i64)1)) TL = (int)((map.nLines + 1) & ~(i64)1);
        const size_t lds = (size_t)(TL / 2) * (n + DCT_PAD) * sizeof(double2);
        const unsigned blocks = (unsigned)((map.nLines + TL - 1) / TL);
        const int lg = p->log2n;
#define LAUNCH_POW2(INV, LF)                                                                                       \
    hipLaunchKernelGGL((k_dct_pow2<INV, LF>), dim3(blocks), dim3(DCT_THREADS), lds, st, src, dst, map, (int)n, lg, \
                       TL, p->tw, p->ww)
        if (inverse) { if (lineFast) LAUNCH_POW2(true, true); else LAUNCH_POW2(true, false); }
        else         { if (lineFast) LAUNCH_POW2(false, true); else LAUNCH_POW2(false, false); }
#undef LAUNCH_POW2
    } else {
        if (src == dst) {
            set_error("dense DCT path needs distinct src/dst");
            return DOTSOCP_EINVAL;
        }
        int TL = (int)(65536 / (n * 8));
        if (TL > 16) TL = 16;
        if (TL < 1) TL = 1;
        const size_t lds = (size_t)TL * n * sizeof(double);
        const unsigned blocks = (unsigned)((map.nLines + TL - 1) / TL);
        const double *M = inverse ? p->Cinv : p->Cfwd;
        if (lineFast)
            hipLaunchKernelGGL((k_dct_dense<true>), dim3(blocks), dim3(DCT_THREADS), lds, st, src, dst, map, (int)n, TL, M);
        else
            hipLaunchKernelGGL((k_dct_dense<false>), dim3(blocks), dim3(DCT_THREADS), lds, st, src, dst, map, (int)n, TL, M);
    }
    DS_HIP(hipGetLastError());
    return 0;
}

// data ./= kscale * ((CY[ky] + CX[kx]) + CT[kt]) with the zero eigenvalue replaced by 1
// (initialize_FFTkernel.m:6-15, solver_socp_inPALM.m:96).
__global__ void __launch_bounds__(256) k_spectral_divide(double *__restrict__ data, i64 ny, i64 nxl, i64 nt, i64 x0,
                                                          double kscale, const double *__restrict__ cy,
                                                          const double *__restrict__ cx,
                                                          const double *__restrict__ ct) {
    const i64 y = (i64)blockIdx.x * 64 + threadIdx.x;
    const i64 x = (i64)blockIdx.y * 4 + threadIdx.y;
    const i64 t = blockIdx.z;
    if (y >= ny || x >= nxl) return;
    double lam = (cy[y] + cx[x0 + x]) + ct[t];
    if (lam == 0.0) lam = 1.0;
    const i64 i = y + ny * (x + nxl * t);
    data[i] = data[i] / (kscale * lam);
}

int launch_spectral_divide(double *data, i64 ny, i64 nx, i64 nt, i64 x0, i64 nxl, double kscale, const double *cy,
                           const double *cx, const double *ct, hipStream_t st) {
    (void)nx;
    if (ny * nxl * nt <= 0) return 0;
    dim3 grid((unsigned)((ny + 63) / 64), (unsigned)((nxl + 3) / 4), (unsigned)nt);
    hipLaunchKernelGGL(k_spectral_divide, grid, dim3(64, 4), 0, st, data, ny, nxl, nt, x0, kscale, cy, cx, ct);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
