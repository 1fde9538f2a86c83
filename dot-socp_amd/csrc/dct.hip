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
#include "device_utils.h"
#include "kernels.h"
#include "pfa.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

namespace dotsocp {

struct DctPlan {
    i64 n;
    int log2n;      // -1 when n is not a power of two
    double2 *tw;    // [n/2]  exp(-2 pi i k / n)
    double2 *ww;    // [n]    2 exp(-i pi k / 2n) / sqrt(2n), ww[0] /= sqrt(2)   (mirt_dctn.m:69-70)
    double *Cfwd;   // dense: Cfwd[j*n + k] = C[k][j]   (forward,  out_k = sum_j C[k][j] in_j)
    double *Cinv;   // dense: Cinv[j*n + k] = C[j][k]   (inverse)
    // even / odd split of the dense matrix (k_dct_mfma_split), [contraction index][output index]:
    double *Ef, *Of;   // Ef[j*ne + k'] = C[2k'][j] (j < njE), Of[j*no + k'] = C[2k'+1][j] (j < h)
    double *Ei, *Oi;   // Ei[k'*njE + j] = C[2k'][j],          Oi[k'*h + j]  = C[2k'+1][j]
    int ne, no, h, njE;
    PfaPlan *pfa;   // prime-factor transform for the 2^k+1 lengths (pfa.hip); nullptr: dense product
};

DctPlan *dct_plan_create(i64 n) {
    DctPlan *p = new DctPlan();
    p->n = n;
    p->log2n = -1;
    p->tw = nullptr;
    p->ww = nullptr;
    p->Cfwd = p->Cinv = nullptr;
    p->Ef = p->Of = p->Ei = p->Oi = nullptr;
    p->ne = p->no = p->h = p->njE = 0;
    p->pfa = nullptr;
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
        static const bool pfa_on = !(getenv("DOTSOCP_PFA") && atoi(getenv("DOTSOCP_PFA")) == 0);
        if (pfa_supported(n) && pfa_on) {
            // the prime-factor transform needs three small tables; the n x n matrices of the dense product (16 MB and
            // two million long-double cosines at n = 1025) are not built
            p->pfa = pfa_plan_create(n);
            if (!p->pfa) {
                dct_plan_destroy(p);
                return nullptr;
            }
            return p;
        }
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
        if (n >= 48) {
            const int h = (int)(n / 2), ne = (int)((n + 1) / 2), no = (int)(n / 2), njE = h + (int)(n & 1);
            p->ne = ne; p->no = no; p->h = h; p->njE = njE;
            auto Cm = [&](i64 k, i64 j) { return cf[(size_t)j * n + k]; };
            std::vector<double> ef((size_t)njE * ne), of((size_t)h * no), ei((size_t)ne * njE), oi((size_t)no * h);
            for (int j = 0; j < njE; ++j)
                for (int k = 0; k < ne; ++k) ef[(size_t)j * ne + k] = ei[(size_t)k * njE + j] = Cm(2 * k, j);
            for (int j = 0; j < h; ++j)
                for (int k = 0; k < no; ++k) of[(size_t)j * no + k] = oi[(size_t)k * h + j] = Cm(2 * k + 1, j);
            double **dst4[4] = {&p->Ef, &p->Of, &p->Ei, &p->Oi};
            std::vector<double> *src4[4] = {&ef, &of, &ei, &oi};
            for (int i = 0; i < 4; ++i) {
                if (hipMalloc(dst4[i], sizeof(double) * src4[i]->size()) != hipSuccess) {
                    dct_plan_destroy(p);
                    return nullptr;
                }
                (void)hipMemcpy(*dst4[i], src4[i]->data(), sizeof(double) * src4[i]->size(), hipMemcpyHostToDevice);
            }
        }
    }
    return p;
}

void dct_plan_destroy(DctPlan *p) {
    if (!p) return;
    if (p->tw) (void)hipFree(p->tw);
    if (p->ww) (void)hipFree(p->ww);
    if (p->Cfwd) (void)hipFree(p->Cfwd);
    if (p->Cinv) (void)hipFree(p->Cinv);
    if (p->Ef) (void)hipFree(p->Ef);
    if (p->Of) (void)hipFree(p->Of);
    if (p->Ei) (void)hipFree(p->Ei);
    if (p->Oi) (void)hipFree(p->Oi);
    pfa_plan_destroy(p->pfa);
    delete p;
}

// Line addressing shared by all axes: line L, element k lives at
//   (L % nin) + (L / nin) * outerStride + k * nin
// axis 0: nin = 1, outerStride = n;  axis 1: nin = n0, outerStride = n0*n1;  axis 2: nin = n0*n1.
// With pitched rows (pitch >= n0) the element stride is no longer the line count per group:
// axis 0: nin = 1, outerStride = pitch, es = 1;  axis 1: nin = n0, outerStride = pitch*n1, es = pitch;
// axis 2: nin = n0, outerStride = pitch, es = pitch*n1.  (The power-of-two kernels run unpitched: es == nin there.)
struct LineMap {
    i64 nin, outerStride, nLines, es;
    __device__ __forceinline__ i64 base(i64 L) const { return (L % nin) + (L / nin) * outerStride; }
    __device__ __forceinline__ i64 addr(i64 L, i64 k) const { return base(L) + k * es; }
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ int bitrev(int k, int lg) { return (int)(__brev((unsigned)k) >> (32 - lg)); }

#define DCT_THREADS 256
#define DCT_WAVES 4
#define DCT_BATCH 8   // global loads in flight per lane before the first dependent LDS write
// Position of element p inside its LDS row: the low four bits (which sixteenth of the 64 banks a 16-byte element
// falls on) are XOR-ed with the next two groups of four bits, so that the stride-16 / stride-64 / ... accesses of the
// grouped FFT stages AND the bit-reversed reads of the post-processing (consecutive k -> multiples of n / 16 apart)
// spread over all banks; a permutation inside aligned blocks of 16, so rows need no padding.  (Additive padding
// p + p / 16 left the bit-reversed reads four deep on the same banks and cost n / 16 elements per row.)
__device__ __host__ __forceinline__ int padi(int p) { return p ^ ((p >> 4) & 15) ^ ((p >> 8) & 15); }
__device__ __host__ __forceinline__ int row_stride(int n) { return n + (n >> 4) + 1; }

// LDS hand-off between the lanes of ONE wavefront: DS operations of a wave execute in order, so
// draining the wave's outstanding LDS operations is all the synchronisation that is needed.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Twiddle tables as the kernels see them: a plain pointer, or -- for the 2048-point lines of the pipelined kernels, whose
// two tile buffers leave 32 KB of LDS for tables -- the symmetric part only:
//   exp(-2 pi i (j + n/4) / n) = -i exp(-2 pi i j / n)               -> a quarter of the FFT twiddles,
//   ww[n - k] = (-imag ww[k], -real ww[k])   (0 < k < n/2)           -> half of the DCT weights (+ the entry n/2).
struct TwQuarter {
    const double2 *t;
    int q;                  // n / 4 entries
    __device__ __forceinline__ double2 operator[](int j) const {
        const double2 v = t[j & (q - 1)];
        return (j & q) ? make_double2(v.y, -v.x) : v;
    }
};
struct WwHalf {
    const double2 *t;
    int h;                  // n / 2: entries 0 .. h
    __device__ __forceinline__ double2 operator[](int m) const {
        const double2 v = t[m <= h ? m : 2 * h - m];
        return (m <= h) ? v : make_double2(-v.y, -v.x);
    }
};

// d * exp(-2 pi i t / 16), t in [0, 8): the constant part of the twiddles inside a register group
__device__ __forceinline__ double2 mul_w16(double2 d, int t) {
    const double h = 0.70710678118654752440;   // cos(pi/4)
    const double c1 = 0.92387953251128675613;  // cos(pi/8)
    const double s1 = 0.38268343236508977173;  // sin(pi/8)
    switch (t) {
        case 0: return d;
        case 1: return make_double2(d.x * c1 + d.y * s1, d.y * c1 - d.x * s1);
        case 2: return make_double2(h * (d.x + d.y), h * (d.y - d.x));
        case 3: return make_double2(d.x * s1 + d.y * c1, d.y * s1 - d.x * c1);
        case 4: return make_double2(d.y, -d.x);
        case 5: return make_double2(d.y * c1 - d.x * s1, -(d.x * c1 + d.y * s1));
        case 6: return make_double2(h * (d.y - d.x), -h * (d.x + d.y));
        default: return make_double2(d.y * s1 - d.x * c1, -(d.x * s1 + d.y * c1));
    }
}

// One group of LR radix-2 decimation-in-frequency stages done in registers: the lane owns the
// R = 2^LR elements base + m * (S/R) of one sub-transform of span S = 2^sl and performs the
// butterflies of spans S, S/2, ..., S/2^(LR-1) on them (same data flow as LR passes of the
// textbook in-place radix-2 DIF, so the output order is plain bit reversal).
// LES > 0: the rows of a tile are interleaved element by element (element p of row r at (padi(p) << LES) + r, `row`
// = tile + r) -- the image an LDS-DMA piece leaves when each lane fetches one (pair, k) element; LES = 0: plain rows
template <int LR, int LES = 0, class TW = const double2 *>
__device__ __forceinline__ void dif_group(double2 *__restrict__ row, int sl, int bidx, int lg, TW tw) {
    constexpr int R = 1 << LR;
    const int strideLog = sl - LR;
    const int j = bidx & ((1 << strideLog) - 1);
    const int base = ((bidx >> strideLog) << sl) + j;
    double2 x[R];
#pragma unroll
    for (int m = 0; m < R; ++m) x[m] = row[padi(base + (m << strideLog)) << LES];
    const int tj = j << (lg - sl);   // j * N / S
#pragma unroll
    for (int u = 0; u < LR; ++u) {
        const int hm = R >> (u + 1);
        const double2 bu = tw[tj << u];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            if ((m / hm) & 1) continue;
            const int mm = m % hm;
            const double2 a = x[m], b = x[m + hm];
            x[m] = make_double2(a.x + b.x, a.y + b.y);
            double2 d = make_double2(a.x - b.x, a.y - b.y);
            d = mul_w16(d, (mm << u) * (16 / R));
            x[m + hm] = cmul(d, bu);
        }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) row[padi(base + (m << strideLog)) << LES] = x[m];
}

// FFT of the `nrows` = 2^lrw complex rows (length n = 2^lg) owned by the CALLING WAVE, in LDS:
// natural order in, bit-reversed order out; ceil(lg/4) register groups with a wave-level LDS
// hand-off after each (no workgroup barrier).
__device__ __forceinline__ void fft_rows_wave(double2 *rows, int lrw, int lg, int rowStride, int lane,
                                              const double2 *__restrict__ tw) {
    const int nst = (lg + 3) >> 2;
    const int baseBits = lg / nst, extra = lg % nst;
    int sl = lg;
    for (int st = 0; st < nst; ++st) {
        const int lr = baseBits + (st < extra ? 1 : 0);
        const int lpr = lg - lr;                        // log2(butterflies per row)
        const int total = 1 << (lrw + lpr);
        for (int b = lane; b < total; b += 64) {
            double2 *r = rows + (b >> lpr) * rowStride;
            const int bidx = b & ((1 << lpr) - 1);
            switch (lr) {
                case 4: dif_group<4>(r, sl, bidx, lg, tw); break;
                case 3: dif_group<3>(r, sl, bidx, lg, tw); break;
                case 2: dif_group<2>(r, sl, bidx, lg, tw); break;
                default: dif_group<1>(r, sl, bidx, lg, tw); break;
            }
        }
        sl -= lr;
        wave_lds_sync();
    }
}

// Decimation-in-time twin of dif_group: same element set (base + m * S/R), the butterflies of spans S/2^(LR-1),
// ..., S/2, S in INCREASING order with the twiddle applied before the add / subtract -- bit-reversed input,
// natural-order output.  Used where the spectrum is needed in place in natural order (fused t-axis solve).
template <int LR, int LES = 0, class TW = const double2 *>
__device__ __forceinline__ void dit_group(double2 *__restrict__ row, int sl, int bidx, int lg, TW tw) {
    constexpr int R = 1 << LR;
    const int strideLog = sl - LR;
    const int j = bidx & ((1 << strideLog) - 1);
    const int base = ((bidx >> strideLog) << sl) + j;
    double2 x[R];
#pragma unroll
    for (int m = 0; m < R; ++m) x[m] = row[padi(base + (m << strideLog)) << LES];
    const int tj = j << (lg - sl);   // j * N / S
#pragma unroll
    for (int u = LR - 1; u >= 0; --u) {
        const int hm = R >> (u + 1);
        const double2 bu = tw[tj << u];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            if ((m / hm) & 1) continue;
            const int mm = m % hm;
            const double2 a = x[m];
            const double2 t = cmul(mul_w16(x[m + hm], (mm << u) * (16 / R)), bu);
            x[m] = make_double2(a.x + t.x, a.y + t.y);
            x[m + hm] = make_double2(a.x - t.x, a.y - t.y);
        }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) row[padi(base + (m << strideLog)) << LES] = x[m];
}

// FFT of the calling wave's rows, bit-reversed order in, natural order out (the register groups of
// fft_rows_wave in reverse order).
__device__ __forceinline__ void fft_rows_wave_dit(double2 *rows, int lrw, int lg, int rowStride, int lane,
                                                  const double2 *__restrict__ tw) {
    const int nst = (lg + 3) >> 2;
    const int baseBits = lg / nst, extra = lg % nst;
    int sl = 0;
    for (int st = nst - 1; st >= 0; --st) {
        const int lr = baseBits + (st < extra ? 1 : 0);
        sl += lr;
        const int lpr = lg - lr;
        const int total = 1 << (lrw + lpr);
        for (int b = lane; b < total; b += 64) {
            double2 *r = rows + (b >> lpr) * rowStride;
            const int bidx = b & ((1 << lpr) - 1);
            switch (lr) {
                case 4: dit_group<4>(r, sl, bidx, lg, tw); break;
                case 3: dit_group<3>(r, sl, bidx, lg, tw); break;
                case 2: dit_group<2>(r, sl, bidx, lg, tw); break;
                default: dit_group<1>(r, sl, bidx, lg, tw); break;
            }
        }
        wave_lds_sync();
    }
}

// Makhoul reordering v[j] = x[2j], v[n-1-j] = x[2j+1] (mirt_dctn.m:71) -- also the output
// reordering of the inverse (mirt_idctn.m:71-73,120).
__device__ __forceinline__ int makhoul(int k, int n) { return (k & 1) ? (n - 1 - (k >> 1)) : (k >> 1); }

// LDS position of input element k while staging a line: forward transforms take the Makhoul order, the inverse
// the natural one, the fused t-axis solve the bit-reversed Makhoul order (its forward FFT is decimation-in-time)
template <int MODE>
__device__ __forceinline__ int stage_pos(int k, int n, int lg) {
    return MODE == 1 ? k : (MODE == 2 ? bitrev(makhoul(k, n), lg) : makhoul(k, n));
}

// Inverse pre-processing on the calling wave's rows (natural order, Xa + i Xb elementwise):
//   G[k] = (ww[k] X[k] + conj(ww[n-k]) X[n-k]) / 2, so that fft(G) = real(fft(ww .* X))
//   (mirt_idctn.m:109,119-120).  k and n-k are handled by the same lane.
__device__ __forceinline__ void idct_combine_wave(double2 *rows, int lrw, int lg, int rowStride, int lane,
                                                  const double2 *__restrict__ ww) {
    const int n = 1 << lg, lh = lg - 1;
    const int total = 1 << (lrw + lh);
    for (int b = lane; b < total; b += 64) {
        double2 *r = rows + (b >> lh) * rowStride;
        const int k = (b & ((1 << lh) - 1)) + 1;          // 1 .. n/2
        const int m = n - k;
        const double2 xk = r[padi(k)], xm = r[padi(m)];
        const double2 wk = ww[k], wm = ww[m];
        const double gar = 0.5 * (wk.x * xk.x + wm.x * xm.x), gai = 0.5 * (wk.y * xk.x - wm.y * xm.x);
        const double gbr = 0.5 * (wk.x * xk.y + wm.x * xm.y), gbi = 0.5 * (wk.y * xk.y - wm.y * xm.y);
        r[padi(k)] = make_double2(gar - gbi, gai + gbr);
        if (m != k) {
            const double har = 0.5 * (wm.x * xm.x + wk.x * xk.x), hai = 0.5 * (wm.y * xm.x - wk.y * xk.x);
            const double hbr = 0.5 * (wm.x * xm.y + wk.x * xk.y), hbi = 0.5 * (wm.y * xm.y - wk.y * xk.y);
            r[padi(m)] = make_double2(har - hbi, hai + hbr);
        }
    }
    if (lane < (1 << lrw)) {
        double2 *r = rows + lane * rowStride;
        const double w0 = ww[0].x;
        r[0] = make_double2(w0 * r[0].x, w0 * r[0].y);
    }
    wave_lds_sync();
}

// (Xa[k], Xb[k]) = real(ww[k] * V_{a,b}[k]) from the bit-reversed FFT of va + i vb:
// V_a = (V[k] + conj(V[n-k])) / 2, V_b = (V[k] - conj(V[n-k])) / (2i)   (mirt_dctn.m:130)
template <int LES = 0, class WW = const double2 *>
__device__ __forceinline__ double2 dct_post(const double2 *__restrict__ r, int k, int n, int lg, WW ww) {
    const double2 vk = r[padi(bitrev(k, lg)) << LES];
    const double2 vm = r[padi(bitrev((n - k) & (n - 1), lg)) << LES];
    const double2 w = ww[k];
    const double ar = 0.5 * (vk.x + vm.x), ai = 0.5 * (vk.y - vm.y);
    const double br = 0.5 * (vk.y + vm.y), bi = -0.5 * (vk.x - vm.x);
    return make_double2(w.x * ar - w.y * ai, w.x * br - w.y * bi);
}

// ---------------------------------------------------------------------------------------------
// Axis 0 (lines contiguous in memory): every wave works alone on its own 2^lrw complex rows
// (pairs of consecutive lines) -- 16-byte global accesses, no workgroup barrier at all.
// ---------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ void __launch_bounds__(DCT_THREADS) k_dct_axis0(const double *__restrict__ src, double *__restrict__ dst,
                                                            i64 nLines, i64 ls /* doubles between lines */, int lg, int lrw,
                                                            const double2 *__restrict__ tw,
                                                            const double2 *__restrict__ ww) {
    extern __shared__ double2 lds[];
    const int n = 1 << lg, lh = lg - 1;
    const int rowStride = row_stride(n);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rw = 1 << lrw;
    double2 *rows = lds + (wave << lrw) * rowStride;
    const i64 pair0 = ((i64)blockIdx.x * DCT_WAVES + wave) << lrw;      // first pair of lines of this wave
    const int total = 1 << (lrw + lh);                                    // (row, j) with j = k / 2
    // ---- load: two consecutive elements of both lines per lane; DCT_BATCH iterations' worth of
    // 16-byte global loads are issued before the first LDS write so that their latencies overlap ----
    for (int b0 = lane; b0 < total; b0 += 64 * DCT_BATCH) {
        double2 A[DCT_BATCH], B[DCT_BATCH];
#pragma unroll
        for (int u = 0; u < DCT_BATCH; ++u) {
            const int b = b0 + 64 * u;
            const int rr = b >> lh, j = b & ((1 << lh) - 1);
            const i64 La = 2 * (pair0 + rr);
            A[u] = make_double2(0.0, 0.0);
            B[u] = A[u];
            if (b < total && La < nLines) A[u] = *(const double2 *)(src + La * ls + 2 * j);
            if (b < total && La + 1 < nLines) B[u] = *(const double2 *)(src + (La + 1) * ls + 2 * j);
        }
#pragma unroll
        for (int u = 0; u < DCT_BATCH; ++u) {
            const int b = b0 + 64 * u;
            if (b >= total) break;
            const int rr = b >> lh, j = b & ((1 << lh) - 1);
            double2 *r = rows + rr * rowStride;
            if (!INVERSE) {
                r[padi(j)] = make_double2(A[u].x, B[u].x);               // x[2j]   -> v[j]
                r[padi(n - 1 - j)] = make_double2(A[u].y, B[u].y);       // x[2j+1] -> v[n-1-j]
            } else {
                r[padi(2 * j)] = make_double2(A[u].x, B[u].x);
                r[padi(2 * j + 1)] = make_double2(A[u].y, B[u].y);
            }
        }
    }
    wave_lds_sync();
    if (INVERSE) idct_combine_wave(rows, lrw, lg, rowStride, lane, ww);
    fft_rows_wave(rows, lrw, lg, rowStride, lane, tw);
    // ---- store ----
    for (int b = lane; b < total; b += 64) {
        const int rr = b >> lh, j = b & ((1 << lh) - 1);
        const i64 La = 2 * (pair0 + rr);
        const double2 *r = rows + rr * rowStride;
        double2 A, B;
        if (!INVERSE) {
            const double2 p0 = dct_post(r, 2 * j, n, lg, ww), p1 = dct_post(r, 2 * j + 1, n, lg, ww);
            A = make_double2(p0.x, p1.x);
            B = make_double2(p0.y, p1.y);
        } else {
            const double2 v0 = r[padi(bitrev(j, lg))], v1 = r[padi(bitrev(n - 1 - j, lg))];
            A = make_double2(v0.x, v1.x);                      // x[2j] = v[j], x[2j+1] = v[n-1-j]
            B = make_double2(v0.y, v1.y);
        }
        if (La < nLines) *(double2 *)(dst + La * ls + 2 * j) = A;
        if (La + 1 < nLines) *(double2 *)(dst + (La + 1) * ls + 2 * j) = B;
    }
    (void)rw;
}

// ---------------------------------------------------------------------------------------------
// Strided axes (x, t): the workgroup stages 2^lp complex rows = 2^(lp+1) lines that are
// CONSECUTIVE in memory, loads / stores them cooperatively (VEC: one 16-byte access carries both
// lines of a pair), and every wave runs the FFT of its own rows between the two barriers.
// TSOLVE: forward DCT, division by the spectral kernel, inverse DCT in one pass (t axis).
// ---------------------------------------------------------------------------------------------
struct SolveArgs {
    i64 ny, line0, nplane; // TSOLVE: local line L is column (y, x) = (G % ny, G / ny), G = line0 + L, of ny*nx = nplane columns
    double kscale;
    const double *cy, *cx, *ct;
};

template <int MODE /*0 fwd, 1 inv, 2 t-solve*/, bool VEC>
__global__ void __launch_bounds__(DCT_THREADS) k_dct_strided(const double *__restrict__ src, double *__restrict__ dst,
                                                              LineMap map, int lg, int lp, SolveArgs sa,
                                                              const double2 *__restrict__ tw,
                                                              const double2 *__restrict__ ww) {
    extern __shared__ double2 lds[];
    const int n = 1 << lg;
    const int rowStride = row_stride(n);
    const int npairs = 1 << lp;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const i64 L0 = xcd_tile(blockIdx.x, gridDim.x) << (lp + 1);
    // rows of this wave: npairs / 4 each (all rows go to the first waves when npairs < 4)
    const int lrw = (lp >= 2) ? lp - 2 : 0;
    const bool waveActive = (wave << lrw) < npairs;
    double2 *rows = lds + (wave << lrw) * rowStride;
    // ---- cooperative load ----
    if (VEC) {
        const int r = tid & (npairs - 1);
        const i64 L = L0 + 2 * r;
        const bool ok = L < map.nLines;
        const i64 lb = ok ? map.base(L) : 0;
        const int kstep = DCT_THREADS >> lp;
        for (int k0 = tid >> lp; k0 < n; k0 += kstep * DCT_BATCH) {
            double2 gv[DCT_BATCH];
#pragma unroll
            for (int u = 0; u < DCT_BATCH; ++u) {
                const int k = k0 + u * kstep;
                gv[u] = (ok && k < n) ? *(const double2 *)(src + lb + (i64)k * map.es) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < DCT_BATCH; ++u) {
                const int k = k0 + u * kstep;
                if (k < n) lds[r * rowStride + padi(stage_pos<MODE>(k, n, lg))] = gv[u];
            }
        }
    } else {
        const int l = tid & (2 * npairs - 1);
        const i64 L = L0 + l;
        const bool ok = L < map.nLines;
        const i64 lb = ok ? map.base(L) : 0;
        for (int k = tid >> (lp + 1); k < n; k += DCT_THREADS >> (lp + 1)) {
            const double g = ok ? src[lb + (i64)k * map.es] : 0.0;
            ((double *)&lds[(l >> 1) * rowStride + padi(stage_pos<MODE>(k, n, lg))])[l & 1] = g;
        }
    }
    __syncthreads();
    if (waveActive) {
        if (MODE == 1) idct_combine_wave(rows, lrw, lg, rowStride, lane, ww);
        if (MODE != 2) {
            fft_rows_wave(rows, lrw, lg, rowStride, lane, tw);
        } else {
            // forward transform with natural-order output, then -- in place, the lane that owns k also owns n-k --
            // X = DCT post-processing (dct_post), Y = X / (kscale * lambda), G = inverse pre-processing
            // (idct_combine_wave) in one go, then the inverse transform on the same rows
            fft_rows_wave_dit(rows, lrw, lg, rowStride, lane, tw);
            const int lh = lg - 1;
            const int total = 1 << (lrw + lh);
            for (int b = lane; b < total; b += 64) {
                const int rr = b >> lh;
                double2 *r = rows + rr * rowStride;
                i64 La = L0 + 2 * ((wave << lrw) + rr);
                if (La + 1 >= map.nLines) La = (map.nLines >= 2) ? map.nLines - 2 : 0;
                const i64 Ga = sa.line0 + La;
                const i64 Gb = (Ga + 1 < sa.nplane) ? Ga + 1 : Ga;
                const double ea = sa.cy[Ga % sa.ny] + sa.cx[Ga / sa.ny];                    // CY + CX of line a
                const double eb = sa.cy[Gb % sa.ny] + sa.cx[Gb / sa.ny];
                const int k = (b & ((1 << lh) - 1)) + 1;          // 1 .. n/2
                const int m = n - k;
                const double2 vk = r[padi(k)], vm = r[padi(m)];
                const double2 wk = ww[k], wm = ww[m];
                const double ar = 0.5 * (vk.x + vm.x), ai = 0.5 * (vk.y - vm.y);
                const double br = 0.5 * (vk.y + vm.y), bi = -0.5 * (vk.x - vm.x);
                const double ctk = sa.ct[k], ctm = sa.ct[m];
                double lak = ea + ctk, lbk = eb + ctk, lam = ea + ctm, lbm = eb + ctm;
                if (lak == 0.0) lak = 1.0;
                if (lbk == 0.0) lbk = 1.0;
                if (lam == 0.0) lam = 1.0;
                if (lbm == 0.0) lbm = 1.0;
                // Y[k], Y[n-k]: .x = line a, .y = line b
                const double2 xk = make_double2((wk.x * ar - wk.y * ai) / (sa.kscale * lak),
                                                (wk.x * br - wk.y * bi) / (sa.kscale * lbk));
                const double2 xm = make_double2((wm.x * ar + wm.y * ai) / (sa.kscale * lam),
                                                (wm.x * br + wm.y * bi) / (sa.kscale * lbm));
                const double gar = 0.5 * (wk.x * xk.x + wm.x * xm.x), gai = 0.5 * (wk.y * xk.x - wm.y * xm.x);
                const double gbr = 0.5 * (wk.x * xk.y + wm.x * xm.y), gbi = 0.5 * (wk.y * xk.y - wm.y * xm.y);
                r[padi(k)] = make_double2(gar - gbi, gai + gbr);
                if (m != k) {
                    const double har = 0.5 * (wm.x * xm.x + wk.x * xk.x), hai = 0.5 * (wm.y * xm.x - wk.y * xk.x);
                    const double hbr = 0.5 * (wm.x * xm.y + wk.x * xk.y), hbi = 0.5 * (wm.y * xm.y - wk.y * xk.y);
                    r[padi(m)] = make_double2(har - hbi, hai + hbr);
                }
            }
            if (lane < (1 << lrw)) {                               // k = 0: V[0] is its own partner
                double2 *r = rows + lane * rowStride;
                i64 La = L0 + 2 * ((wave << lrw) + lane);
                if (La + 1 >= map.nLines) La = (map.nLines >= 2) ? map.nLines - 2 : 0;
                const i64 Ga = sa.line0 + La;
                const i64 Gb = (Ga + 1 < sa.nplane) ? Ga + 1 : Ga;
                double la = (sa.cy[Ga % sa.ny] + sa.cx[Ga / sa.ny]) + sa.ct[0];
                double lb2 = (sa.cy[Gb % sa.ny] + sa.cx[Gb / sa.ny]) + sa.ct[0];
                if (la == 0.0) la = 1.0;
                if (lb2 == 0.0) lb2 = 1.0;
                const double w0 = ww[0].x;
                const double2 v0 = r[0];
                r[0] = make_double2(w0 * ((w0 * v0.x) / (sa.kscale * la)), w0 * ((w0 * v0.y) / (sa.kscale * lb2)));
            }
            wave_lds_sync();
            fft_rows_wave(rows, lrw, lg, rowStride, lane, tw);
        }
    }
    __syncthreads();
    // ---- cooperative store ----
    const double2 *out = lds;
    if (VEC) {
        const int r = tid & (npairs - 1);
        const i64 L = L0 + 2 * r;
        if (L < map.nLines) {
            const i64 lb = map.base(L);
            const double2 *rr = out + r * rowStride;
            for (int k = tid >> lp; k < n; k += DCT_THREADS >> lp) {
                double2 v;
                if (MODE == 0) v = dct_post(rr, k, n, lg, ww);
                else v = rr[padi(bitrev(makhoul(k, n), lg))];
                *(double2 *)(dst + lb + (i64)k * map.es) = v;
            }
        }
    } else {
        const int l = tid & (2 * npairs - 1);
        const i64 L = L0 + l;
        if (L < map.nLines) {
            const i64 lb = map.base(L);
            const double2 *rr = out + (l >> 1) * rowStride;
            for (int k = tid >> (lp + 1); k < n; k += DCT_THREADS >> (lp + 1)) {
                double2 v;
                if (MODE == 0) v = dct_post(rr, k, n, lg, ww);
                else v = rr[padi(bitrev(makhoul(k, n), lg))];
                dst[lb + (i64)k * map.es] = (l & 1) ? v.y : v.x;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Workgroup-wide flavour: ALL threads of the workgroup share ALL staged rows (butterfly groups are
// dealt round-robin to the T threads, __syncthreads() between register groups).  Twice the waves per
// staged row of the per-wave flavour above at the same LDS footprint -- the footprint, not registers,
// caps the resident workgroups per CU, so this doubles the waves that overlap VALU, LDS and HBM phases.
// ---------------------------------------------------------------------------------------------
template <bool RAWB = false>
__device__ __forceinline__ void fft_rows_wg(double2 *rows, int lrows, int lg, int rowStride, int t, int T,
                                            const double2 *__restrict__ tw) {
    const int nst = (lg + 3) >> 2;
    const int baseBits = lg / nst, extra = lg % nst;
    int sl = lg;
    for (int st = 0; st < nst; ++st) {
        const int lr = baseBits + (st < extra ? 1 : 0);
        const int lpr = lg - lr;
        const int total = 1 << (lrows + lpr);
        for (int b = t; b < total; b += T) {
            double2 *r = rows + (b >> lpr) * rowStride;
            const int bidx = b & ((1 << lpr) - 1);
            switch (lr) {
                case 4: dif_group<4>(r, sl, bidx, lg, tw); break;
                case 3: dif_group<3>(r, sl, bidx, lg, tw); break;
                case 2: dif_group<2>(r, sl, bidx, lg, tw); break;
                default: dif_group<1>(r, sl, bidx, lg, tw); break;
            }
        }
        sl -= lr;
        if (RAWB) lds_barrier(); else __syncthreads();
    }
}

template <bool RAWB = false, class WW = const double2 *>
__device__ __forceinline__ void idct_combine_wg(double2 *rows, int lrows, int lg, int rowStride, int t, int T, WW ww) {
    const int n = 1 << lg, lh = lg - 1;
    const int total = 1 << (lrows + lh);
    for (int b = t; b < total; b += T) {
        double2 *r = rows + (b >> lh) * rowStride;
        const int k = (b & ((1 << lh) - 1)) + 1;          // 1 .. n/2
        const int m = n - k;
        const double2 xk = r[padi(k)], xm = r[padi(m)];
        const double2 wk = ww[k], wm = ww[m];
        const double gar = 0.5 * (wk.x * xk.x + wm.x * xm.x), gai = 0.5 * (wk.y * xk.x - wm.y * xm.x);
        const double gbr = 0.5 * (wk.x * xk.y + wm.x * xm.y), gbi = 0.5 * (wk.y * xk.y - wm.y * xm.y);
        r[padi(k)] = make_double2(gar - gbi, gai + gbr);
        if (m != k) {
            const double har = 0.5 * (wm.x * xm.x + wk.x * xk.x), hai = 0.5 * (wm.y * xm.x - wk.y * xk.x);
            const double hbr = 0.5 * (wm.x * xm.y + wk.x * xk.y), hbi = 0.5 * (wm.y * xm.y - wk.y * xk.y);
            r[padi(m)] = make_double2(har - hbi, hai + hbr);
        }
    }
    if (t < (1 << lrows)) {
        double2 *r = rows + t * rowStride;
        const double w0 = ww[0].x;
        r[0] = make_double2(w0 * r[0].x, w0 * r[0].y);
    }
    if (RAWB) lds_barrier(); else __syncthreads();
}

#define DCT_WG_THREADS 512
// Axis 0, workgroup-wide: the workgroup stages 2^lrows complex rows (pairs of consecutive lines).
template <bool INVERSE>
__global__ void __launch_bounds__(DCT_WG_THREADS, 4) k_dct_axis0_wg(const double *__restrict__ src,
                                                                     double *__restrict__ dst, i64 nLines, i64 ls, int lg,
                                                                     int lrows, const double2 *__restrict__ tw,
                                                                     const double2 *__restrict__ ww) {
    extern __shared__ double2 lds[];
    const int n = 1 << lg, lh = lg - 1;
    const int rowStride = row_stride(n);
    const int tid = threadIdx.x;
    const i64 pair0 = (i64)blockIdx.x << lrows;
    const int total = 1 << (lrows + lh);
    for (int b0 = tid; b0 < total; b0 += DCT_WG_THREADS * DCT_BATCH) {
        double2 A[DCT_BATCH], B[DCT_BATCH];
#pragma unroll
        for (int u = 0; u < DCT_BATCH; ++u) {
            const int b = b0 + DCT_WG_THREADS * u;
            const int rr = b >> lh, j = b & ((1 << lh) - 1);
            const i64 La = 2 * (pair0 + rr);
            A[u] = make_double2(0.0, 0.0);
            B[u] = A[u];
            if (b < total && La < nLines) A[u] = *(const double2 *)(src + La * ls + 2 * j);
            if (b < total && La + 1 < nLines) B[u] = *(const double2 *)(src + (La + 1) * ls + 2 * j);
        }
#pragma unroll
        for (int u = 0; u < DCT_BATCH; ++u) {
            const int b = b0 + DCT_WG_THREADS * u;
            if (b >= total) break;
            const int rr = b >> lh, j = b & ((1 << lh) - 1);
            double2 *r = lds + rr * rowStride;
            if (!INVERSE) {
                r[padi(j)] = make_double2(A[u].x, B[u].x);
                r[padi(n - 1 - j)] = make_double2(A[u].y, B[u].y);
            } else {
                r[padi(2 * j)] = make_double2(A[u].x, B[u].x);
                r[padi(2 * j + 1)] = make_double2(A[u].y, B[u].y);
            }
        }
    }
    __syncthreads();
    if (INVERSE) idct_combine_wg(lds, lrows, lg, rowStride, tid, DCT_WG_THREADS, ww);
    fft_rows_wg(lds, lrows, lg, rowStride, tid, DCT_WG_THREADS, tw);
    for (int b = tid; b < total; b += DCT_WG_THREADS) {
        const int rr = b >> lh, j = b & ((1 << lh) - 1);
        const i64 La = 2 * (pair0 + rr);
        const double2 *r = lds + rr * rowStride;
        double2 A, B;
        if (!INVERSE) {
            const double2 p0 = dct_post(r, 2 * j, n, lg, ww), p1 = dct_post(r, 2 * j + 1, n, lg, ww);
            A = make_double2(p0.x, p1.x);
            B = make_double2(p0.y, p1.y);
        } else {
            const double2 v0 = r[padi(bitrev(j, lg))], v1 = r[padi(bitrev(n - 1 - j, lg))];
            A = make_double2(v0.x, v1.x);
            B = make_double2(v0.y, v1.y);
        }
        if (La < nLines) *(double2 *)(dst + La * ls + 2 * j) = A;
        if (La + 1 < nLines) *(double2 *)(dst + (La + 1) * ls + 2 * j) = B;
    }
}

// ---------------------------------------------------------------------------------------------
// Pipelined flavour (axis 0, n = 128 .. 2048).  What limits the kernels above is not a unit but the bytes in flight:
// while a workgroup computes, its tile sits in LDS and nothing of it travels, and the LDS holds two tiles only (the
// same kernels with the transform skipped run at the copy rate; the transform's time adds in full).  Here ONE
// persistent workgroup of 512 threads per CU (two waves per SIMD, 256 registers each) walks tiles of 8192 doubles
// (whole lines, contiguous in memory) through two LDS buffers, and the lines of a tile arrive by LDS-DMA
// (global_load_lds_dwordx4: no registers, so the loads of tile k+1 and k+2 are in flight during the transform and the
// stores of tile k).  Per tile:
//   raw lines -> paired rows in Makhoul / natural order, in place (all reads, barrier, all writes) | [inverse
//   pre-processing] | FFT | post-processing + stores | wait for the DMA of tile k+1 (counted: only the stores just
//   issued are younger and stay in flight) | barrier | DMA of tile k+2 into the buffer just drained.
// The twiddle tables live in LDS too: an ordinary global load in the loop would make the compiler wait for
// everything in flight.  LDS: 2 x 4 x (n + 1) x 16 B + 1.5 n x 16 B = 152 KB at n = 1024.
// ---------------------------------------------------------------------------------------------
#define PIPE_THREADS 512
#define PIPE_IT ((1 << (PIPE_LG_CPLX - 1)) / PIPE_THREADS)   // (row, j) items per thread
#define PIPE_LG_CPLX 12   // complex elements per tile: 4096 = 8192 doubles = 64 KB of lines
#define PIPE_NS 8         // global store instructions per wave and tile
#define PIPE_ND 8         // LDS-DMA instructions per wave and tile (64 pieces of 1 KB over 8 waves)

// A double from a wave-uniform global address through the scalar cache: no vector-memory operation, so nothing the
// counted waits of the pipelined kernels would have to account for (the compiler takes a vector load for such a read
// when it cannot prove that the kernel's stores leave the table alone, and then waits for everything in flight).
__device__ __forceinline__ double sload_f64(const double *p) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
    const double *sp = (const double *)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    double v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(sp) : "memory");
    return v;
}

// the register groups of fft_rows_wg for a length known at compile time (same plan, same arithmetic)
// the same register groups on a pair-interleaved tile (dif_group<., LES>): item b = (row b % rows, group b / rows), so the
// lanes of a wave sweep the rows of one element first -- consecutive LDS addresses
template <int LG, int LROWS, int T, int ST = 0, int SL = LG, class TW = const double2 *>
__device__ __forceinline__ void fft_tile_pipe(double2 *tile, int t, TW tw) {
    constexpr int NST = (LG + 3) >> 2;
    constexpr int BASEB = LG / NST, EXTRA = LG % NST;
    if constexpr (ST < NST) {
        constexpr int LR = BASEB + (ST < EXTRA ? 1 : 0);
        constexpr int LPR = LG - LR;
        constexpr int TOTAL = 1 << (LROWS + LPR);
#pragma unroll
        for (int b = t; b < TOTAL; b += T)
            dif_group<LR, LROWS>(tile + (b & ((1 << LROWS) - 1)), SL, b >> LROWS, LG, tw);
        lds_barrier();
        fft_tile_pipe<LG, LROWS, T, ST + 1, SL - LR>(tile, t, tw);
    }
}

template <int LG, int LROWS, int T, int ST = ((LG + 3) >> 2) - 1, int SL = 0>
__device__ __forceinline__ void fft_tile_pipe_dit(double2 *tile, int t, const double2 *__restrict__ tw) {
    constexpr int NST = (LG + 3) >> 2;
    constexpr int BASEB = LG / NST, EXTRA = LG % NST;
    if constexpr (ST >= 0) {
        constexpr int LR = BASEB + (ST < EXTRA ? 1 : 0);
        constexpr int SL2 = SL + LR;
        constexpr int LPR = LG - LR;
        constexpr int TOTAL = 1 << (LROWS + LPR);
#pragma unroll
        for (int b = t; b < TOTAL; b += T)
            dit_group<LR, LROWS>(tile + (b & ((1 << LROWS) - 1)), SL2, b >> LROWS, LG, tw);
        lds_barrier();
        fft_tile_pipe_dit<LG, LROWS, T, ST - 1, SL2>(tile, t, tw);
    }
}

// inverse pre-processing (idct_combine_wg) on a pair-interleaved tile
template <int LG, int LROWS, int T, class WW = const double2 *>
__device__ __forceinline__ void idct_combine_tile(double2 *tile, int t, WW ww) {
    constexpr int n = 1 << LG, lh = LG - 1;
    constexpr int TOTAL = 1 << (LROWS + lh);
#pragma unroll
    for (int b = t; b < TOTAL; b += T) {
        double2 *r = tile + (b & ((1 << LROWS) - 1));
        const int k = (b >> LROWS) + 1;                     // 1 .. n/2
        const int m = n - k;
        const int ik = padi(k) << LROWS, im = padi(m) << LROWS;
        const double2 xk = r[ik], xm = r[im];
        const double2 wk = ww[k], wm = ww[m];
        const double gar = 0.5 * (wk.x * xk.x + wm.x * xm.x), gai = 0.5 * (wk.y * xk.x - wm.y * xm.x);
        const double gbr = 0.5 * (wk.x * xk.y + wm.x * xm.y), gbi = 0.5 * (wk.y * xk.y - wm.y * xm.y);
        r[ik] = make_double2(gar - gbi, gai + gbr);
        if (m != k) {
            const double har = 0.5 * (wm.x * xm.x + wk.x * xk.x), hai = 0.5 * (wm.y * xm.x - wk.y * xk.x);
            const double hbr = 0.5 * (wm.x * xm.y + wk.x * xk.y), hbi = 0.5 * (wm.y * xm.y - wk.y * xk.y);
            r[im] = make_double2(har - hbi, hai + hbr);
        }
    }
    if (t < (1 << LROWS)) {
        const double w0 = ww[0].x;
        tile[t] = make_double2(w0 * tile[t].x, w0 * tile[t].y);
    }
    lds_barrier();
}

template <int LG, int LROWS, int T, int RS, int ST = 0, int SL = LG, class TW = const double2 *>
__device__ __forceinline__ void fft_rows_pipe(double2 *rows, int t, TW tw) {
    constexpr int NST = (LG + 3) >> 2;
    constexpr int BASEB = LG / NST, EXTRA = LG % NST;
    if constexpr (ST < NST) {
        constexpr int LR = BASEB + (ST < EXTRA ? 1 : 0);
        constexpr int LPR = LG - LR;
        constexpr int TOTAL = 1 << (LROWS + LPR);
#pragma unroll
        for (int b = t; b < TOTAL; b += T) dif_group<LR>(rows + (b >> LPR) * RS, SL, b & ((1 << LPR) - 1), LG, tw);
        lds_barrier();
        fft_rows_pipe<LG, LROWS, T, RS, ST + 1, SL - LR>(rows, t, tw);
    }
}

template <bool INVERSE, int LG>
__global__ void __launch_bounds__(PIPE_THREADS) k_dct_axis0_pipe(const double *__restrict__ src, double *__restrict__ dst,
                                                                  int nTiles, i64 ls /* doubles between lines */,
                                                                  const double2 *__restrict__ tw,
                                                                  const double2 *__restrict__ ww) {
    extern __shared__ double2 lds[];
    constexpr int n = 1 << LG, lh = LG - 1;
    constexpr int LPT = (2 << PIPE_LG_CPLX) / n;      // lines per tile
    constexpr int PPL = n / 128;                      // 1-KB DMA pieces per line
    constexpr int RS = n + 1;                         // odd row stride: rows start on different banks
    constexpr int lrows = PIPE_LG_CPLX - LG;          // 2^lrows rows (pairs of lines) per tile
    constexpr int BUF = RS << lrows;                  // complex elements per buffer
    // 2048-point lines: only the symmetric part of the tables fits beside the two buffers (TwQuarter, WwHalf)
    constexpr bool BIG = LG > 10;
    constexpr int NTW = BIG ? (n >> 2) : (n >> 1), NWW = BIG ? (n >> 1) + 1 : n;
    double2 *twS = lds + 2 * BUF, *wwS = twS + NTW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NTW; i += PIPE_THREADS) twS[i] = tw[i];
    for (int i = tid; i < NWW; i += PIPE_THREADS) wwS[i] = ww[i];
    typename std::conditional<BIG, TwQuarter, const double2 *>::type twA;
    typename std::conditional<BIG, WwHalf, const double2 *>::type wwA;
    if constexpr (BIG) { twA = TwQuarter{twS, n >> 2}; wwA = WwHalf{wwS, n >> 1}; } else { twA = twS; wwA = wwS; }
    const unsigned ldsBase = (unsigned)(uintptr_t)lds;
    // the lines of a tile land back to back in LDS (n doubles apart) whatever their distance in memory
    auto dma = [&](int tile, int b) {
        const char *g = (const char *)(src + (i64)tile * LPT * ls) + lane * 16;
        const unsigned l0 = ldsBase + (unsigned)b * (unsigned)(BUF * 16) + (unsigned)(wave * PIPE_ND) * 1024u;
#pragma unroll
        for (int i = 0; i < PIPE_ND; ++i) {
            const int piece = wave * PIPE_ND + i;
            glds16(g + (i64)(piece / PPL) * (ls * 8) + (piece % PPL) * 1024, l0 + (unsigned)i * 1024u);
        }
    };
    int tile = blockIdx.x;
    const int stride = gridDim.x;
    if (tile < nTiles) dma(tile, 0);
    if (tile + stride < nTiles) dma(tile + stride, 1);
    // the first tile has landed when only the second one's DMA is outstanding (vector-memory operations of a wave
    // complete in issue order)
    if (tile + stride < nTiles) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIPE_ND) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    // items of the staging / store loops: (row, j < n / 2) = (b >> lh, b & (n / 2 - 1)) for b = tid + u * threads, four per thread
    auto item_row = [&](int u) { return (tid + u * PIPE_THREADS) >> lh; };
    auto item_j = [&](int u) { return (tid + u * PIPE_THREADS) & ((1 << lh) - 1); };
    for (int it = 0; tile < nTiles; tile += stride, ++it) {
        const int b = it & 1;
        double2 *buf = lds + b * BUF;
        // raw lines (n doubles apart, unpadded) -> rows of pairs: all reads, barrier, all writes (same buffer)
        {
            const double *raw = (const double *)buf;
            double2 A[PIPE_IT], B[PIPE_IT];
#pragma unroll
            for (int u = 0; u < PIPE_IT; ++u) {
                const int rr = item_row(u), j0 = item_j(u);
                A[u] = *(const double2 *)(raw + (2 * rr) * n + 2 * j0);
                B[u] = *(const double2 *)(raw + (2 * rr + 1) * n + 2 * j0);
            }
            lds_barrier();
#pragma unroll
            for (int u = 0; u < PIPE_IT; ++u) {
                double2 *r = buf + item_row(u) * RS;
                const int j0 = item_j(u);
                if (!INVERSE) {
                    r[padi(j0)] = make_double2(A[u].x, B[u].x);
                    r[padi(n - 1 - j0)] = make_double2(A[u].y, B[u].y);
                } else {
                    r[padi(2 * j0)] = make_double2(A[u].x, B[u].x);
                    r[padi(2 * j0 + 1)] = make_double2(A[u].y, B[u].y);
                }
            }
        }
        lds_barrier();
        if (INVERSE) idct_combine_wg<true>(buf, lrows, LG, RS, tid, PIPE_THREADS, wwA);
        fft_rows_pipe<LG, lrows, PIPE_THREADS, RS>(buf, tid, twA);
        double *out = dst + (i64)tile * LPT * ls;
#pragma unroll
        for (int u = 0; u < PIPE_IT; ++u) {
            const int rr = item_row(u), j0 = item_j(u);
            const double2 *r = buf + rr * RS;
            double2 Av, Bv;
            if (!INVERSE) {
                const double2 p0 = dct_post(r, 2 * j0, n, LG, wwA), p1 = dct_post(r, 2 * j0 + 1, n, LG, wwA);
                Av = make_double2(p0.x, p1.x);
                Bv = make_double2(p0.y, p1.y);
            } else {
                const double2 v0 = r[padi(bitrev(j0, LG))], v1 = r[padi(bitrev(n - 1 - j0, LG))];
                Av = make_double2(v0.x, v1.x);
                Bv = make_double2(v0.y, v1.y);
            }
            *(double2 *)(out + (2 * rr) * ls + 2 * j0) = Av;
            *(double2 *)(out + (2 * rr + 1) * ls + 2 * j0) = Bv;
        }
        // the next tile has landed when only this tile's stores are outstanding; one barrier then says both "every wave's
        // pieces of the next tile are in LDS" and "this buffer is drained"
        if (tile + stride < nTiles) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIPE_NS) : "memory");
        lds_barrier();
        if (tile + 2 * stride < nTiles) dma(tile + 2 * stride, b);
    }
}

// Strided axes, pipelined (forward / inverse): a tile = 2^lrows pairs of lines that are consecutive in memory x all n
// elements = 4096 complex values, every (pair, k) one 16-byte access.  The tile lives in LDS pair-interleaved and in the
// order the transform wants: slot (padi(p) << lrows) + r holds position p of pair r, i.e. line element k = 2 p resp.
// 2 (n - 1 - p) + 1 (Makhoul order, forward) or k = p (inverse).  An LDS-DMA piece fills 64 consecutive slots = 64 / NP
// positions of all NP pairs: each lane fetches its own (pair, k) element, a piece still reads 64 / NP whole segments of
// NP x 16 bytes, and the tile is ready for the first butterfly group when it has landed -- no staging pass.  At
// n = 1024 a tile is 64 bytes wide: the workgroups are ordered such that the two tiles sharing every 128-byte line run
// at the same time on the same XCD (one fetch into its L2).
template <int MODE /*0 fwd, 1 inv*/, int LG>
__global__ void __launch_bounds__(PIPE_THREADS) k_dct_strided_pipe(const double *__restrict__ src, double *__restrict__ dst,
                                                                    LineMap map, int nTiles, const double2 *__restrict__ tw,
                                                                    const double2 *__restrict__ ww) {
    extern __shared__ double2 lds[];
    constexpr int n = 1 << LG;
    constexpr int lrows = PIPE_LG_CPLX - LG;          // log2(pairs per tile)
    constexpr int NP = 1 << lrows;
    constexpr int BUF = 1 << PIPE_LG_CPLX;            // complex elements per buffer (no padding: the swizzle permutes)
    constexpr bool BIG = LG > 10;                     // 2048-point lines: symmetric part of the tables only (see k_dct_axis0_pipe)
    constexpr int NTW = BIG ? (n >> 2) : (n >> 1), NWW = BIG ? (n >> 1) + 1 : n;
    double2 *twS = lds + 2 * BUF, *wwS = twS + NTW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NTW; i += PIPE_THREADS) twS[i] = tw[i];
    for (int i = tid; i < NWW; i += PIPE_THREADS) wwS[i] = ww[i];
    typename std::conditional<BIG, TwQuarter, const double2 *>::type twA;
    typename std::conditional<BIG, WwHalf, const double2 *>::type wwA;
    if constexpr (BIG) { twA = TwQuarter{twS, n >> 2}; wwA = WwHalf{wwS, n >> 1}; } else { twA = twS; wwA = wwS; }
    const unsigned ldsBase = (unsigned)(uintptr_t)lds;
    // element offset of a tile's first line (the 2 NP lines of a tile are consecutive in memory: nin % (2 NP) == 0)
    auto tile_base = [&](int tile) { return map.base((i64)tile << (lrows + 1)); };
    auto dma = [&](int tile, int b) {
        const double *g0 = src + tile_base(tile) + 2 * (lane & (NP - 1));
        const unsigned l0 = ldsBase + (unsigned)b * (unsigned)(BUF * 16);
#pragma unroll
        for (int i = 0; i < PIPE_ND; ++i) {
            const int c = wave * PIPE_ND + i;                         // piece: slots 64 c .. 64 c + 63
            const int p = padi(((c << 6) + lane) >> lrows);          // position held by this lane's slot
            const int k = (MODE == 1) ? p : ((p < (n >> 1)) ? 2 * p : 2 * (n - 1 - p) + 1);
            glds16(g0 + (i64)k * map.es, l0 + (unsigned)c * 1024u);
        }
    };
    // tile order: workgroup w runs on XCD w % 8; the tiles 2p and 2p + 1 (n = 2048, tiles 32 bytes wide: 4p .. 4p + 3) that
    // share every 128-byte line go to workgroups of one XCD at the same time
    const int w = blockIdx.x, stride = gridDim.x;     // gridDim.x is a multiple of 32
    int tile = BIG ? (((((w >> 5) << 3) + (w & 7)) << 2) | ((w >> 3) & 3))
                   : (((((w >> 4) << 3) + (w & 7)) << 1) | ((w >> 3) & 1));
    if (tile < nTiles) dma(tile, 0);
    if (tile + stride < nTiles) dma(tile + stride, 1);
    // the first tile has landed when only the second one's DMA is outstanding (vector-memory operations of a wave
    // complete in issue order)
    if (tile + stride < nTiles) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIPE_ND) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int it = 0; tile < nTiles; tile += stride, ++it) {
        const int b = it & 1;
        double2 *buf = lds + b * BUF;
        if (MODE == 1) idct_combine_tile<LG, lrows, PIPE_THREADS>(buf, tid, wwA);
        fft_tile_pipe<LG, lrows, PIPE_THREADS>(buf, tid, twA);
        {
            // item u of this thread: pair r0, k = k0 + u * (threads / NP)
            const int r0 = tid & (NP - 1), k0 = tid >> lrows;
            const double2 *rr = buf + r0;
            double *o = dst + tile_base(tile) + 2 * r0 + (i64)k0 * map.es;
            const i64 ostep = (i64)(PIPE_THREADS >> lrows) * map.es;
#pragma unroll
            for (int u = 0; u < 2 * PIPE_IT; ++u) {
                const int k = k0 + u * (PIPE_THREADS >> lrows);
                double2 v;
                if (MODE == 0) v = dct_post<lrows>(rr, k, n, LG, wwA);
                else v = rr[padi(bitrev(makhoul(k, n), LG)) << lrows];
                *(double2 *)o = v;
                o += ostep;
            }
        }
        // the next tile has landed when only this tile's stores are outstanding; one barrier then says both "every wave's
        // pieces of the next tile are in LDS" and "this buffer is drained"
        if (tile + stride < nTiles) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIPE_NS) : "memory");
        lds_barrier();
        if (tile + 2 * stride < nTiles) dma(tile + 2 * stride, b);
    }
}

// Fused t-axis solve (k_dct_strided<2>: forward DCT, division by the spectral kernel, inverse DCT), pipelined.  A tile =
// 2^lrows pairs of consecutive columns (y, y + 1) x all n time nodes, pair-interleaved in LDS like the strided kernel's;
// the eigenvalue tables CY, CT sit in LDS beside the twiddles (no ordinary global load inside the loop; CX of the tile's
// one x is a scalar load, which the vector memory counter does not see).  This pass is bound by its own chain of LDS /
// VALU phases (two transforms, seven barriers per tile), not by HBM: tiles of 2048 values and workgroups of 256 threads, so that TWO workgroups fit a CU
// and fill each other's gaps (tiles of 1024 values with 256 threads, three workgroups per CU: 2.71 instead of 2.48 ms for the
// whole solve at 1024 x 1024 x 128; with 128 threads: 2.48 -- measured, not kept).  Needs ny % (lines per tile) == 0: a tile has one x.
#define TS_THREADS 256
#define TS_LG_CPLX 11
#define TS_IT ((1 << (TS_LG_CPLX - 1)) / TS_THREADS)
template <int LG>
__global__ void __launch_bounds__(TS_THREADS) k_dct_tsolve_pipe(const double *__restrict__ src, double *__restrict__ dst,
                                                                 LineMap map, int nTiles, SolveArgs sa,
                                                                 const double2 *__restrict__ tw,
                                                                 const double2 *__restrict__ ww) {
    extern __shared__ double2 lds[];
    constexpr int n = 1 << LG;
    constexpr int lrows = TS_LG_CPLX - LG;
    constexpr int NP = 1 << lrows;
    constexpr int BUF = 1 << TS_LG_CPLX;              // pair-interleaved tile (see k_dct_strided_pipe), no padding
    constexpr int TS_ND = (1 << (TS_LG_CPLX - 6)) / (TS_THREADS / 64);     // DMA pieces per wave and tile
    static_assert(TS_ND == PIPE_ND && (1 << TS_LG_CPLX) / TS_THREADS == PIPE_NS, "wait counts are shared with the other pipes");
    double2 *twS = lds + 2 * BUF, *wwS = twS + (n >> 1);
    double *ctS = (double *)(wwS + n), *cyS = ctS + n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (n >> 1); i += TS_THREADS) twS[i] = tw[i];
    for (int i = tid; i < n; i += TS_THREADS) wwS[i] = ww[i];
    for (int i = tid; i < n; i += TS_THREADS) ctS[i] = sa.ct[i];
    for (int i = tid; i < (int)sa.ny; i += TS_THREADS) cyS[i] = sa.cy[i];
    const unsigned ldsBase = (unsigned)(uintptr_t)lds;
    // slot (padi(p) << lrows) + r holds position p of pair r; the forward transform is decimation-in-time, so position p
    // is element makhoul^-1(bitrev(p)) of the line
    auto dma = [&](int tile, int b) {
        const double *g0 = src + map.base((i64)tile << (lrows + 1)) + 2 * (lane & (NP - 1));   // pitched rows: a tile lies in one row
        const unsigned l0 = ldsBase + (unsigned)b * (unsigned)(BUF * 16);
#pragma unroll
        for (int i = 0; i < TS_ND; ++i) {
            const int c = wave * TS_ND + i;
            const int q = bitrev(padi(((c << 6) + lane) >> lrows), LG);
            const int k = (q < (n >> 1)) ? 2 * q : 2 * (n - 1 - q) + 1;
            glds16(g0 + (i64)k * map.es, l0 + (unsigned)c * 1024u);
        }
    };
    int tile = blockIdx.x;
    const int stride = gridDim.x;
    if (tile < nTiles) dma(tile, 0);
    if (tile + stride < nTiles) dma(tile + stride, 1);
    // the first tile has landed when only the second one's DMA is outstanding (vector-memory operations of a wave
    // complete in issue order)
    if (tile + stride < nTiles) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIPE_ND) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int it = 0; tile < nTiles; tile += stride, ++it) {
        const int b = it & 1;
        double2 *buf = lds + b * BUF;
        fft_tile_pipe_dit<LG, lrows, TS_THREADS>(buf, tid, twS);
        // spectrum in natural order: DCT post-processing, division, inverse pre-processing on the pair (k, n - k)
        {
            const i64 G0 = sa.line0 + ((i64)tile << (lrows + 1));     // first column of the tile: (y0, x0)
            const int y0 = (int)(G0 % sa.ny), x0 = (int)(G0 / sa.ny);
            const double ex = sload_f64(sa.cx + x0);
#pragma unroll
            for (int u = 0; u < TS_IT; ++u) {
                const int bb = tid + u * TS_THREADS;
                const int rr = bb & (NP - 1);
                double2 *r = buf + rr;
                const double ea = cyS[y0 + 2 * rr] + ex, eb = cyS[y0 + 2 * rr + 1] + ex;
                const int k = (bb >> lrows) + 1;                   // 1 .. n/2
                const int m = n - k;
                const int ik = padi(k) << lrows, im = padi(m) << lrows;
                const double2 vk = r[ik], vm = r[im];
                const double2 wk = wwS[k], wm = wwS[m];
                const double ar = 0.5 * (vk.x + vm.x), ai = 0.5 * (vk.y - vm.y);
                const double br = 0.5 * (vk.y + vm.y), bi = -0.5 * (vk.x - vm.x);
                const double ctk = ctS[k], ctm = ctS[m];
                double lak = ea + ctk, lbk = eb + ctk, lam = ea + ctm, lbm = eb + ctm;
                if (lak == 0.0) lak = 1.0;
                if (lbk == 0.0) lbk = 1.0;
                if (lam == 0.0) lam = 1.0;
                if (lbm == 0.0) lbm = 1.0;
                const double2 xk = make_double2((wk.x * ar - wk.y * ai) / (sa.kscale * lak),
                                                (wk.x * br - wk.y * bi) / (sa.kscale * lbk));
                const double2 xm = make_double2((wm.x * ar + wm.y * ai) / (sa.kscale * lam),
                                                (wm.x * br + wm.y * bi) / (sa.kscale * lbm));
                const double gar = 0.5 * (wk.x * xk.x + wm.x * xm.x), gai = 0.5 * (wk.y * xk.x - wm.y * xm.x);
                const double gbr = 0.5 * (wk.x * xk.y + wm.x * xm.y), gbi = 0.5 * (wk.y * xk.y - wm.y * xm.y);
                r[ik] = make_double2(gar - gbi, gai + gbr);
                if (m != k) {
                    const double har = 0.5 * (wm.x * xm.x + wk.x * xk.x), hai = 0.5 * (wm.y * xm.x - wk.y * xk.x);
                    const double hbr = 0.5 * (wm.x * xm.y + wk.x * xk.y), hbi = 0.5 * (wm.y * xm.y - wk.y * xk.y);
                    r[im] = make_double2(har - hbi, hai + hbr);
                }
            }
            if (tid < NP) {                                        // k = 0: V[0] is its own partner
                const int rr = tid;
                double2 *r = buf + rr;
                double la = (cyS[y0 + 2 * rr] + ex) + ctS[0];
                double lb2 = (cyS[y0 + 2 * rr + 1] + ex) + ctS[0];
                if (la == 0.0) la = 1.0;
                if (lb2 == 0.0) lb2 = 1.0;
                const double w0 = wwS[0].x;
                const double2 v0 = r[0];
                r[0] = make_double2(w0 * ((w0 * v0.x) / (sa.kscale * la)), w0 * ((w0 * v0.y) / (sa.kscale * lb2)));
            }
        }
        lds_barrier();
        fft_tile_pipe<LG, lrows, TS_THREADS>(buf, tid, twS);
        {
            const int r0 = tid & (NP - 1), k0 = tid >> lrows;
            const double2 *rr = buf + r0;
            double *o = dst + map.base((i64)tile << (lrows + 1)) + 2 * r0 + (i64)k0 * map.es;
            const i64 ostep = (i64)(TS_THREADS >> lrows) * map.es;
#pragma unroll
            for (int u = 0; u < 2 * TS_IT; ++u) {
                const int k = k0 + u * (TS_THREADS >> lrows);
                *(double2 *)o = rr[padi(bitrev(makhoul(k, n), LG)) << lrows];
                o += ostep;
            }
        }
        // the next tile has landed when only this tile's stores are outstanding; one barrier then says both "every wave's
        // pieces of the next tile are in LDS" and "this buffer is drained"
        if (tile + stride < nTiles) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIPE_NS) : "memory");
        lds_barrier();
        if (tile + 2 * stride < nTiles) dma(tile + 2 * stride, b);
    }
}

// Strided axes, workgroup-wide (forward / inverse only; 16-byte accesses: both lines of a pair per access).
template <int MODE /*0 fwd, 1 inv*/, int T>
__global__ void __launch_bounds__(T, 4) k_dct_strided_wg(const double *__restrict__ src,
                                                                       double *__restrict__ dst, LineMap map, int lg,
                                                                       int lp, const double2 *__restrict__ tw,
                                                                       const double2 *__restrict__ ww) {
    extern __shared__ double2 lds[];
    const int n = 1 << lg;
    const int rowStride = row_stride(n);
    const int npairs = 1 << lp;
    const int tid = threadIdx.x;
    const i64 L0 = xcd_tile(blockIdx.x, gridDim.x) << (lp + 1);
    const int r = tid & (npairs - 1);
    const i64 L = L0 + 2 * r;
    const bool ok = L < map.nLines;
    const i64 lb = ok ? map.base(L) : 0;
    const int kstep = T >> lp;
    for (int k0 = tid >> lp; k0 < n; k0 += kstep * DCT_BATCH) {
        double2 gv[DCT_BATCH];
#pragma unroll
        for (int u = 0; u < DCT_BATCH; ++u) {
            const int k = k0 + u * kstep;
            gv[u] = (ok && k < n) ? *(const double2 *)(src + lb + (i64)k * map.es) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < DCT_BATCH; ++u) {
            const int k = k0 + u * kstep;
            if (k < n) lds[r * rowStride + padi(MODE == 1 ? k : makhoul(k, n))] = gv[u];
        }
    }
    __syncthreads();
    if (MODE == 1) idct_combine_wg(lds, lp, lg, rowStride, tid, T, ww);
    fft_rows_wg(lds, lp, lg, rowStride, tid, T, tw);
    if (ok) {
        const double2 *rr = lds + r * rowStride;
        for (int k = tid >> lp; k < n; k += kstep) {
            double2 v;
            if (MODE == 0) v = dct_post(rr, k, n, lg, ww);
            else v = rr[padi(bitrev(makhoul(k, n), lg))];
            *(double2 *)(dst + lb + (i64)k * map.es) = v;
        }
    }
}

// Dense fallback (any length n): out_k = sum_j M[j*n + k] in_j.  A workgroup stages TL lines in
// LDS and produces the outputs k in [blockIdx.y * KC, +KC) of each of them.
//   axis 0 (lines contiguous):  thread <-> k, accumulating all TL lines per load of M (M is read once
//                               per TL lines);  KC = 256
//   strided axes (LINE_FAST):   thread <-> (line, k) with the line index fastest so that global
//                               accesses stay coalesced;  KC = 256 / TL
#define DENSE_TL 8
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
    if (LINE_FAST) {
        const int KC = DCT_THREADS / TL;
        const int l = threadIdx.x % TL, k = blockIdx.y * KC + threadIdx.x / TL;
        const i64 L = L0 + l;
        if (k >= n || L >= map.nLines) return;
        const double *in = tile + l * n;
        double acc = 0.0;
#pragma unroll 4
        for (int j = 0; j < n; ++j) acc += M[(i64)j * n + k] * in[j];
        dst[map.addr(L, k)] = acc;
    } else {
        const int k = blockIdx.y * DCT_THREADS + threadIdx.x;
        if (k >= n) return;
        double acc[DENSE_TL];
#pragma unroll
        for (int l = 0; l < DENSE_TL; ++l) acc[l] = 0.0;
#pragma unroll 2
        for (int j = 0; j < n; ++j) {
            const double m = M[(i64)j * n + k];
#pragma unroll
            for (int l = 0; l < DENSE_TL; ++l)
                if (l < TL) acc[l] += m * tile[l * n + j];
        }
#pragma unroll
        for (int l = 0; l < DENSE_TL; ++l)
            if (l < TL && L0 + l < map.nLines) dst[map.addr(L0 + l, k)] = acc[l];
    }
}


// ---------------------------------------------------------------------------------------------
// Dense lengths on the fp64 matrix cores.  Along a non-power-of-two axis the transform is the product of the
// n x n DCT matrix with the lines, out_l[k] = sum_j M[k][j] in_l[j]: v_mfma_f64_16x16x4_f64 tiles, a workgroup
// computes 64 outputs k of 128 lines, its four waves 64 k x 32 lines each (4 x 2 accumulator tiles), the j
// range streamed through LDS in double-buffered chunks of 8 (16: fewer barriers but half the resident workgroups,
// 27.0 vs 24.3 ms per 1025^2 x 129 Poisson solve) (next chunk's global loads in flight in registers
// during the MFMAs, one barrier per chunk).  M is staged as [j][k] (k contiguous, as stored); the lines as
// [j][line] on the strided axes (lines consecutive in memory) and as [line][j] on axis 0 (j contiguous in memory),
// so that global loads, LDS fragment reads (row strides 16 mod 32 doubles / 18 doubles: conflict-free) and the
// stores of the 16 x 16 result tiles (16 consecutive addresses per row) are all coalesced.
//   strided axes: D[k][line] = M . X      a = M fragment, b = line fragment
//   axis 0      : D[line][k] = X' . M'    a = line fragment, b = M fragment
// Operand / result lane maps of the f64 MFMA: a: A[lane & 15][lane >> 4], b: B[lane >> 4][lane & 15],
// d[r]: D[(lane >> 4) + 4 r][lane & 15]   (cdna_hip_programming.md, fragment layout).
// ---------------------------------------------------------------------------------------------
typedef double mf_double4 __attribute__((ext_vector_type(4)));
#define MF_KT 64
#define MF_LT 128
#define MF_MS (MF_KT + 16)
#define MF_XS (MF_LT + 16)      // strided axes: [j][line]

template <bool AXIS0, int MF_KC>
__global__ void __launch_bounds__(256) k_dct_mfma(const double *__restrict__ src, double *__restrict__ dst, LineMap map,
                                                   int n, const double *__restrict__ M) {
    constexpr int MF_XZ = MF_KC + 2;                  // axis 0: [line][j]
    constexpr int XSZ = (MF_KC * MF_XS > MF_LT * MF_XZ) ? MF_KC * MF_XS : MF_LT * MF_XZ;
    __shared__ double Ms[2][MF_KC * MF_MS];
    __shared__ double Xs[2][XSZ];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int k0 = blockIdx.x * MF_KT;
    const i64 L0 = (i64)blockIdx.y * MF_LT;
    const int li = lane & 15, lh = lane >> 4;
    mf_double4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (mf_double4){0.0, 0.0, 0.0, 0.0};
    // ---- global -> register staging of one chunk ----
    constexpr int MU = MF_KC / 4, XU = MF_KC / 2, ZSTEP = 256 / MF_KC;
    double mreg[MU], xreg[XU];
    const int m_kk = tid & 63, m_jj = tid >> 6;                  // matrix: element (jj + 4 u, kk)
    const bool m_ok = (k0 + m_kk) < n;
    // lines: strided axes -> thread owns line ll = tid & 127, rows jj = (tid >> 7) + 2 u
    //        axis 0       -> thread owns jj = tid % KC, lines ll = tid / KC + (256 / KC) u
    const int x_ll = AXIS0 ? (tid / MF_KC) : (tid & 127);
    const int x_jj = AXIS0 ? (tid % MF_KC) : (tid >> 7);
    i64 xbase = 0;
    bool x_ok = false;
    if (!AXIS0) {
        x_ok = (L0 + x_ll) < map.nLines;
        xbase = x_ok ? map.base(L0 + x_ll) : 0;
    }
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int j = j0 + m_jj + 4 * u;
            mreg[u] = (m_ok && j < n) ? M[(i64)j * n + k0 + m_kk] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (AXIS0) {
                const i64 L = L0 + x_ll + ZSTEP * u;
                const int j = j0 + x_jj;
                xreg[u] = (L < map.nLines && j < n) ? src[L * map.outerStride + j] : 0.0;     // axis 0: nin = 1
            } else {
                const int j = j0 + x_jj + 2 * u;
                xreg[u] = (x_ok && j < n) ? src[xbase + (i64)j * map.es] : 0.0;
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < MU; ++u) Ms[buf][(m_jj + 4 * u) * MF_MS + m_kk] = mreg[u];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (AXIS0) Xs[buf][(x_ll + ZSTEP * u) * MF_XZ + x_jj] = xreg[u];
            else Xs[buf][(x_jj + 2 * u) * MF_XS + x_ll] = xreg[u];
        }
    };
    const int nch = (n + MF_KC - 1) / MF_KC;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        if (c + 1 < nch) fetch((c + 1) * MF_KC);                  // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < MF_KC; kk += 4) {
            double mf[4], xf[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) mf[a] = Ms[buf][(kk + lh) * MF_MS + a * 16 + li];
#pragma unroll
            for (int b = 0; b < 2; ++b)
                xf[b] = AXIS0 ? Xs[buf][(wave * 32 + b * 16 + li) * MF_XZ + kk + lh]
                              : Xs[buf][(kk + lh) * MF_XS + wave * 32 + b * 16 + li];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = AXIS0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(xf[b], mf[a], acc[a][b], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_f64_16x16x4f64(mf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        if (c + 1 < nch) stash(buf ^ 1);
        __syncthreads();
    }
    // ---- store the 4 x 2 result tiles ----
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (AXIS0) {
                    const int k = k0 + a * 16 + li;
                    const i64 L = L0 + wave * 32 + b * 16 + lh + 4 * r;
                    if (k < n && L < map.nLines) dst[L * map.outerStride + k] = acc[a][b][r];
                } else {
                    const int k = k0 + a * 16 + lh + 4 * r;
                    const i64 L = L0 + wave * 32 + b * 16 + li;
                    if (k < n && L < map.nLines) dst[map.base(L) + (i64)k * map.es] = acc[a][b][r];
                }
            }
}


// Even / odd split of the dense transform: C[k][n-1-j] = (-1)^k C[k][j], so with h = floor(n/2)
//   forward:  X[2k']   = sum_{j<h} C[2k'][j]   (x[j] + x[n-1-j])  (+ C[2k'][h] x[h] for odd n)
//             X[2k'+1] = sum_{j<h} C[2k'+1][j] (x[j] - x[n-1-j])
//   inverse:  E[j] = sum_k' C[2k'][j] X[2k'], O[j] = sum_k' C[2k'+1][j] X[2k'+1],  x[j] = E + O, x[n-1-j] = E - O
// -- half the multiply-adds of the full product.  Same tiling as k_dct_mfma; the forward kernel forms the sums /
// differences while staging the lines (parity = blockIdx.z), the inverse kernel keeps two accumulator sets
// (E, O) per tile of j <= h and writes both mirror images.  Matrices (DctPlan): Ef, Of stored [j][k'] (k'
// contiguous), Ei, Oi stored [k'][j] (j contiguous) -- always [contraction index][output index].
struct SplitArgs {
    const double *Me, *Mo;
    int ne, no, h, njE;       // even / odd k counts, floor(n/2), h + (n odd ? 1 : 0)
    int xcd;                  // XCD-aware tile order
};

template <bool AXIS0, int MF_KC, bool INV>
__global__ void __launch_bounds__(256) k_dct_mfma_split(const double *__restrict__ src, double *__restrict__ dst,
                                                         LineMap map, int n, SplitArgs sp) {
    constexpr int MF_XZ = MF_KC + 2;
    constexpr int XSZ = (MF_KC * MF_XS > MF_LT * MF_XZ) ? MF_KC * MF_XS : MF_LT * MF_XZ;
    constexpr int NPH = INV ? 2 : 1;
    __shared__ double Ms[2][MF_KC * MF_MS];
    __shared__ double Xs[2][XSZ];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the output tiles that share a line tile run back to back on ONE XCD (xcd_tile), so that the lines come
    // from that XCD's L2 for all but the first of them
    const i64 P = sp.xcd ? xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y)
                         : (i64)blockIdx.x + (i64)gridDim.x * blockIdx.y;
    const int o0 = (int)(P % gridDim.x) * MF_KT;       // first output index of the tile (k' forward, j inverse)
    const i64 L0 = (P / gridDim.x) * MF_LT;
    const int li = lane & 15, lh = lane >> 4;
    mf_double4 acc[NPH][4][2];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[ph][a][b] = (mf_double4){0.0, 0.0, 0.0, 0.0};
    constexpr int MU = MF_KC / 4, XU = MF_KC / 2, ZSTEP = 256 / MF_KC;
    double mreg[MU], xreg[XU];
    const int m_oo = tid & 63, m_cc = tid >> 6;
    const int x_ll = AXIS0 ? (tid / MF_KC) : (tid & 127);
    const int x_cc = AXIS0 ? (tid % MF_KC) : (tid >> 7);
    i64 xbase = 0;
    bool x_ok = false;
    if (!AXIS0) {
        x_ok = (L0 + x_ll) < map.nLines;
        xbase = x_ok ? map.base(L0 + x_ll) : 0;
    }
    auto at = [&](i64 L, i64 lbase, int j) { return AXIS0 ? src[L * map.outerStride + j] : src[lbase + (i64)j * map.es]; };
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int par = INV ? ph : (int)blockIdx.z;               // 0: even part, 1: odd part
        const double *__restrict__ M = par ? sp.Mo : sp.Me;
        const int ld = INV ? (par ? sp.h : sp.njE) : (par ? sp.no : sp.ne);         // output indices the matrix holds
        const int ncontr = INV ? (par ? sp.no : sp.ne) : (par ? sp.h : sp.njE);
        const bool m_ok = (o0 + m_oo) < ld;
        // the contraction element c of a line: forward x[c] +- x[n-1-c] (the middle one alone), inverse x[2c + par]
        auto elem = [&](i64 L, i64 lbase, int c) {
            if (INV) return at(L, lbase, 2 * c + par);
            if (c >= sp.h) return at(L, lbase, c);                // c == h: middle element of an odd length (even part)
            const double u = at(L, lbase, c), v = at(L, lbase, n - 1 - c);
            return par ? u - v : u + v;
        };
        auto fetch = [&](int c0) {
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int c = c0 + m_cc + 4 * u;
                mreg[u] = (m_ok && c < ncontr) ? M[(i64)c * ld + o0 + m_oo] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                if (AXIS0) {
                    const i64 L = L0 + x_ll + ZSTEP * u;
                    const int c = c0 + x_cc;
                    xreg[u] = (L < map.nLines && c < ncontr) ? elem(L, 0, c) : 0.0;
                } else {
                    const int c = c0 + x_cc + 2 * u;
                    xreg[u] = (x_ok && c < ncontr) ? elem(0, xbase, c) : 0.0;
                }
            }
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int u = 0; u < MU; ++u) Ms[buf][(m_cc + 4 * u) * MF_MS + m_oo] = mreg[u];
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                if (AXIS0) Xs[buf][(x_ll + ZSTEP * u) * MF_XZ + x_cc] = xreg[u];
                else Xs[buf][(x_cc + 2 * u) * MF_XS + x_ll] = xreg[u];
            }
        };
        const int nch = (ncontr + MF_KC - 1) / MF_KC;
        fetch(0);
        stash(0);
        __syncthreads();
        for (int c = 0; c < nch; ++c) {
            const int buf = c & 1;
            if (c + 1 < nch) fetch((c + 1) * MF_KC);
#pragma unroll
            for (int kk = 0; kk < MF_KC; kk += 4) {
                double mf[4], xf[2];
#pragma unroll
                for (int a = 0; a < 4; ++a) mf[a] = Ms[buf][(kk + lh) * MF_MS + a * 16 + li];
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    xf[b] = AXIS0 ? Xs[buf][(wave * 32 + b * 16 + li) * MF_XZ + kk + lh]
                                  : Xs[buf][(kk + lh) * MF_XS + wave * 32 + b * 16 + li];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[ph][a][b] = AXIS0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(xf[b], mf[a], acc[ph][a][b], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f64_16x16x4f64(mf[a], xf[b], acc[ph][a][b], 0, 0, 0);
            }
            if (c + 1 < nch) stash(buf ^ 1);
            __syncthreads();
        }
    }
    // ---- store ----
    const int nout = INV ? sp.njE : (blockIdx.z ? sp.no : sp.ne);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + a * 16 + (AXIS0 ? li : lh + 4 * r);
                const i64 L = L0 + wave * 32 + b * 16 + (AXIS0 ? lh + 4 * r : li);
                if (o >= nout || L >= map.nLines) continue;
                const i64 lbase = AXIS0 ? L * map.outerStride : map.base(L);
                const i64 es = map.es;
                if (!INV) {
                    dst[lbase + (i64)(2 * o + (int)blockIdx.z) * es] = acc[0][a][b][r];
                } else {
                    const double E = acc[0][a][b][r], O = acc[NPH - 1][a][b][r];
                    dst[lbase + (i64)o * es] = E + O;
                    if (o < sp.h) dst[lbase + (i64)(n - 1 - o) * es] = E - O;
                }
            }
}

__global__ void __launch_bounds__(256) k_copy(const double *__restrict__ src, double *__restrict__ dst, i64 n) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) dst[i] = src[i];
}

#define DCT_LDS_BUDGET (72 * 1024)
#define DCT_LDS_MAX (160 * 1024)

static int floor_log2(i64 v) {
    int l = 0;
    while (((i64)2 << l) <= v) ++l;
    return l;
}

// log2 of the complex rows (pairs of lines) a workgroup stages: as many as fit the LDS budget
// with `nbuf` buffers, a power of two, at most 32 and no more than the problem has; for the
// strided axes the lines are consecutive in memory, so more rows = wider coalesced segments.
static int tile_log2_rows(int n, i64 nLines, int nbuf) {
    const size_t rowBytes = (size_t)row_stride(n) * sizeof(double2) * nbuf;
    i64 rows = (i64)(DCT_LDS_BUDGET / rowBytes);
    if (rows < 1) rows = 1;
    if (rows > 32) rows = 32;
    const i64 havePairs = (nLines + 1) / 2;
    int lp = floor_log2(rows);
    while (lp > 0 && ((i64)1 << lp) > havePairs) --lp;
    return lp;
}

template <class K>
static void allow_big_lds(K kernel) {
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DCT_LDS_MAX);
}

// Function attributes belong to the (function, device) pair: a process that drives several GPUs (dotsocp_create_multi)
// has to raise the dynamic-LDS limit once on EVERY device it launches on.  true = not done yet on the current device.
static std::mutex attr_mutex;
struct DeviceOnce {
    // `if (DeviceOnce once(mask); once) { raise the attributes }`: the lock is held while they are raised and the device's
    // bit is set only afterwards, so a second host thread can neither skip the block early nor launch in between
    std::unique_lock<std::mutex> lock;
    unsigned long long *mask;
    unsigned long long bit = 0;
    bool first = true;
    explicit DeviceOnce(unsigned long long &m) : lock(attr_mutex), mask(&m) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            bit = 1ull << dev;
            first = !(m & bit);
        }
    }
    ~DeviceOnce() { if (first && bit) *mask |= bit; }
    explicit operator bool() const { return first; }
};

bool dct_plan_is_pow2(const DctPlan *p) { return p->log2n > 0; }
bool dct_plan_has_tsolve(const DctPlan *p) {
    static const bool pfa_on = !(getenv("DOTSOCP_PFA") && atoi(getenv("DOTSOCP_PFA")) == 0);
    return p->log2n > 0 || (p->pfa && pfa_on);
}

static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

static bool dct_pipe_enabled() {
    static const bool on = !(getenv("DOTSOCP_DCT_PIPE") && atoi(getenv("DOTSOCP_DCT_PIPE")) == 0);
    return on;
}

static bool dct_wg_enabled() { return true; }

static int launch_strided(int mode, const DctPlan *p, const double *src, double *dst, const LineMap &map,
                          const SolveArgs &sa, hipStream_t st) {
    const int n = (int)p->n, lg = p->log2n;
    // t-axis solve (two transforms per row, in place): half the rows per workgroup so that twice as many
    // workgroups are resident
    const int lp = tile_log2_rows(n, map.nLines, mode == 2 ? 2 : 1);
    const size_t lds = ((size_t)1 << lp) * row_stride(n) * sizeof(double2);
    if (lds > DCT_LDS_MAX) {
        set_error("power-of-two DCT length %d does not fit the LDS (largest supported: 8192)", n);
        return DOTSOCP_EINVAL;
    }
    const i64 linesPerBlock = (i64)2 << lp;
    const unsigned blocks = (unsigned)((map.nLines + linesPerBlock - 1) / linesPerBlock);
    // one 16-byte access carries both lines of a pair when consecutive lines are adjacent, even-aligned doubles
    const bool vec = (map.nin % 2 == 0) && (map.outerStride % 2 == 0) && (map.es % 2 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    // fused t-axis solve, pipelined: eigenvalue tables in LDS, a tile = consecutive columns of one x
    // (rows of whole layers may be pitched: map.nin = ny lines per row, rows map.outerStride apart, time nodes map.es apart)
    if (dct_pipe_enabled() && dct_wg_enabled() && vec && mode == 2 && lg >= 5 && lg <= 10 &&
        ((map.outerStride == 0 && map.es == map.nin) || (map.nin == sa.ny && sa.line0 == 0))) {
        const i64 tileLines = ((i64)2 << TS_LG_CPLX) / n;
        const i64 nxv = sa.ny > 0 ? sa.nplane / sa.ny : 0;
        const size_t ldsPipe = (((size_t)2 << TS_LG_CPLX) + (size_t)(n >> 1) + (size_t)n) * sizeof(double2) +
                               ((size_t)n + (size_t)sa.ny) * sizeof(double);
        const int G = device_cus() * (ldsPipe <= DCT_LDS_MAX / 2 ? 2 : 1);     // two workgroups per CU when they fit
        if (tileLines >= 2 && sa.ny % tileLines == 0 && sa.line0 % tileLines == 0 && map.nLines % tileLines == 0 &&
            nxv * sa.ny == sa.nplane && ldsPipe <= DCT_LDS_MAX && map.nLines / tileLines >= 2 * (i64)G &&
            map.nLines / tileLines < (1ll << 30)) {
            const int nTiles = (int)(map.nLines / tileLines);
            static unsigned long long done_tp = 0;
            if (DeviceOnce once_(done_tp); once_) {
                allow_big_lds(k_dct_tsolve_pipe<5>); allow_big_lds(k_dct_tsolve_pipe<6>);
                allow_big_lds(k_dct_tsolve_pipe<7>); allow_big_lds(k_dct_tsolve_pipe<8>);
                allow_big_lds(k_dct_tsolve_pipe<9>); allow_big_lds(k_dct_tsolve_pipe<10>);
            }
#define TPIPE_LAUNCH(LGV)                                                                                              \
    DS_KLAUNCH((k_dct_tsolve_pipe<LGV>), dim3((unsigned)G), dim3(TS_THREADS), ldsPipe, st, src, dst, map, nTiles, sa, \
                       p->tw, p->ww)
            if (lg == 10) TPIPE_LAUNCH(10); else if (lg == 9) TPIPE_LAUNCH(9); else if (lg == 8) TPIPE_LAUNCH(8);
            else if (lg == 7) TPIPE_LAUNCH(7); else if (lg == 6) TPIPE_LAUNCH(6); else TPIPE_LAUNCH(5);
#undef TPIPE_LAUNCH
            DS_HIP(hipGetLastError());
            return 0;
        }
    }
    // pipelined persistent kernel (see k_dct_axis0_pipe): whole tiles of 4096 complex values, the chip filled twice over
    if (dct_pipe_enabled() && dct_wg_enabled() && vec && mode != 2 && lg >= 7 && lg <= 11) {
        const i64 tileLines = ((i64)2 << PIPE_LG_CPLX) / n;
        const int G = device_cus() & ~31;
        if (map.nin % tileLines == 0 && map.nLines % tileLines == 0 && G >= 32 && map.nLines / tileLines >= 2 * (i64)G &&
            map.nLines / tileLines < (1ll << 30)) {
            const int nTiles = (int)(map.nLines / tileLines);
            static unsigned long long done_sp = 0;
            if (DeviceOnce once_(done_sp); once_) {
                allow_big_lds(k_dct_strided_pipe<0, 7>); allow_big_lds(k_dct_strided_pipe<1, 7>);
                allow_big_lds(k_dct_strided_pipe<0, 8>); allow_big_lds(k_dct_strided_pipe<1, 8>);
                allow_big_lds(k_dct_strided_pipe<0, 9>); allow_big_lds(k_dct_strided_pipe<1, 9>);
                allow_big_lds(k_dct_strided_pipe<0, 10>); allow_big_lds(k_dct_strided_pipe<1, 10>);
                allow_big_lds(k_dct_strided_pipe<0, 11>); allow_big_lds(k_dct_strided_pipe<1, 11>);
            }
            // tables: n / 2 twiddles + n weights (2048-point lines: n / 4 + n / 2 + 1, TwQuarter / WwHalf)
            const size_t ntab = lg > 10 ? (size_t)(n >> 2) + (size_t)(n >> 1) + 1 : (size_t)(n >> 1) + (size_t)n;
            const size_t ldsPipe = (((size_t)2 << PIPE_LG_CPLX) + ntab) * sizeof(double2);
#define SPIPE_LAUNCH(M, LGV)                                                                                         \
    DS_KLAUNCH((k_dct_strided_pipe<M, LGV>), dim3((unsigned)G), dim3(PIPE_THREADS), ldsPipe, st, src, dst, map, \
                       nTiles, p->tw, p->ww)
            if (mode == 0) {
                if (lg == 11) SPIPE_LAUNCH(0, 11); else if (lg == 10) SPIPE_LAUNCH(0, 10); else if (lg == 9) SPIPE_LAUNCH(0, 9);
                else if (lg == 8) SPIPE_LAUNCH(0, 8); else SPIPE_LAUNCH(0, 7);
            } else {
                if (lg == 11) SPIPE_LAUNCH(1, 11); else if (lg == 10) SPIPE_LAUNCH(1, 10); else if (lg == 9) SPIPE_LAUNCH(1, 9);
                else if (lg == 8) SPIPE_LAUNCH(1, 8); else SPIPE_LAUNCH(1, 7);
            }
#undef SPIPE_LAUNCH
            DS_HIP(hipGetLastError());
            return 0;
        }
    }
    if (dct_wg_enabled() && vec && mode != 2 && ((i64)n << lp) >= 2 * DCT_WG_THREADS) {
        static unsigned long long done_wg = 0;
        if (DeviceOnce once_(done_wg); once_) {
            allow_big_lds(k_dct_strided_wg<0, 512>); allow_big_lds(k_dct_strided_wg<1, 512>);
            allow_big_lds(k_dct_strided_wg<0, 1024>); allow_big_lds(k_dct_strided_wg<1, 1024>);
        }
        // long lines: one workgroup of 1024 threads with the whole LDS (twice the rows) keeps 16 waves per CU
        // like two workgroups of 512 would, and widens the contiguous segment per line to 128 bytes
        const bool wide = true;
        const size_t lds2 = lds * 2;
        if (wide && lds2 <= DCT_LDS_MAX && lds2 > DCT_LDS_MAX / 2 && map.nLines >= ((i64)4 << lp)) {
            const int lp2 = lp + 1;
            const unsigned blocks2 = (unsigned)((map.nLines + ((i64)2 << lp2) - 1) / ((i64)2 << lp2));
            if (mode == 0)
                DS_KLAUNCH((k_dct_strided_wg<0, 1024>), dim3(blocks2), dim3(1024), lds2, st, src, dst, map, lg, lp2,
                                   p->tw, p->ww);
            else
                DS_KLAUNCH((k_dct_strided_wg<1, 1024>), dim3(blocks2), dim3(1024), lds2, st, src, dst, map, lg, lp2,
                                   p->tw, p->ww);
            DS_HIP(hipGetLastError());
            return 0;
        }
        if (mode == 0)
            DS_KLAUNCH((k_dct_strided_wg<0, 512>), dim3(blocks), dim3(DCT_WG_THREADS), lds, st, src, dst, map, lg, lp,
                               p->tw, p->ww);
        else
            DS_KLAUNCH((k_dct_strided_wg<1, 512>), dim3(blocks), dim3(DCT_WG_THREADS), lds, st, src, dst, map, lg, lp,
                               p->tw, p->ww);
        DS_HIP(hipGetLastError());
        return 0;
    }
#define LAUNCH_STRIDED(M, V)                                                                                   \
    do {                                                                                                       \
        static unsigned long long done = 0;                                                                    \
        if (DeviceOnce once_(done); once_) allow_big_lds(k_dct_strided<M, V>);                                    \
        DS_KLAUNCH((k_dct_strided<M, V>), dim3(blocks), dim3(DCT_THREADS), lds, st, src, dst, map, lg, \
                           lp, sa, p->tw, p->ww);                                                              \
    } while (0)
    if (mode == 0) { if (vec) LAUNCH_STRIDED(0, true); else LAUNCH_STRIDED(0, false); }
    else if (mode == 1) { if (vec) LAUNCH_STRIDED(1, true); else LAUNCH_STRIDED(1, false); }
    else { if (vec) LAUNCH_STRIDED(2, true); else LAUNCH_STRIDED(2, false); }
#undef LAUNCH_STRIDED
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_dct_t_solve(const DctPlan *p, const double *src, double *dst, i64 ny, i64 nplane, i64 line0, i64 nl,
                       i64 nt, double kscale, const double *cy, const double *cx, const double *ct, hipStream_t st,
                       i64 pitch0) {
    if (p->n != nt || !dct_plan_has_tsolve(p)) {
        set_error("fused t-axis solve needs a power-of-two nt or one of the prime-factor lengths");
        return DOTSOCP_EINVAL;
    }
    const bool pitched = pitch0 > ny;
    if (pitched && (line0 != 0 || nl != nplane || nplane % ny != 0)) {
        set_error("fused t-axis solve: pitched rows need whole layers");
        return DOTSOCP_EINVAL;
    }
    if (p->log2n <= 0) {
        if (nl <= 0) return 0;
        if (pitched) {
            // whole layers: nx rows of ny lines, rows pitch0 apart, time nodes pitch0 * nx apart
            const i64 nxv = nplane / ny;
            PfaSolveArgs a{kscale, cy, cx, ct, ny, 0, ny};
            return pfa_launch_strided(p->pfa, src, dst, ny, nxv, pitch0, pitch0 * nxv, pitch0, pitch0 * nxv, 2, &a, st);
        }
        PfaSolveArgs a{kscale, cy, cx, ct, ny, line0, 0};
        return pfa_launch_strided(p->pfa, src, dst, nl, 1, 0, nl, 0, nl, 2, &a, st);
    }
    LineMap map;
    map.nin = nl;
    map.outerStride = 0;
    map.nLines = nl;
    map.es = nl;
    if (pitched) {          // whole layers with pitched rows: line L = (y, x) = (L % ny, L / ny) starts at y + pitch0 * x
        map.nin = ny;
        map.outerStride = pitch0;
        map.es = pitch0 * (nplane / ny);
    }
    if (map.nLines <= 0) return 0;
    SolveArgs sa{ny, line0, nplane, kscale, cy, cx, ct};
    return launch_strided(2, p, src, dst, map, sa, st);
}

int launch_dct_axis(const DctPlan *p, const double *src, double *dst, i64 n0, i64 n1, i64 n2, int axis, int inverse,
                    hipStream_t st, i64 pitch0) {
    const i64 P0 = pitch0 > n0 ? pitch0 : n0;           // row pitch of both arrays
    const i64 dims[3] = {n0, n1, n2};
    const i64 n = dims[axis];
    const i64 total = n0 * n1 * n2;
    if (total <= 0) return 0;
    if (n != p->n) {
        set_error("dct plan length mismatch (%lld vs %lld)", (long long)n, (long long)p->n);
        return DOTSOCP_EINVAL;
    }
    if (n == 1) {
        const i64 all = P0 * n1 * n2;                     // pad entries travel with the rows
        if (src != dst)
            DS_KLAUNCH(k_copy, dim3(launch_blocks(all, 256, 1 << 14)), dim3(256), 0, st, src, dst, all);
        DS_HIP(hipGetLastError());
        return 0;
    }
    LineMap map;
    if (P0 == n0) {
        if (axis == 0) { map.nin = 1; map.outerStride = n; map.es = 1; }
        else if (axis == 1) { map.nin = n0; map.outerStride = n0 * n1; map.es = n0; }
        else { map.nin = n0 * n1; map.outerStride = 0; map.es = n0 * n1; }
    } else {
        // (power-of-two lengths: the axis-0 kernels take the line distance as an argument, the strided ones honour map.es)
        if (axis == 0) { map.nin = 1; map.outerStride = P0; map.es = 1; }
        else if (axis == 1) { map.nin = n0; map.outerStride = P0 * n1; map.es = P0; }
        else { map.nin = n0; map.outerStride = P0; map.es = P0 * n1; }
    }
    map.nLines = total / n;
    if (p->log2n > 0 && axis != 0) {
        SolveArgs sa{};
        return launch_strided(inverse ? 1 : 0, p, src, dst, map, sa, st);
    }
    if (p->log2n > 0) {
        // axis 0: each wave owns 2^lrw rows; a workgroup of 4 waves stages 4 * 2^lrw rows
        const int lg = p->log2n;
        int lp = tile_log2_rows((int)n, map.nLines, 1);
        const int lrw = lp >= 2 ? lp - 2 : 0;
        const size_t lds = ((size_t)DCT_WAVES << lrw) * row_stride((int)n) * sizeof(double2);
        if (lds > DCT_LDS_MAX) {
            set_error("power-of-two DCT length %lld does not fit the LDS (largest supported: 2048 along y, 8192 along x / t)",
                      (long long)n);
            return DOTSOCP_EINVAL;
        }
        const i64 linesPerBlock = (i64)(2 * DCT_WAVES) << lrw;
        const unsigned blocks = (unsigned)((map.nLines + linesPerBlock - 1) / linesPerBlock);
        static unsigned long long done = 0;
        if (DeviceOnce once_(done); once_) {
            allow_big_lds(k_dct_axis0<false>); allow_big_lds(k_dct_axis0<true>);
            allow_big_lds(k_dct_axis0_wg<false>); allow_big_lds(k_dct_axis0_wg<true>);
        }
        // pipelined persistent kernel: whole tiles of 8192 doubles, enough of them to fill the chip twice
        const i64 tileLines = ((i64)2 << PIPE_LG_CPLX) / n;
        if (dct_pipe_enabled() && dct_wg_enabled() && lg >= 7 && lg <= 11 && map.nLines % tileLines == 0 &&
            (((uintptr_t)src | (uintptr_t)dst) % 16 == 0)) {
            const i64 nTiles = map.nLines / tileLines;
            const int ncu = device_cus();
            if (nTiles >= 2 * (i64)ncu && nTiles < (1ll << 30)) {
                static unsigned long long done_pipe = 0;
                if (DeviceOnce once_(done_pipe); once_) {
                    allow_big_lds(k_dct_axis0_pipe<false, 7>); allow_big_lds(k_dct_axis0_pipe<true, 7>);
                    allow_big_lds(k_dct_axis0_pipe<false, 8>); allow_big_lds(k_dct_axis0_pipe<true, 8>);
                    allow_big_lds(k_dct_axis0_pipe<false, 9>); allow_big_lds(k_dct_axis0_pipe<true, 9>);
                    allow_big_lds(k_dct_axis0_pipe<false, 10>); allow_big_lds(k_dct_axis0_pipe<true, 10>);
                    allow_big_lds(k_dct_axis0_pipe<false, 11>); allow_big_lds(k_dct_axis0_pipe<true, 11>);
                }
                const size_t rs = (size_t)n + 1;
                const size_t ntab = lg > 10 ? (size_t)(n >> 2) + (size_t)(n >> 1) + 1 : (size_t)(n >> 1) + (size_t)n;
                const size_t ldsPipe = (2 * (rs << (PIPE_LG_CPLX - lg)) + ntab) * sizeof(double2);
#define PIPE_LAUNCH(INV, LGV)                                                                                        \
    DS_KLAUNCH((k_dct_axis0_pipe<INV, LGV>), dim3((unsigned)ncu), dim3(PIPE_THREADS), ldsPipe, st, src, dst, \
                       (int)nTiles, map.outerStride, p->tw, p->ww)
                if (inverse) {
                    if (lg == 11) PIPE_LAUNCH(true, 11); else if (lg == 10) PIPE_LAUNCH(true, 10); else if (lg == 9) PIPE_LAUNCH(true, 9);
                    else if (lg == 8) PIPE_LAUNCH(true, 8); else PIPE_LAUNCH(true, 7);
                } else {
                    if (lg == 11) PIPE_LAUNCH(false, 11); else if (lg == 10) PIPE_LAUNCH(false, 10); else if (lg == 9) PIPE_LAUNCH(false, 9);
                    else if (lg == 8) PIPE_LAUNCH(false, 8); else PIPE_LAUNCH(false, 7);
                }
#undef PIPE_LAUNCH
                DS_HIP(hipGetLastError());
                return 0;
            }
        }
        if (dct_wg_enabled() && ((n / 2) << lp) >= 2 * DCT_WG_THREADS) {
            // same rows per workgroup (4 << lrw complex rows), twice the threads, shared by all of them
            const int lrows = lrw + 2;
            const unsigned wblocks = (unsigned)((((map.nLines + 1) / 2) + ((i64)1 << lrows) - 1) >> lrows);
            if (inverse)
                DS_KLAUNCH(k_dct_axis0_wg<true>, dim3(wblocks), dim3(DCT_WG_THREADS), lds, st, src, dst, map.nLines,
                                   map.outerStride, lg, lrows, p->tw, p->ww);
            else
                DS_KLAUNCH(k_dct_axis0_wg<false>, dim3(wblocks), dim3(DCT_WG_THREADS), lds, st, src, dst, map.nLines,
                                   map.outerStride, lg, lrows, p->tw, p->ww);
            DS_HIP(hipGetLastError());
            return 0;
        }
        if (inverse)
            DS_KLAUNCH(k_dct_axis0<true>, dim3(blocks), dim3(DCT_THREADS), lds, st, src, dst, map.nLines, map.outerStride,
                               lg, lrw, p->tw, p->ww);
        else
            DS_KLAUNCH(k_dct_axis0<false>, dim3(blocks), dim3(DCT_THREADS), lds, st, src, dst, map.nLines, map.outerStride,
                               lg, lrw, p->tw, p->ww);
    } else {
        // DOTSOCP_PFA=0: the dense product also for the lengths that have a prime-factor transform
        static const bool pfa_on = !(getenv("DOTSOCP_PFA") && atoi(getenv("DOTSOCP_PFA")) == 0);
        if (p->pfa && pfa_on) {
            if (axis == 0) return pfa_launch_axis0(p->pfa, src, dst, map.nLines, P0, P0, inverse, st);
            if (axis == 1) return pfa_launch_strided(p->pfa, src, dst, n0, n2, P0 * n1, P0, P0 * n1, P0, inverse ? 1 : 0, nullptr, st);
            return pfa_launch_strided(p->pfa, src, dst, n0, n1, P0, P0 * n1, P0, P0 * n1, inverse ? 1 : 0, nullptr, st);
        }
        if (src == dst) {
            set_error("dense DCT path needs distinct src/dst");
            return DOTSOCP_EINVAL;
        }
        if (p->Ef && map.nLines >= 64) {
            const int mfma_xcd = 1;
            SplitArgs sp{inverse ? p->Ei : p->Ef, inverse ? p->Oi : p->Of, p->ne, p->no, p->h, p->njE, mfma_xcd};
            const unsigned lt = (unsigned)((map.nLines + MF_LT - 1) / MF_LT);
            if (inverse) {
                dim3 grid((unsigned)((p->njE + MF_KT - 1) / MF_KT), lt, 1);
                if (axis == 0) DS_KLAUNCH((k_dct_mfma_split<true, 8, true>), grid, dim3(256), 0, st, src, dst, map, (int)n, sp);
                else DS_KLAUNCH((k_dct_mfma_split<false, 8, true>), grid, dim3(256), 0, st, src, dst, map, (int)n, sp);
            } else {
                dim3 grid((unsigned)((p->ne + MF_KT - 1) / MF_KT), lt, 2);
                if (axis == 0) DS_KLAUNCH((k_dct_mfma_split<true, 8, false>), grid, dim3(256), 0, st, src, dst, map, (int)n, sp);
                else DS_KLAUNCH((k_dct_mfma_split<false, 8, false>), grid, dim3(256), 0, st, src, dst, map, (int)n, sp);
            }
            DS_HIP(hipGetLastError());
            return 0;
        }
        if (n >= 48 && map.nLines >= 64) {
            const double *Mm = inverse ? p->Cinv : p->Cfwd;
            dim3 grid((unsigned)((n + MF_KT - 1) / MF_KT), (unsigned)((map.nLines + MF_LT - 1) / MF_LT));
            if (axis == 0) DS_KLAUNCH((k_dct_mfma<true, 8>), grid, dim3(256), 0, st, src, dst, map, (int)n, Mm);
            else DS_KLAUNCH((k_dct_mfma<false, 8>), grid, dim3(256), 0, st, src, dst, map, (int)n, Mm);
            DS_HIP(hipGetLastError());
            return 0;
        }
        int TL = DENSE_TL;
        while (TL > 1 && (size_t)TL * n * sizeof(double) > 65536) TL >>= 1;
        while (TL > 1 && map.nLines < (i64)TL * 64) TL >>= 1;      // few lines: favour more workgroups
        const size_t lds = (size_t)TL * n * sizeof(double);
        const unsigned bx = (unsigned)((map.nLines + TL - 1) / TL);
        const double *M = inverse ? p->Cinv : p->Cfwd;
        if (axis != 0) {
            const int KC = DCT_THREADS / TL;
            DS_KLAUNCH((k_dct_dense<true>), dim3(bx, (unsigned)((n + KC - 1) / KC)), dim3(DCT_THREADS), lds, st, src,
                               dst, map, (int)n, TL, M);
        } else {
            DS_KLAUNCH((k_dct_dense<false>), dim3(bx, (unsigned)((n + DCT_THREADS - 1) / DCT_THREADS)),
                               dim3(DCT_THREADS), lds, st, src, dst, map, (int)n, TL, M);
        }
    }
    DS_HIP(hipGetLastError());
    return 0;
}

// data ./= kscale * ((CY[ky] + CX[kx]) + CT[kt]) with the zero eigenvalue replaced by 1
// (initialize_FFTkernel.m:6-15, solver_socp_inPALM.m:96).
__global__ void __launch_bounds__(256) k_spectral_divide(double *__restrict__ data, i64 ny, i64 py, i64 nxl, i64 nt, i64 x0,
                                                          double kscale, const double *__restrict__ cy,
                                                          const double *__restrict__ cx,
                                                          const double *__restrict__ ct) {
    const i64 y = (i64)blockIdx.x * 64 + threadIdx.x;
    const i64 x = (i64)blockIdx.y * 4 + threadIdx.y;
    const i64 t = blockIdx.z;
    if (y >= ny || x >= nxl) return;
    double lam = (cy[y] + cx[x0 + x]) + ct[t];
    if (lam == 0.0) lam = 1.0;
    const i64 i = y + py * (x + nxl * t);
    data[i] = data[i] / (kscale * lam);
}

__global__ void __launch_bounds__(256) k_spectral_divide_pencil(double *__restrict__ data, i64 ny, i64 line0, i64 nl,
                                                                 i64 nt, double kscale, const double *__restrict__ cy,
                                                                 const double *__restrict__ cx,
                                                                 const double *__restrict__ ct) {
    const i64 L = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 t = blockIdx.y;
    if (L >= nl) return;
    const i64 G = line0 + L;
    double lam = (cy[G % ny] + cx[G / ny]) + ct[t];
    if (lam == 0.0) lam = 1.0;
    data[L + nl * t] = data[L + nl * t] / (kscale * lam);
}

int launch_spectral_divide_pencil(double *data, i64 ny, i64 nplane, i64 line0, i64 nl, i64 nt, double kscale,
                                  const double *cy, const double *cx, const double *ct, hipStream_t st) {
    (void)nplane;
    if (nl * nt <= 0) return 0;
    dim3 grid((unsigned)((nl + 255) / 256), (unsigned)nt);
    DS_KLAUNCH(k_spectral_divide_pencil, grid, dim3(256), 0, st, data, ny, line0, nl, nt, kscale, cy, cx, ct);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_spectral_divide(double *data, i64 ny, i64 nx, i64 nt, i64 x0, i64 nxl, double kscale, const double *cy,
                           const double *cx, const double *ct, hipStream_t st, i64 pitch0) {
    (void)nx;
    if (ny * nxl * nt <= 0) return 0;
    dim3 grid((unsigned)((ny + 63) / 64), (unsigned)((nxl + 3) / 4), (unsigned)nt);
    DS_KLAUNCH(k_spectral_divide, grid, dim3(64, 4), 0, st, data, ny, pitch0 > ny ? pitch0 : ny, nxl, nt, x0, kscale, cy, cx, ct);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
