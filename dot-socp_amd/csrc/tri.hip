// The t axis of the Poisson solve as tridiagonal systems: across time slabs WITHOUT the slab <-> pencil transposes
// (option B), and -- the same elimination inside one workgroup -- on the single slab (k_tsolve_*).
//
// After the y and x transforms every (ky, kx) mode is an independent system along t,
//     D^2 ((CY[ky] + CX[kx]) I + T) phi = r ,   T = (nt-1)^2 tridiag(-1, [1, 2, ..., 2, 1], -1)
// -- the Neumann matrix whose eigen-decomposition is the t-axis DCT with the eigenvalues CT of
// initialize_FFTkernel.m:6-15, so its solution IS idct_t(dct_t(r) ./ kernel) (to rounding; zero mode below).
// With the time axis cut into slabs the system is solved by partitioning (Wang / SPIKE):
//   k_tri_local   every slab, per mode: first / last entry of A_p^{-1} g_p (two eliminations from the two ends;
//                 A_p = the slab's diagonal block, g = r / (D^2 (nt-1)^2))                 -> 2 numbers per mode
//   exchange A    the 2 numbers of every mode go to the rank that owns the mode (pencil ranges of the columns)
//   k_tri_reduced owner, per mode: the 2P interface values from the block-bidiagonal reduced system
//   exchange B    every slab gets the neighbours' interface values of its modes back
//   k_tri_final   every slab, per mode: A_p x = g_p + e_first x_left + e_last x_right (Thomas)
// Volume per rank and solve: 4 numbers per mode instead of 2 ntl (two transposes): 8x less at 8 slabs of 16.
// The (0, 0) mode is singular (kernel == 0 -> 1, initialize_FFTkernel.m:15): its single line of nt values
// travels whole to the owner of column 0, which solves T x = g - mean(g) by recurrence, removes the mean and adds
// (nt-1)^2 mean(g) -- the k = 0 coefficient divided by D^2 * 1.
#include "device_utils.h"
#include "kernels.h"

#include <cstdint>
#include <cstdlib>
#include <mutex>

namespace dotsocp {

struct TriGeom {
    i64 ny, plane, ntl;       // local slab: ntl time nodes; ny = row pitch (a pad entry of a row is a mode of its own, all zeros)
    int first, last;          // slab holds global t = 0 / t = nt-1
    double beta;              // (nt-1)^2
    double kscale;            // D^2
    const double *cy, *cx;
    PencilCuts pc;            // owner ranges of the modes (columns)
};

__device__ __forceinline__ double tri_aprime(const TriGeom &g, i64 m) { return (g.cy[m % g.ny] + g.cx[m / g.ny]) / g.beta; }

__device__ __forceinline__ int tri_owner(const PencilCuts &pc, i64 m, i64 plane) {
    int j = (int)((m * pc.world) / plane);
    while (j > 0 && m < pc.cut[j]) --j;
    while (j < pc.world - 1 && m >= pc.cut[j + 1]) ++j;
    return j;
}

// message to owner j starts at 2 cut[j] + TRI_EXTRA j and holds [first values | last values | TRI_EXTRA extras]
__device__ __forceinline__ i64 tri_msg_off(const PencilCuts &pc, int j) { return 2 * pc.cut[j] + (i64)TRI_EXTRA * j; }

// ---- division-free eliminations (round 4) ----
// The pivots of a block depend on the mode and the row only, and have a closed form.  With x = 1 + a'/2 = cosh(theta),
// r = e^theta = x + sqrt(x^2 - 1), rho = 1 / r, the elimination from a block's first row gives
//     piv_t = r N_{t+1} / N_t ,   N_t = 1 - rho^(2t+2)  (row 0 couples to a neighbour slab: delta_0 = a' + 2)
//                                 N_t = 1 + rho^(2t+1)  (row 0 is the global first / last row: delta_0 = a' + 1)
// (both satisfy N_{t+1} = (1 - rho^2) + rho^2 N_t -- a recurrence of positive terms, no cancellation), and a global
// boundary row at the END of the sweep has piv = r N_n / N_{n-1} - 1.  In the scaled variable D_t = d_t N_t the sweep
// d_t = g_t + d_{t-1} / piv_{t-1} becomes  D_t = g_t N_t + rho D_{t-1}:  no division per row (the kernels of round 2
// spent 2 n dependent IEEE divisions per mode here and were bound by them: 43 + 57 us on a slab of 16 layers that
// streams in 27 + 54, 530 + 630 us on 64 layers).  The last unknown is rho D_{n-1} / N_n (boundary end:
// D_{n-1} / (r N_n - N_{n-1})).  The sweep from the other end is the same recurrence on the reversed column; as a
// weighted sum, sum_t rho^t N'_{n-1-t} g_t, it runs in the same ascending pass over the column.
struct TriCoef {
    double rho, rho2, r, r2, n0d;      // n0d = 1 - rho^2, formed without cancellation
};
__device__ __forceinline__ TriCoef tri_coef(double ap) {
    TriCoef c;
    const double s = sqrt(ap * (1.0 + 0.25 * ap));     // sqrt(x^2 - 1)
    const double rm1 = 0.5 * ap + s;                    // r - 1
    c.r = 1.0 + rm1;
    c.rho = 1.0 / c.r;
    c.rho2 = c.rho * c.rho;
    c.r2 = c.r * c.r;
    c.n0d = (rm1 * c.rho) * (1.0 + c.rho);              // (1 - rho) (1 + rho)
    return c;
}
__device__ __forceinline__ double tri_n0(const TriCoef &c, bool bnd) { return bnd ? 1.0 + c.rho : c.n0d; }
// x / n for n in (0, 2]: reciprocal seed, two Newton steps, one residual correction (the result of the IEEE sequence to
// the last bit or one off it, at about half its instructions)
__device__ __forceinline__ double tri_div(double x, double n) {
    double r = __builtin_amdgcn_rcp(n);
    double e = __builtin_fma(-n, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-n, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    return __builtin_fma(__builtin_fma(-n, q, x), r, q);
}
// N_0, N_{n-1}, N_n of the sweep over n rows that starts on a boundary row (bnd: N_j = 1 + rho^(2j+1)) or not
// (N_j = 1 - rho^(2j+2)); rn1 = rho^(n-1).  s, pe: N_j = 1 + s pe rho^(2j).
struct TriEnds {
    double N0, N1, Nn, s, pe;
};
__device__ __forceinline__ TriEnds tri_ends(const TriCoef &c, bool bnd, int n, double rn1) {
    TriEnds e;
    e.s = bnd ? 1.0 : -1.0;
    e.pe = bnd ? c.rho : c.rho2;
    e.N0 = tri_n0(c, bnd);
    e.N1 = (n == 1) ? e.N0 : 1.0 + e.s * ((rn1 * rn1) * e.pe);
    e.Nn = c.n0d + c.rho2 * e.N1;
    return e;
}
__device__ __forceinline__ double tri_powi(double x, int k) {
    double r = 1.0;
    while (k > 0) {
        if (k & 1) r *= x;
        x *= x;
        k >>= 1;
    }
    return r;
}
// the last unknown of a sweep over n rows from its D_{n-1}; bnd: the sweep ENDS on a global boundary row
__device__ __forceinline__ double tri_last(const TriCoef &c, double D, double N1, double Nn, bool bnd) {
    return bnd ? tri_div(D, c.r * Nn - N1) : tri_div(c.rho * D, Nn);
}
// Both sweeps of a column come out of two running sums, H_t = g_t + rho H_{t-1} and G_t = sum_{s <= t} rho^s g_s:
//     D_t = sum_{s <= t} rho^(t-s) N_s g_s = H_t + s_f pe_f rho^t G_t            (front sweep, kept per row by k_tri_final)
//     F   = sum_t rho^t N'_{n-1-t} g_t     = G_{n-1} + s_b pe_b rho^(n-1) H_{n-1}  (back sweep, as a weighted sum)
// -- no power ever has to be walked downwards from a value that may have underflowed.
// first / last entries of A_p^-1 e_first (vf, vl) and A_p^-1 e_last (wf, wl) of a block of n rows:
// prod_{s < n-1} 1 / piv_s = rho^(n-1) N_0 / N_{n-1}, 1 / piv_{n-1} = rho N_{n-1} / N_n (boundary end: N_{n-1} / (r N_n - N_{n-1}))
struct TriSpike {
    double vf, vl, wf, wl;
};
__device__ __forceinline__ TriSpike tri_spike(const TriCoef &c, int n, bool first, bool last) {
    const double rn1 = tri_powi(c.rho, n - 1);
    const TriEnds f = tri_ends(c, first, n, rn1), b = tri_ends(c, last, n, rn1);
    TriSpike k;
    if (last) {
        const double den = c.r * f.Nn - f.N1;
        k.vl = tri_div(rn1 * f.N0, den);
        k.wl = tri_div(f.N1, den);
    } else {
        k.vl = tri_div((rn1 * c.rho) * f.N0, f.Nn);
        k.wl = tri_div(c.rho * f.N1, f.Nn);
    }
    if (first) {
        const double den = c.r * b.Nn - b.N1;
        k.wf = tri_div(rn1 * b.N0, den);
        k.vf = tri_div(b.N1, den);
    } else {
        k.wf = tri_div((rn1 * c.rho) * b.N0, b.Nn);
        k.vf = tri_div(c.rho * b.N1, b.Nn);
    }
    if (first) k.vf = k.vl = 0.0;                 // no left / right neighbour
    if (last) k.wf = k.wl = 0.0;
    return k;
}

// One ascending pass over the column of a mode: both eliminations at once, nothing kept (any slab length)
__global__ void __launch_bounds__(256) k_tri_local(TriGeom g, const double *__restrict__ r, double *__restrict__ send) {
    const i64 m = (i64)blockIdx.x * 256 + threadIdx.x;
    if (m >= g.plane) return;
    const double sc = 1.0 / (g.kscale * g.beta);
    const int n = (int)g.ntl;
    const int j = tri_owner(g.pc, m, g.plane);
    const i64 off = tri_msg_off(g.pc, j), w = g.pc.cut[j + 1] - g.pc.cut[j];
    if (m == 0) {                                 // the singular mode travels whole
        for (int t = 0; t < n; ++t) send[off + 2 * w + t] = r[g.plane * t] * sc;
        send[off] = 0.0;
        send[off + w] = 0.0;
        return;
    }
    const TriCoef c = tri_coef(tri_aprime(g, m));
    const bool first = g.first != 0, last = g.last != 0;
    double H = 0.0, G = 0.0, pw = 1.0;
    constexpr int U = 8;
    for (int t0 = 0; t0 < n; t0 += U) {
        double gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) gv[u] = (t0 + u < n) ? r[m + g.plane * (t0 + u)] * sc : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u < n) {
                H = gv[u] + c.rho * H;
                G += pw * gv[u];
                if (t0 + u + 1 < n) pw *= c.rho;           // ends as rho^(n-1)
            }
        }
    }
    const TriEnds f = tri_ends(c, first, n, pw), b = tri_ends(c, last, n, pw);
    const double D = H + ((f.s * f.pe) * pw) * G, F = G + ((b.s * b.pe) * pw) * H;
    send[off + (m - g.pc.cut[j])] = tri_last(c, F, b.N1, b.Nn, first);        // first entry of A^-1 g
    send[off + w + (m - g.pc.cut[j])] = tri_last(c, D, f.N1, f.Nn, last);     // last entry
}

struct TriReduced {
    int P, rank;              // slabs, this owner
    i64 l0, nl;               // owned modes [l0, l0 + nl)
    i64 nt;
    i64 slab_n[DS_MAX_WORLD]; // time nodes of every slab
    // one slab per process: the message of slab `own` (this rank's) does not travel -- it is read where k_tri_local wrote it
    // (own_recv, inside tri_send) and its answer is written where k_tri_final reads it (own_back, inside tri_brecv)
    int own;                  // -1: every message lies in recv / back
    const double *own_recv;
    double *own_back;
};

// PMAX: compile-time bound on the number of slabs -- the four sweep arrays are then fully unrolled and live in registers
// (a run-time bound of DS_MAX_WORLD puts 2 KB per thread into scratch: the generic instance, used above 16 slabs only)
template <int PMAX>
__global__ void __launch_bounds__(128) k_tri_reduced(TriGeom g, TriReduced q, const double *__restrict__ recv,
                                                      double *__restrict__ back, double *__restrict__ zero_work) {
    const i64 i = (i64)blockIdx.x * 128 + threadIdx.x;
    if (i >= q.nl) return;
    const i64 m = q.l0 + i;
    const i64 stride = 2 * q.nl + TRI_EXTRA;      // one message per slab
    auto msg_in = [&](int p) { return (p == q.own) ? q.own_recv : recv + p * stride; };
    auto msg_out = [&](int p) { return (p == q.own) ? q.own_back : back + p * stride; };
    if (m == 0) {
        // T x = g - mean(g) by recurrence from x_0 = 0, then zero mean, plus beta * mean(g)
        double sum = 0.0;
        i64 tg = 0;
        for (int p = 0; p < q.P; ++p)
            for (i64 t = 0; t < q.slab_n[p]; ++t, ++tg) {
                const double v = msg_in(p)[2 * q.nl + t];
                zero_work[tg] = v;
                sum += v;
            }
        const double gbar = sum / (double)q.nt;
        double xm = 0.0, xc = 0.0, acc = 0.0;      // x_{t-1}, x_t
        for (i64 t = 0; t < q.nt; ++t) {
            const double gt = zero_work[t] - gbar;
            zero_work[t] = xc;
            acc += xc;
            const double xn = (t == 0) ? xc - gt : 2.0 * xc - xm - gt;
            xm = xc;
            xc = xn;
        }
        const double shift = g.beta * gbar - acc / (double)q.nt;
        tg = 0;
        for (int p = 0; p < q.P; ++p)
            for (i64 t = 0; t < q.slab_n[p]; ++t, ++tg) msg_out(p)[2 * q.nl + t] = zero_work[tg] + shift;
        for (int p = 0; p < q.P; ++p) { msg_out(p)[i] = 0.0; msg_out(p)[q.nl + i] = 0.0; }
        return;
    }
    const TriCoef c = tri_coef(tri_aprime(g, m));
    double A[PMAX], B[PMAX], al[PMAX], ga[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        if (p >= q.P) break;
        const TriSpike k = tri_spike(c, (int)q.slab_n[p], p == 0, p == q.P - 1);
        const double vf = k.vf, vl = k.vl, wf = k.wf, wl = k.wl;
        const double *mi = msg_in(p);
        const double Gf = mi[i], Gl = mi[q.nl + i];
        // unknowns F_p (first value of slab p), L_p (last value):  F_p = Gf + vf L_{p-1} + wf F_{p+1},  L_p = Gl + vl L_{p-1} + wl F_{p+1}
        // sweep: L_{p-1} = al + ga F_p  ->  F_p = A + B F_{p+1},  L_p = al' + ga' F_{p+1}
        if (p == 0) {
            A[0] = Gf; B[0] = wf; al[0] = Gl; ga[0] = wl;
        } else {
            const double den = 1.0 - vf * ga[p - 1];
            A[p] = (Gf + vf * al[p - 1]) / den;
            B[p] = wf / den;
            al[p] = Gl + vl * (al[p - 1] + ga[p - 1] * A[p]);
            ga[p] = wl + vl * ga[p - 1] * B[p];
        }
    }
    // back substitution; slab p needs L_{p-1} and F_{p+1}
    double Fnext = 0.0;                            // F_{p+1}
#pragma unroll
    for (int p = PMAX - 1; p >= 0; --p) {
        if (p >= q.P) continue;
        const double F = A[p] + B[p] * Fnext;
        const double Lprev = (p > 0) ? al[p - 1] + ga[p - 1] * F : 0.0;
        double *mo = msg_out(p);
        mo[i] = Lprev;
        mo[q.nl + i] = Fnext;
        Fnext = F;
    }
}

// A_p x = g + e_first x_left + e_last x_right: forward D_t from the two running sums (no division), backward
// x_t = rho (D_t + N_t x_{t+1}) / N_{t+1} (one fast division per row), N_t = 1 + s pe rho^(2t) from rho^t walked back up.
// Generic slab length: D_t is parked in x between the sweeps (two reads and two writes of the slab).
__global__ void __launch_bounds__(256) k_tri_final(TriGeom g, const double *__restrict__ back, double *__restrict__ x) {
    const i64 m = (i64)blockIdx.x * 256 + threadIdx.x;
    if (m >= g.plane) return;
    const int n = (int)g.ntl;
    const int j = tri_owner(g.pc, m, g.plane);
    const i64 off = tri_msg_off(g.pc, j), w = g.pc.cut[j + 1] - g.pc.cut[j];
    if (m == 0) {
        for (int t = 0; t < n; ++t) x[g.plane * t] = back[off + 2 * w + t];
        return;
    }
    const TriCoef c = tri_coef(tri_aprime(g, m));
    const bool first = g.first != 0, last = g.last != 0;
    const double sc = 1.0 / (g.kscale * g.beta);
    const double xl = back[off + (m - g.pc.cut[j])], xr = back[off + w + (m - g.pc.cut[j])];
    const double spe = first ? c.rho : -c.rho2;                     // s pe of the front sequence
    double H = 0.0, G = 0.0, pw = 1.0, D = 0.0;
    for (int t = 0; t < n; ++t) {
        double gt = x[m + g.plane * t] * sc;
        if (t == 0) gt += xl;
        if (t == n - 1) gt += xr;
        H = gt + c.rho * H;
        G += pw * gt;
        D = H + (spe * pw) * G;
        x[m + g.plane * t] = D;
        if (t + 1 < n) pw *= c.rho;
    }
    const TriEnds f = tri_ends(c, first, n, pw);
    double xn = tri_last(c, D, f.N1, f.Nn, last);
    x[m + g.plane * (n - 1)] = xn;
    double Nt1 = f.N1;
    for (int t = n - 2; t >= 0; --t) {
        pw *= c.r;
        const double Nt = (t == 0) ? f.N0 : 1.0 + spe * (pw * pw);
        xn = tri_div(c.rho * (x[m + g.plane * t] + Nt * xn), Nt1);
        x[m + g.plane * t] = xn;
        Nt1 = Nt;
    }
}

// Register-resident flavour for short slabs (ntl <= NTL): the column of a mode is read ONCE into registers, both sweeps
// run there, and the result is written once -- one read and one write of the slab.
template <int NTL>
__global__ void __launch_bounds__(256) k_tri_final_reg(TriGeom g, const double *__restrict__ back, double *__restrict__ x) {
    const i64 m = (i64)blockIdx.x * 256 + threadIdx.x;
    if (m >= g.plane) return;
    const int n = (int)g.ntl;
    const int j = tri_owner(g.pc, m, g.plane);
    const i64 off = tri_msg_off(g.pc, j), w = g.pc.cut[j + 1] - g.pc.cut[j];
    if (m == 0) {
        for (int t = 0; t < n; ++t) x[g.plane * t] = back[off + 2 * w + t];
        return;
    }
    const TriCoef c = tri_coef(tri_aprime(g, m));
    const bool first = g.first != 0, last = g.last != 0;
    const double sc = 1.0 / (g.kscale * g.beta);
    const double xl = back[off + (m - g.pc.cut[j])], xr = back[off + w + (m - g.pc.cut[j])];
    double X[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) X[t] = (t < n) ? x[m + g.plane * t] : 0.0;
    const double spe = first ? c.rho : -c.rho2;
    double H = 0.0, G = 0.0, pw = 1.0, D = 0.0;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        if (t < n) {
            double gt = X[t] * sc;
            if (t == 0) gt += xl;
            if (t == n - 1) gt += xr;
            H = gt + c.rho * H;
            G += pw * gt;
            D = H + (spe * pw) * G;
            X[t] = D;
            if (t + 1 < n) pw *= c.rho;
        }
    }
    const TriEnds f = tri_ends(c, first, n, pw);
    double xn = tri_last(c, D, f.N1, f.Nn, last);
    double Nt1 = f.N1;
#pragma unroll
    for (int t = NTL - 1; t >= 0; --t) {
        if (t == n - 1) {
            X[t] = xn;
        } else if (t < n - 1) {
            pw *= c.r;
            const double Nt = (t == 0) ? f.N0 : 1.0 + spe * (pw * pw);
            xn = tri_div(c.rho * (X[t] + Nt * xn), Nt1);
            X[t] = xn;
            Nt1 = Nt;
        }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t)
        if (t < n) x[m + g.plane * t] = X[t];
}

static int tri_reg_width(i64 ntl) {
    return ntl <= 16 ? 16 : (ntl <= 32 ? 32 : (ntl <= 64 ? 64 : 0));
}

static TriGeom make_geom(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, const PencilCuts &pc) {
    TriGeom t{};
    t.ny = g.py; t.plane = g.plane; t.ntl = g.ntl;
    t.first = g.first ? 1 : 0; t.last = g.last ? 1 : 0;
    t.beta = (double)(nt - 1) * (double)(nt - 1);
    t.kscale = kscale;
    t.cy = cy; t.cx = cx;
    t.pc = pc;
    return t;
}

int launch_tri_local(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, const PencilCuts &pc,
                     const double *r, double *send, hipStream_t st) {
    const TriGeom t = make_geom(g, nt, kscale, cy, cx, pc);
    const dim3 grid((unsigned)((g.plane + 255) / 256));
    DS_KLAUNCH(k_tri_local, grid, dim3(256), 0, st, t, r, send);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_tri_reduced(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, const PencilCuts &pc,
                       int rank, i64 l0, i64 nl, const i64 *slab_n, const double *recv, double *back, double *zero_work,
                       hipStream_t st, const double *own_recv, double *own_back) {
    if (nl <= 0) return 0;
    const TriGeom t = make_geom(g, nt, kscale, cy, cx, pc);
    TriReduced q{};
    q.P = pc.world; q.rank = rank; q.l0 = l0; q.nl = nl; q.nt = nt;
    q.own = (own_recv && own_back) ? rank : -1;
    q.own_recv = own_recv; q.own_back = own_back;
    for (int p = 0; p < pc.world; ++p) q.slab_n[p] = slab_n[p];
    const dim3 grid((unsigned)((nl + 127) / 128));
    if (pc.world <= 4) DS_KLAUNCH(k_tri_reduced<4>, grid, dim3(128), 0, st, t, q, recv, back, zero_work);
    else if (pc.world <= 8) DS_KLAUNCH(k_tri_reduced<8>, grid, dim3(128), 0, st, t, q, recv, back, zero_work);
    else if (pc.world <= 16) DS_KLAUNCH(k_tri_reduced<16>, grid, dim3(128), 0, st, t, q, recv, back, zero_work);
    else DS_KLAUNCH(k_tri_reduced<DS_MAX_WORLD>, grid, dim3(128), 0, st, t, q, recv, back, zero_work);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_tri_final(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, const PencilCuts &pc,
                     const double *back, double *x, hipStream_t st) {
    const TriGeom t = make_geom(g, nt, kscale, cy, cx, pc);
    const dim3 grid((unsigned)((g.plane + 255) / 256));
    const int rw = tri_reg_width(g.ntl);
    if (rw == 16) DS_KLAUNCH(k_tri_final_reg<16>, grid, dim3(256), 0, st, t, back, x);
    else if (rw == 32) DS_KLAUNCH(k_tri_final_reg<32>, grid, dim3(256), 0, st, t, back, x);
    else if (rw == 64) DS_KLAUNCH(k_tri_final_reg<64>, grid, dim3(256), 0, st, t, back, x);
    else DS_KLAUNCH(k_tri_final, grid, dim3(256), 0, st, t, back, x);
    DS_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The single slab's t-axis solve by the same elimination (round 4): phi^ = idct_t(dct_t(r^) ./ kernel) IS the solution of
// D^2 ((CY + CX) I + T) phi^ = r^ per (ky, kx) mode, so the t axis needs no transform at all -- for ANY nt.  A workgroup of
// NSUB wavefronts owns 64 consecutive modes (one coalesced 512-byte segment per time layer); wavefront w holds the rows
// [t_w, t_{w+1}) of those modes in registers (R = 32 at nt = 128 / 129: ~110 registers, four waves per SIMD) and the NSUB
// pieces of a column are coupled exactly like time slabs: scaled sweep from the front (D_t kept in the registers) and the
// weighted sum from the back -> first / last entry of A_p^-1 g_p, exchanged through LDS -> every wave solves the small
// reduced system of its modes (closed-form coefficients) -> backward sweep with the neighbours' interface values
// (D_t is linear in the right-hand side: the left value enters as xl N_0 rho^t) -> one write.  One read and one write of
// the array, ~20 flops per entry: bound by HBM where the fused transform pass (two FFTs, seven barriers per tile) is
// bound by its own LDS / VALU chain.  Measured: 0.62 ms at 1024 x 1024 x 128 against 0.56 ms for the pipelined transform pass
// (which therefore stays for the power-of-two lengths), 0.66 ms at 1025 x 1025 x 129 against 0.86 ms for the prime-factor pass,
// and no dense t-axis product at all for the other lengths: the default whenever nt is no power of two (Solver::poisson_all).
// The singular (0, 0) mode: its column is parked in LDS, one thread runs k_tri_reduced's recurrence on it.
template <int R, int NSUB>
__global__ void __launch_bounds__(64 * NSUB) k_tsolve_single(TriGeom g, i64 nt, double *__restrict__ x) {
    // per piece and mode: first / last entry of A_p^-1 g_p and the piece's four spike values; then the interface values
    __shared__ double ex[NSUB][6][64];
    __shared__ double sw[NSUB][4][64];                 // the reduced sweep's A, B, al, ga (wave 0)
    __shared__ double zcol[NSUB * R];                  // the singular mode's column (workgroup 0 only)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 m = (i64)blockIdx.x * 64 + lane;
    const bool ok = m < g.plane;
    const i64 mc = ok ? m : 0;
    // rows of piece p: as evenly as possible (dotsocp_slab_range_impl's rule)
    const int base = (int)(nt / NSUB), rem = (int)(nt % NSUB);
    auto t_begin = [&](int p) { return p * base + (p < rem ? p : rem); };
    const int t0 = t_begin(w), n = t_begin(w + 1) - t0;
    const bool first = (w == 0), last = (w == NSUB - 1);
    const double sc = 1.0 / (g.kscale * g.beta);
    double X[R];
#pragma unroll
    for (int t = 0; t < R; ++t) X[t] = (t < n) ? x[mc + g.plane * (t0 + t)] * sc : 0.0;
    const bool zero = (m == 0);
    const TriCoef c = tri_coef(zero ? 1.0 : tri_aprime(g, mc));      // (the singular mode takes its own path below)
    if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int t = 0; t < R; ++t)
            if (t < n) zcol[t0 + t] = X[t];
    }
    // ---- local eliminations (the two running sums, see above) ----
    const double spe = first ? c.rho : -c.rho2;
    double H = 0.0, G = 0.0, pw = 1.0, D = 0.0;
#pragma unroll
    for (int t = 0; t < R; ++t) {
        if (t < n) {
            const double gt = X[t];
            H = gt + c.rho * H;
            G += pw * gt;
            D = H + (spe * pw) * G;
            X[t] = D;
            if (t + 1 < n) pw *= c.rho;                            // ends as rho^(n-1)
        }
    }
    const TriEnds f = tri_ends(c, first, n, pw), b = tri_ends(c, last, n, pw);
    if (NSUB > 1) {
        ex[w][0][lane] = tri_last(c, G + ((b.s * b.pe) * pw) * H, b.N1, b.Nn, first);
        ex[w][1][lane] = tri_last(c, D, f.N1, f.Nn, last);
        // the piece's spike values (tri_spike, from the ends already at hand)
        double vf, vl, wf, wl;
        if (last) {
            const double den = c.r * f.Nn - f.N1;
            vl = tri_div(pw * f.N0, den);
            wl = tri_div(f.N1, den);
        } else {
            vl = tri_div((pw * c.rho) * f.N0, f.Nn);
            wl = tri_div(c.rho * f.N1, f.Nn);
        }
        if (first) {
            const double den = c.r * b.Nn - b.N1;
            wf = tri_div(pw * b.N0, den);
            vf = tri_div(b.N1, den);
        } else {
            wf = tri_div((pw * c.rho) * b.N0, b.Nn);
            vf = tri_div(c.rho * b.N1, b.Nn);
        }
        ex[w][2][lane] = first ? 0.0 : vf;
        ex[w][3][lane] = first ? 0.0 : vl;
        ex[w][4][lane] = last ? 0.0 : wf;
        ex[w][5][lane] = last ? 0.0 : wl;
    }
    __syncthreads();
    // ---- reduced system of the NSUB pieces (k_tri_reduced's sweep), by wave 0 for the workgroup's 64 modes ----
    double xl = 0.0, xr = 0.0;
    if (NSUB > 1) {
        if (w == 0) {
#pragma unroll 1
            for (int p = 0; p < NSUB; ++p) {
                const double Gf = ex[p][0][lane], Gl = ex[p][1][lane];
                const double vf = ex[p][2][lane], vl = ex[p][3][lane], wf = ex[p][4][lane], wl = ex[p][5][lane];
                double A, B, al, ga;
                if (p == 0) {
                    A = Gf; B = wf; al = Gl; ga = wl;
                } else {
                    const double alp = sw[p - 1][2][lane], gap = sw[p - 1][3][lane];
                    const double den = 1.0 - vf * gap;
                    A = (Gf + vf * alp) / den;
                    B = wf / den;
                    al = Gl + vl * (alp + gap * A);
                    ga = wl + vl * gap * B;
                }
                sw[p][0][lane] = A; sw[p][1][lane] = B; sw[p][2][lane] = al; sw[p][3][lane] = ga;
            }
            double Fnext = 0.0;
#pragma unroll 1
            for (int p = NSUB - 1; p >= 0; --p) {
                const double Fp = sw[p][0][lane] + sw[p][1][lane] * Fnext;
                const double Lprev = (p > 0) ? sw[p - 1][2][lane] + sw[p - 1][3][lane] * Fp : 0.0;
                ex[p][0][lane] = Lprev;                            // the piece's left / right interface values
                ex[p][1][lane] = Fnext;
                Fnext = Fp;
            }
        }
        __syncthreads();
        xl = ex[w][0][lane];
        xr = ex[w][1][lane];
    }
    // ---- the singular mode ----
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            // T x = g - mean(g) by recurrence from x_0 = 0, then zero mean, plus beta * mean(g) (k_tri_reduced)
            double sum = 0.0;
            for (i64 t = 0; t < nt; ++t) sum += zcol[t];
            const double gbar = sum / (double)nt;
            double xm = 0.0, xc = 0.0, acc = 0.0;
            for (i64 t = 0; t < nt; ++t) {
                const double gt = zcol[t] - gbar;
                zcol[t] = xc;
                acc += xc;
                const double xn = (t == 0) ? xc - gt : 2.0 * xc - xm - gt;
                xm = xc;
                xc = xn;
            }
            const double shift = g.beta * gbar - acc / (double)nt;
            for (i64 t = 0; t < nt; ++t) zcol[t] += shift;
        }
        __syncthreads();
    }
    // ---- backward sweep with the interface values ----
    const double cl = xl * f.N0;                                   // D_t gains cl rho^t; D_{n-1} also xr N_{n-1}
    double xn = tri_last(c, (D + cl * pw) + xr * f.N1, f.N1, f.Nn, last);
    double Nt1 = f.N1;
#pragma unroll
    for (int t = R - 1; t >= 0; --t) {
        if (t == n - 1) {
            X[t] = xn;
        } else if (t < n - 1) {
            pw *= c.r;                                             // rho^t
            const double Nt = (t == 0) ? f.N0 : 1.0 + spe * (pw * pw);
            xn = tri_div(c.rho * ((X[t] + cl * pw) + Nt * xn), Nt1);
            X[t] = xn;
            Nt1 = Nt;
        }
    }
    if (ok) {
#pragma unroll
        for (int t = 0; t < R; ++t)
            if (t < n) x[m + g.plane * (t0 + t)] = zero ? zcol[t0 + t] : X[t];
    }
}

// The same solve as a PERSISTENT kernel fed by LDS-DMA (the recipe of dct.hip's pipelined passes): a workgroup walks tiles of
// 64 modes; the rows of the NEXT tile travel into an LDS image [row][mode] by global_load_lds_dwordx4 (no registers) while
// the current tile is eliminated in registers and stored, so loads are in flight all the time -- the one-tile-per-workgroup
// kernel above alternates between loading and computing (0.62 ms at nt = 128 where the traffic takes 0.4).  One LDS image:
// the DMA of tile i + 1 is issued behind the barrier that follows the forward sweep of tile i (every wave has copied its
// rows to registers by then); it has landed when only the stores issued after it are outstanding (vector-memory operations
// of a wave complete in issue order) -- counted waits and raw barriers, a fence would drain the counter.
template <int R, int NSUB>
__global__ void __launch_bounds__(64 * NSUB) k_tsolve_pipe(TriGeom g, i64 nt, int nTiles, double *__restrict__ x) {
    extern __shared__ double img[];                    // [nt rounded up to even][64]
    __shared__ double ex[NSUB][6][64];                 // per piece: Gf, Gl, vf, vl, wf, wl -> xl, xr, A, B, al, ga (in place)
    __shared__ double zcol[NSUB * R];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = (int)(nt / NSUB), rem = (int)(nt % NSUB);
    auto t_begin = [&](int p) { return p * base + (p < rem ? p : rem); };
    const int t0 = t_begin(w), n = t_begin(w + 1) - t0;
    const bool first = (w == 0), last = (w == NSUB - 1);
    const double sc = 1.0 / (g.kscale * g.beta);
    const unsigned ldsBase = (unsigned)(uintptr_t)img;
    const int npairs = (int)((nt + 1) / 2);            // one DMA instruction moves two rows (2 x 512 bytes)
    // every wave issues the pairs w, w + NSUB, ...; lane l fetches modes 2 (l & 31), +1 of row 2 pair + (l >> 5)
    auto dma = [&](int tile) {
        const i64 m0 = (i64)tile * 64;
        i64 mm = m0 + 2 * (lane & 31);
        if (mm + 1 >= g.plane) mm = g.plane - 2;       // (plane is even; clamped lanes fetch something valid, never used)
        for (int pr = w; pr < npairs; pr += NSUB) {
            int row = 2 * pr + (lane >> 5);
            if (row >= nt) row = (int)nt - 1;
            glds16(x + mm + g.plane * row, ldsBase + (unsigned)pr * 1024u);
        }
    };
    int tile = blockIdx.x;
    const int stride = gridDim.x;
    if (tile < nTiles) dma(tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the waits inside the loop count on the stores of a previous tile)
    for (; tile < nTiles; tile += stride) {
        // the tile has landed when nothing but this wave's stores of the previous tile (issued behind its DMA) is outstanding
        if (n == R) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(R) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(R > 1 ? R - 1 : 0) : "memory");
        lds_barrier();
        const i64 m = (i64)tile * 64 + lane;
        const bool ok = m < g.plane;
        const i64 mc = ok ? m : 0;
        double X[R];
#pragma unroll
        for (int t = 0; t < R; ++t) X[t] = (t < n) ? img[(t0 + t) * 64 + lane] * sc : 0.0;
        const bool zero = (m == 0);
        const TriCoef c = tri_coef(zero ? 1.0 : tri_aprime(g, mc));
        if (tile == 0 && lane == 0) {
#pragma unroll
            for (int t = 0; t < R; ++t)
                if (t < n) zcol[t0 + t] = X[t];
        }
        const double spe = first ? c.rho : -c.rho2;
        double H = 0.0, G = 0.0, pw = 1.0, D = 0.0;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            if (t < n) {
                const double gt = X[t];
                H = gt + c.rho * H;
                G += pw * gt;
                D = H + (spe * pw) * G;
                X[t] = D;
                if (t + 1 < n) pw *= c.rho;
            }
        }
        const TriEnds f = tri_ends(c, first, n, pw), b = tri_ends(c, last, n, pw);
        if (NSUB > 1) {
            ex[w][0][lane] = tri_last(c, G + ((b.s * b.pe) * pw) * H, b.N1, b.Nn, first);
            ex[w][1][lane] = tri_last(c, D, f.N1, f.Nn, last);
            double vf, vl, wf, wl;
            if (last) {
                const double den = c.r * f.Nn - f.N1;
                vl = tri_div(pw * f.N0, den);
                wl = tri_div(f.N1, den);
            } else {
                vl = tri_div((pw * c.rho) * f.N0, f.Nn);
                wl = tri_div(c.rho * f.N1, f.Nn);
            }
            if (first) {
                const double den = c.r * b.Nn - b.N1;
                wf = tri_div(pw * b.N0, den);
                vf = tri_div(b.N1, den);
            } else {
                wf = tri_div((pw * c.rho) * b.N0, b.Nn);
                vf = tri_div(c.rho * b.N1, b.Nn);
            }
            ex[w][2][lane] = first ? 0.0 : vf;
            ex[w][3][lane] = first ? 0.0 : vl;
            ex[w][4][lane] = last ? 0.0 : wf;
            ex[w][5][lane] = last ? 0.0 : wl;
        }
        lds_barrier();                                 // every wave has its rows in registers: the image is free
        if (tile + stride < nTiles) dma(tile + stride);
        double xl = 0.0, xr = 0.0;
        if (NSUB > 1) {
            if (w == 0) {
#pragma unroll 1
                for (int p = 0; p < NSUB; ++p) {
                    const double Gf = ex[p][0][lane], Gl = ex[p][1][lane];
                    const double vf = ex[p][2][lane], vl = ex[p][3][lane], wf = ex[p][4][lane], wl = ex[p][5][lane];
                    double A, B, al, ga;
                    if (p == 0) {
                        A = Gf; B = wf; al = Gl; ga = wl;
                    } else {
                        const double alp = ex[p - 1][4][lane], gap = ex[p - 1][5][lane];
                        const double den = 1.0 - vf * gap;
                        A = (Gf + vf * alp) / den;
                        B = wf / den;
                        al = Gl + vl * (alp + gap * A);
                        ga = wl + vl * gap * B;
                    }
                    ex[p][2][lane] = A; ex[p][3][lane] = B; ex[p][4][lane] = al; ex[p][5][lane] = ga;
                }
                double Fnext = 0.0;
#pragma unroll 1
                for (int p = NSUB - 1; p >= 0; --p) {
                    const double Fp = ex[p][2][lane] + ex[p][3][lane] * Fnext;
                    const double Lprev = (p > 0) ? ex[p - 1][4][lane] + ex[p - 1][5][lane] * Fp : 0.0;
                    ex[p][0][lane] = Lprev;
                    ex[p][1][lane] = Fnext;
                    Fnext = Fp;
                }
            }
            lds_barrier();
            xl = ex[w][0][lane];
            xr = ex[w][1][lane];
        }
        if (tile == 0) {                               // the singular mode (k_tri_reduced's recurrence)
            if (threadIdx.x == 0) {
                double sum = 0.0;
                for (i64 t = 0; t < nt; ++t) sum += zcol[t];
                const double gbar = sum / (double)nt;
                double xm = 0.0, xc = 0.0, acc = 0.0;
                for (i64 t = 0; t < nt; ++t) {
                    const double gt = zcol[t] - gbar;
                    zcol[t] = xc;
                    acc += xc;
                    const double xn = (t == 0) ? xc - gt : 2.0 * xc - xm - gt;
                    xm = xc;
                    xc = xn;
                }
                const double shift = g.beta * gbar - acc / (double)nt;
                for (i64 t = 0; t < nt; ++t) zcol[t] += shift;
            }
            lds_barrier();
        }
        const double cl = xl * f.N0;
        double xn = tri_last(c, (D + cl * pw) + xr * f.N1, f.N1, f.Nn, last);
        double Nt1 = f.N1;
#pragma unroll
        for (int t = R - 1; t >= 0; --t) {
            if (t == n - 1) {
                X[t] = xn;
            } else if (t < n - 1) {
                pw *= c.r;
                const double Nt = (t == 0) ? f.N0 : 1.0 + spe * (pw * pw);
                xn = tri_div(c.rho * ((X[t] + cl * pw) + Nt * xn), Nt1);
                X[t] = xn;
                Nt1 = Nt;
            }
        }
        // exactly n stores per wave and tile (the wait at the top counts them): lanes beyond the plane rewrite a valid entry
        // of their own tile's last mode?  no -- they store nothing: the count is per wave instruction, not per lane
#pragma unroll
        for (int t = 0; t < R; ++t)
            if (t < n) {
                if (ok) x[m + g.plane * (t0 + t)] = zero ? zcol[t0 + t] : X[t];
            }
    }
}

static int tri_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t pr;
        cus[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return cus[dev];
}

bool tsolve_tri_supported(i64 nt) { return nt >= 2 && nt <= 512; }

static bool tsolve_pipe_on() {
    const char *pe = getenv("DOTSOCP_TS_PIPE");        // read per call: the tests switch it inside one process
    return !(pe && atoi(pe) == 0);
}
// the persistent LDS-DMA flavour: grids with enough tiles to keep two workgroups per CU busy for many rounds; its image
// fits twice into a CU's LDS up to nt = 136
static bool tsolve_pipe_fits(i64 nt, i64 plane) {
    return nt > 64 && nt <= 136 && (plane + 63) / 64 >= 16 * (i64)tri_device_cus() && (plane % 2) == 0;
}
// Is the tridiagonal solve the faster t-axis solve of a single slab?  Every length without a power-of-two transform pass
// (prime-factor lengths 0.66 vs 0.86 ms at 1025 x 1025 x 129, and no dense product along t for the rest); powers of two
// where the pipelined flavour applies (0.51 vs 0.56 ms at 1024 x 1024 x 128; the one-tile-per-workgroup kernel: 0.62).
bool tsolve_tri_preferred(i64 nt, bool pow2, i64 plane) {
    if (!tsolve_tri_supported(nt)) return false;
    return !pow2 || (tsolve_pipe_on() && tsolve_pipe_fits(nt, plane));
}

// in place on x: [plane][nt] with plane = py * nx doubles per layer (pad entries of a row are modes of their own: zeros)
int launch_tsolve_tri(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, double *x, hipStream_t st) {
    PencilCuts pc{};
    pc.world = 1;
    pc.cut[0] = 0;
    pc.cut[1] = g.plane;
    const TriGeom t = make_geom(g, nt, kscale, cy, cx, pc);
    const dim3 grid((unsigned)((g.plane + 63) / 64));
    const int nTiles = (int)((g.plane + 63) / 64);
    const size_t img = (size_t)((nt + 1) / 2) * 2 * 64 * sizeof(double);
    // persistent LDS-DMA flavour (DOTSOCP_TS_PIPE=0: the one-tile-per-workgroup kernel)
    const int G = 2 * tri_device_cus();
    const bool pipe = tsolve_pipe_on() && tsolve_pipe_fits(nt, g.plane);
    if (pipe) {
        static std::mutex mu;
        static unsigned long long done = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lock(mu);
            if (dev >= 0 && dev < 64 && !(done & (1ull << dev))) {
                (void)hipFuncSetAttribute((const void *)(k_tsolve_pipe<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
                (void)hipFuncSetAttribute((const void *)(k_tsolve_pipe<34, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
                done |= 1ull << dev;
            }
        }
        if (nt <= 128) DS_KLAUNCH((k_tsolve_pipe<32, 4>), dim3((unsigned)G), dim3(256), img, st, t, nt, nTiles, x);
        else DS_KLAUNCH((k_tsolve_pipe<34, 4>), dim3((unsigned)G), dim3(256), img, st, t, nt, nTiles, x);
        DS_HIP(hipGetLastError());
        return 0;
    }
#define TSOLVE(RR, NS) DS_KLAUNCH((k_tsolve_single<RR, NS>), grid, dim3(64 * NS), 0, st, t, nt, x)
    if (nt <= 8) TSOLVE(8, 1);
    else if (nt <= 16) TSOLVE(8, 2);
    else if (nt <= 32) TSOLVE(16, 2);
    else if (nt <= 64) TSOLVE(16, 4);
    else if (nt <= 128) TSOLVE(32, 4);
    else if (nt <= 136) TSOLVE(34, 4);
    else if (nt <= 256) TSOLVE(32, 8);
    else if (nt <= 272) TSOLVE(34, 8);
    else if (nt <= 512) TSOLVE(64, 8);
    else { set_error("tridiagonal t-solve: nt > 512"); return DOTSOCP_EINVAL; }
#undef TSOLVE
    DS_HIP(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(256) k_gather_msgs(GatherMsgs a) {
    const int m = blockIdx.y;
    const double *__restrict__ s = a.src[m];
    double *__restrict__ d = a.dst[m];
    const i64 c = a.count[m];
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < c; i += (i64)gridDim.x * 256) d[i] = s[i];
}

int launch_gather_msgs(const GatherMsgs &m, hipStream_t st) {
    if (m.n <= 0) return 0;
    i64 cmax = 0;
    for (int i = 0; i < m.n; ++i) cmax = m.count[i] > cmax ? m.count[i] : cmax;
    if (cmax <= 0) return 0;
    const unsigned bx = (unsigned)launch_blocks(cmax, 256, 256);
    DS_KLAUNCH(k_gather_msgs, dim3(bx, (unsigned)m.n), dim3(256), 0, st, m);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
