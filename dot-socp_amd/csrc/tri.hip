// Time-slab Poisson solve WITHOUT the slab <-> pencil transposes (option B).
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

#include <cstdlib>

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
// The sequence N_j, j = 0 .. n, of a sweep that starts on a boundary row (bnd) or not: its last two members, and what a
// DOWNWARD walk through the powers rho^(2j+e) needs to stay clear of underflow: above `jsave` the power is below 1e-20
// (N_j = 1 to the last bit), at jsave it is `psave`, below it grows by r^2 per step.
struct TriSeq {
    int jsave;
    double psave, N1, Nn;              // N_{n-1}, N_n
};
__device__ __forceinline__ TriSeq tri_seq(const TriCoef &c, bool bnd, int n) {
    TriSeq w{-1, 0.0, 0.0, 0.0};
    double P = bnd ? c.rho : c.rho2, N = tri_n0(c, bnd);
    for (int j = 0; j < n; ++j) {
        if (P >= 1e-20) { w.jsave = j; w.psave = P; }
        P *= c.rho2;
        if (j + 1 < n) N = c.n0d + c.rho2 * N;
    }
    w.N1 = N;
    w.Nn = c.n0d + c.rho2 * N;
    return w;
}
// the last unknown of a sweep over n rows from its D_{n-1}; bnd: the sweep ENDS on a global boundary row
__device__ __forceinline__ double tri_last(const TriCoef &c, double D, double N1, double Nn, bool bnd) {
    return bnd ? tri_div(D, c.r * Nn - N1) : tri_div(c.rho * D, Nn);
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
    // from the front: D_t = g_t N_t + rho D_{t-1}; from the back, as a weighted sum: F = sum_t rho^t N'_{n-1-t} g_t
    const TriSeq sb = tri_seq(c, last, n);        // the reversed column starts on the slab's LAST row
    const double sgn = last ? 1.0 : -1.0;
    double N = tri_n0(c, first), N1 = N, D = 0.0, F = 0.0, pw = 1.0, Pb = 0.0;
    constexpr int U = 4;
    for (int t0 = 0; t0 < n; t0 += U) {
        double gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) gv[u] = (t0 + u < n) ? r[m + g.plane * (t0 + u)] * sc : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            if (t < n) {
                D = gv[u] * N + c.rho * D;
                N1 = N;
                N = c.n0d + c.rho2 * N;                    // N_{t+1}
                const int jb = n - 1 - t;
                if (jb == sb.jsave) Pb = sb.psave;
                else if (jb < sb.jsave) Pb *= c.r2;
                F += (pw * (1.0 + sgn * Pb)) * gv[u];
                pw *= c.rho;
            }
        }
    }
    send[off + (m - g.pc.cut[j])] = tri_last(c, F, sb.N1, sb.Nn, first);      // first entry of A^-1 g
    send[off + w + (m - g.pc.cut[j])] = tri_last(c, D, N1, N, last);          // last entry
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
        const int n = (int)q.slab_n[p];
        const bool first = (p == 0), last = (p == q.P - 1);
        // first / last entries of A_p^{-1} e_first (vf, vl) and A_p^{-1} e_last (wf, wl) in closed form:
        // prod_{s < n-1} 1 / piv_s = rho^(n-1) N_0 / N_{n-1}, 1 / piv_{n-1} = rho N_{n-1} / N_n (boundary end: N_{n-1} / (r N_n - N_{n-1}))
        const TriSeq sf = tri_seq(c, first, n), sb = tri_seq(c, last, n);
        double rn1 = 1.0;                                  // rho^(n-1)
        for (int t = 1; t < n; ++t) rn1 *= c.rho;
        double vl, wl, vf, wf;
        if (last) {
            const double den = c.r * sf.Nn - sf.N1;
            vl = tri_div(rn1 * tri_n0(c, first), den);
            wl = tri_div(sf.N1, den);
        } else {
            vl = tri_div((rn1 * c.rho) * tri_n0(c, first), sf.Nn);
            wl = tri_div(c.rho * sf.N1, sf.Nn);
        }
        if (first) {
            const double den = c.r * sb.Nn - sb.N1;
            wf = tri_div(rn1 * tri_n0(c, last), den);
            vf = tri_div(sb.N1, den);
        } else {
            wf = tri_div((rn1 * c.rho) * tri_n0(c, last), sb.Nn);
            vf = tri_div(c.rho * sb.N1, sb.Nn);
        }
        if (first) vf = vl = 0.0;                 // no left / right neighbour
        if (last) wf = wl = 0.0;
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

// A_p x = g + e_first x_left + e_last x_right by the scaled sweeps above: forward D_t = g_t N_t + rho D_{t-1} (no division),
// backward x_t = rho (D_t + N_t x_{t+1}) / N_{t+1} (one fast division per row), N_t of the backward walk from the powers.
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
    const TriSeq sf = tri_seq(c, first, n);
    double N = tri_n0(c, first), N1 = N, D = 0.0;
    for (int t = 0; t < n; ++t) {
        double gt = x[m + g.plane * t] * sc;
        if (t == 0) gt += xl;
        if (t == n - 1) gt += xr;
        D = gt * N + c.rho * D;
        x[m + g.plane * t] = D;
        N1 = N;
        N = c.n0d + c.rho2 * N;
    }
    double xn = tri_last(c, D, N1, N, last);
    x[m + g.plane * (n - 1)] = xn;
    const double sgn = first ? 1.0 : -1.0;
    double P = (sf.jsave == n - 1) ? sf.psave : 0.0, Nt1 = N1;      // N_{t+1} of the row below
    for (int t = n - 2; t >= 0; --t) {
        if (t == sf.jsave) P = sf.psave;
        else if (t < sf.jsave) P *= c.r2;
        const double Nt = 1.0 + sgn * P;
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
    const TriSeq sf = tri_seq(c, first, n);
    double N = tri_n0(c, first), N1 = N, D = 0.0;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        if (t < n) {
            double gt = X[t] * sc;
            if (t == 0) gt += xl;
            if (t == n - 1) gt += xr;
            D = gt * N + c.rho * D;
            X[t] = D;
            N1 = N;
            N = c.n0d + c.rho2 * N;
        }
    }
    double xn = tri_last(c, D, N1, N, last);
    const double sgn = first ? 1.0 : -1.0;
    double P = (sf.jsave == n - 1) ? sf.psave : 0.0, Nt1 = N1;
#pragma unroll
    for (int t = NTL - 1; t >= 0; --t) {
        if (t == n - 1) {
            X[t] = xn;
        } else if (t < n - 1) {
            if (t == sf.jsave) P = sf.psave;
            else if (t < sf.jsave) P *= c.r2;
            const double Nt = 1.0 + sgn * P;
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
                     const double *back, double *x, double *qinv, hipStream_t st) {
    const TriGeom t = make_geom(g, nt, kscale, cy, cx, pc);
    const dim3 grid((unsigned)((g.plane + 255) / 256));
    const int rw = tri_reg_width(g.ntl);
    if (rw == 16) DS_KLAUNCH(k_tri_final_reg<16>, grid, dim3(256), 0, st, t, back, x);
    else if (rw == 32) DS_KLAUNCH(k_tri_final_reg<32>, grid, dim3(256), 0, st, t, back, x);
    else if (rw == 64) DS_KLAUNCH(k_tri_final_reg<64>, grid, dim3(256), 0, st, t, back, x);
    else DS_KLAUNCH(k_tri_final, grid, dim3(256), 0, st, t, back, x);
    (void)qinv;
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
