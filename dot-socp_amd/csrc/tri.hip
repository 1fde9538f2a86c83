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

// delta_t of a block with n rows: a' + 2, minus 1 on the global first / last row
__device__ __forceinline__ double tri_delta(double ap, i64 t, i64 n, bool first, bool last) {
    return ap + 2.0 - ((first && t == 0) ? 1.0 : 0.0) - ((last && t == n - 1) ? 1.0 : 0.0);
}

__device__ __forceinline__ int tri_owner(const PencilCuts &pc, i64 m, i64 plane) {
    int j = (int)((m * pc.world) / plane);
    while (j > 0 && m < pc.cut[j]) --j;
    while (j < pc.world - 1 && m >= pc.cut[j + 1]) ++j;
    return j;
}

// message to owner j starts at 2 cut[j] + TRI_EXTRA j and holds [first values | last values | TRI_EXTRA extras]
__device__ __forceinline__ i64 tri_msg_off(const PencilCuts &pc, int j) { return 2 * pc.cut[j] + (i64)TRI_EXTRA * j; }

__global__ void __launch_bounds__(256) k_tri_local(TriGeom g, const double *__restrict__ r, double *__restrict__ send) {
    const i64 m = (i64)blockIdx.x * 256 + threadIdx.x;
    if (m >= g.plane) return;
    const double ap = tri_aprime(g, m);
    const double sc = 1.0 / (g.kscale * g.beta);
    const i64 n = g.ntl;
    auto G = [&](i64 t) { return r[m + g.plane * t] * sc; };
    // elimination from the front: last entry of A^{-1} g
    double piv = tri_delta(ap, 0, n, g.first, g.last), d = G(0);
    for (i64 t = 1; t < n; ++t) {
        const double inv = 1.0 / piv;
        d = G(t) + d * inv;
        piv = tri_delta(ap, t, n, g.first, g.last) - inv;
    }
    const double Gl = d / piv;
    // elimination from the back: first entry
    piv = tri_delta(ap, n - 1, n, g.first, g.last);
    d = G(n - 1);
    for (i64 t = n - 2; t >= 0; --t) {
        const double inv = 1.0 / piv;
        d = G(t) + d * inv;
        piv = tri_delta(ap, t, n, g.first, g.last) - inv;
    }
    const double Gf = d / piv;
    const int j = tri_owner(g.pc, m, g.plane);
    const i64 off = tri_msg_off(g.pc, j), w = g.pc.cut[j + 1] - g.pc.cut[j];
    send[off + (m - g.pc.cut[j])] = Gf;
    send[off + w + (m - g.pc.cut[j])] = Gl;
    if (m == 0)                                   // the singular mode travels whole
        for (i64 t = 0; t < n; ++t) send[off + 2 * w + t] = G(t);
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
    const double ap = tri_aprime(g, m);
    double A[PMAX], B[PMAX], al[PMAX], ga[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        if (p >= q.P) break;
        const i64 n = q.slab_n[p];
        const bool first = (p == 0), last = (p == q.P - 1);
        // first / last entries of A_p^{-1} e_first (vf, vl) and A_p^{-1} e_last (wf, wl)
        double piv = tri_delta(ap, 0, n, first, last), prod = 1.0;
        for (i64 t = 1; t < n; ++t) {
            const double inv = 1.0 / piv;
            prod *= inv;
            piv = tri_delta(ap, t, n, first, last) - inv;
        }
        double vl = prod / piv, wl = 1.0 / piv;
        piv = tri_delta(ap, n - 1, n, first, last);
        prod = 1.0;
        for (i64 t = n - 2; t >= 0; --t) {
            const double inv = 1.0 / piv;
            prod *= inv;
            piv = tri_delta(ap, t, n, first, last) - inv;
        }
        double vf = 1.0 / piv, wf = prod / piv;
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

// A_p x = g + e_first x_left + e_last x_right; the forward sweep leaves d'_t / m_t in x and 1 / m_t in qinv
__global__ void __launch_bounds__(256) k_tri_final(TriGeom g, const double *__restrict__ back, double *__restrict__ x,
                                                    double *__restrict__ qinv) {
    const i64 m = (i64)blockIdx.x * 256 + threadIdx.x;
    if (m >= g.plane) return;
    const i64 n = g.ntl;
    const int j = tri_owner(g.pc, m, g.plane);
    const i64 off = tri_msg_off(g.pc, j), w = g.pc.cut[j + 1] - g.pc.cut[j];
    if (m == 0) {
        for (i64 t = 0; t < n; ++t) x[g.plane * t] = back[off + 2 * w + t];
        return;
    }
    const double ap = tri_aprime(g, m);
    const double sc = 1.0 / (g.kscale * g.beta);
    const double xl = back[off + (m - g.pc.cut[j])], xr = back[off + w + (m - g.pc.cut[j])];
    double piv = tri_delta(ap, 0, n, g.first, g.last);
    double d = x[m] * sc + xl + ((n == 1) ? xr : 0.0);
    double inv = 1.0 / piv;
    x[m] = d * inv;
    qinv[m] = inv;
    for (i64 t = 1; t < n; ++t) {
        double gt = x[m + g.plane * t] * sc;
        if (t == n - 1) gt += xr;
        d = gt + d * inv;
        piv = tri_delta(ap, t, n, g.first, g.last) - inv;
        inv = 1.0 / piv;
        x[m + g.plane * t] = d * inv;
        qinv[m + g.plane * t] = inv;
    }
    double xn = x[m + g.plane * (n - 1)];
    for (i64 t = n - 2; t >= 0; --t) {
        xn = x[m + g.plane * t] + qinv[m + g.plane * t] * xn;
        x[m + g.plane * t] = xn;
    }
}


// Register-resident flavours for short slabs (ntl <= NTL): the column of a mode is read ONCE into registers, both
// eliminations (k_tri_local) resp. the whole Thomas solve (k_tri_final) run there, and the result is written once --
// 1 and 2 passes over the slab instead of 2 and 6.  Same operations in the same order as the kernels above.
template <int NTL>
__global__ void __launch_bounds__(256) k_tri_local_reg(TriGeom g, const double *__restrict__ r, double *__restrict__ send) {
    const i64 m = (i64)blockIdx.x * 256 + threadIdx.x;
    if (m >= g.plane) return;
    const double ap = tri_aprime(g, m);
    const double sc = 1.0 / (g.kscale * g.beta);
    const int n = (int)g.ntl;
    double G[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) G[t] = (t < n) ? r[m + g.plane * t] * sc : 0.0;
    double piv = tri_delta(ap, 0, n, g.first, g.last), d = G[0];
#pragma unroll
    for (int t = 1; t < NTL; ++t) {
        if (t < n) {
            const double inv = 1.0 / piv;
            d = G[t] + d * inv;
            piv = tri_delta(ap, t, n, g.first, g.last) - inv;
        }
    }
    const double Gl = d / piv;
#pragma unroll
    for (int t = NTL - 1; t >= 0; --t) {
        if (t == n - 1) {
            piv = tri_delta(ap, t, n, g.first, g.last);
            d = G[t];
        } else if (t < n - 1) {
            const double inv = 1.0 / piv;
            d = G[t] + d * inv;
            piv = tri_delta(ap, t, n, g.first, g.last) - inv;
        }
    }
    const double Gf = d / piv;
    const int j = tri_owner(g.pc, m, g.plane);
    const i64 off = tri_msg_off(g.pc, j), w = g.pc.cut[j + 1] - g.pc.cut[j];
    send[off + (m - g.pc.cut[j])] = Gf;
    send[off + w + (m - g.pc.cut[j])] = Gl;
    if (m == 0) {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
            if (t < n) send[off + 2 * w + t] = G[t];
    }
}

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
    const double ap = tri_aprime(g, m);
    const double sc = 1.0 / (g.kscale * g.beta);
    const double xl = back[off + (m - g.pc.cut[j])], xr = back[off + w + (m - g.pc.cut[j])];
    double X[NTL], Q[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) X[t] = (t < n) ? x[m + g.plane * t] : 0.0;
    double piv = tri_delta(ap, 0, n, g.first, g.last);
    double d = X[0] * sc + xl + ((n == 1) ? xr : 0.0);
    double inv = 1.0 / piv;
    X[0] = d * inv;
    Q[0] = inv;
#pragma unroll
    for (int t = 1; t < NTL; ++t) {
        Q[t] = 0.0;
        if (t < n) {
            double gt = X[t] * sc;
            if (t == n - 1) gt += xr;
            d = gt + d * inv;
            piv = tri_delta(ap, t, n, g.first, g.last) - inv;
            inv = 1.0 / piv;
            X[t] = d * inv;
            Q[t] = inv;
        }
    }
    double xn = 0.0;
#pragma unroll
    for (int t = NTL - 1; t >= 0; --t) {
        if (t == n - 1) {
            xn = X[t];
        } else if (t < n - 1) {
            xn = X[t] + Q[t] * xn;
            X[t] = xn;
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
    const int rw = tri_reg_width(g.ntl);
    if (rw == 16) DS_KLAUNCH(k_tri_local_reg<16>, grid, dim3(256), 0, st, t, r, send);
    else if (rw == 32) DS_KLAUNCH(k_tri_local_reg<32>, grid, dim3(256), 0, st, t, r, send);
    else if (rw == 64) DS_KLAUNCH(k_tri_local_reg<64>, grid, dim3(256), 0, st, t, r, send);
    else DS_KLAUNCH(k_tri_local, grid, dim3(256), 0, st, t, r, send);
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
    else DS_KLAUNCH(k_tri_final, grid, dim3(256), 0, st, t, back, x, qinv);
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
