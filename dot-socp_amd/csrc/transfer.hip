// Device-side versions of what the reference drivers do around the loop:
//   * multilevel transfer  socp/dot2d/utils/jump_nextLevel.m:5-16 with interpolate.m:20-84
//     (phi: linear in y, x, t; beta: nearest in t -- every coarse cell feeds two fine cells -- and linear in
//     y, x; q = grad * phi; alpha = -F*B*beta), fused with recoverOrgVar of the coarse level and
//     InitialScaling of the fine one (solver_dotsocp2d.m:304-386) so that no state array crosses PCIe;
//   * outputs  recover_RhoE.m:14-25, recover_q.m:12-22 (+ recoverOrgVar, :368-386).
// Index conventions as everywhere: y fastest, then x, then t; 1-D problems are ny = nx1d, nx = 1.
#include "device_utils.h"
#include "kernels.h"

namespace dotsocp {

// value of the fine point (yf, xf) of one (coarse) layer: linear along y first, then along x -- the order of
// the successive passes of interpolate.m; `f(yc, xc)` reads the (already scaled) coarse layer
template <class F>
__device__ __forceinline__ double interp_yx(const F &f, i64 yf, i64 xf) {
    const i64 yc = yf >> 1, xc = xf >> 1;
    const bool yo = yf & 1, xo = xf & 1;
    double a = f(yc, xc);
    if (yo) a = (a + f(yc + 1, xc)) / 2;
    if (xo) {
        double b = f(yc, xc + 1);
        if (yo) b = (b + f(yc + 1, xc + 1)) / 2;
        a = (a + b) / 2;
    }
    return a;
}

// phi_f = sc_out * interp(sc_in * phi_c)
// Time slabs: blockIdx.z is the slab-local fine layer, t0f + blockIdx.z the global one; `phic` holds the coarse layers
// tc0, tc0 + 1, ... (gathered from the coarse slabs that own them).
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_prolong_phi(i64 nyf, i64 nxf, i64 pyf, i64 t0f, i64 tc0, i64 pyc, i64 nxc,
                                                                 const double *__restrict__ phic,
                                                                 double *__restrict__ phif, double sc_in,
                                                                 double sc_out) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tloc = blockIdx.z, t = t0f + tloc;
    if (y >= nyf || x >= nxf) return;
    const i64 planec = pyc * nxc;                 // pyf, pyc: row pitches of the fine / coarse arrays (common.h)
    const i64 tc = (t >> 1) - tc0;
    auto layer = [&](i64 tt) {
        const double *p = phic + planec * tt;
        return interp_yx([&](i64 yc, i64 xc) { return sc_in * p[yc + pyc * xc]; }, y, x);
    };
    double v = layer(tc);
    if (t & 1) v = (v + layer(tc + 1)) / 2;
    phif[y + pyf * (x + nxf * tloc)] = sc_out * v;
}

// per cone column: betaR = interp(sc_in1 * (sc_in0 * beta_c)) ; beta_f = sc_out * betaR ; neg = -betaR
// (time slabs as in k_prolong_phi: t0f = global index of the slab's first fine cell, `betac` holds the coarse cell
// layers tc0, tc0 + 1, ... of every cone column, Nzc doubles per column)
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_prolong_beta(i64 nyf, i64 nxf, i64 pyf, i64 t0f, i64 tc0, i64 pyc, i64 nxc, i64 Nzc,
                                                                  i64 Nzf, const double *__restrict__ betac,
                                                                  double *__restrict__ betaf, double *__restrict__ neg,
                                                                  double sc_in0, double sc_in1, double sc_out) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tloc = blockIdx.z;              // slab-local fine cell
    if (y >= nyf || x >= nxf) return;
    const i64 tc = ((t0f + tloc) >> 1) - tc0; // interpolate.m:73-84: fR(:, :, oddT) = fR(:, :, evenT) = f
    const i64 i = y + pyf * (x + nxf * tloc);
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const double *p = betac + j * Nzc + pyc * nxc * tc;
        const double v = interp_yx([&](i64 yc, i64 xc) { return sc_in1 * (sc_in0 * p[yc + pyc * xc]); }, y, x);
        betaf[j * Nzf + i] = sc_out * v;
        neg[j * Nzf + i] = -v;
    }
}

// x = sc * x ./ w  (w == nullptr: x = sc * x)
__global__ void __launch_bounds__(256) k_scale_div(double *__restrict__ x, const double *__restrict__ w, i64 n, double sc) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        double v = x[i];
        if (w) v = v / w[i];
        x[i] = sc * v;
    }
}

int launch_prolong_phi(const Grid &gf, const Grid &gc, const double *phic, double *phif, double sc_in, double sc_out,
                       hipStream_t st, i64 tc0) {
    if (gf.ntl <= 0) return 0;
    dim3 grid((unsigned)((gf.ny + TILE_Y - 1) / TILE_Y), (unsigned)((gf.nx + TILE_X - 1) / TILE_X), (unsigned)gf.ntl);
    DS_KLAUNCH(k_prolong_phi, grid, dim3(TILE_Y, TILE_X), 0, st, gf.ny, gf.nx, gf.py, gf.t0, tc0, gc.py, gc.nx, phic, phif,
                       sc_in, sc_out);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_prolong_beta(const Grid &gf, const Grid &gc, const double *betac, double *betaf, double *neg, double sc_in0,
                        double sc_in1, double sc_out, hipStream_t st, i64 tc0, i64 Nzc) {
    if (gf.ncl <= 0) return 0;
    dim3 grid((unsigned)((gf.ny + TILE_Y - 1) / TILE_Y), (unsigned)((gf.nx + TILE_X - 1) / TILE_X), (unsigned)gf.ncl);
    DS_KLAUNCH(k_prolong_beta, grid, dim3(TILE_Y, TILE_X), 0, st, gf.ny, gf.nx, gf.py, gf.t0, tc0, gc.py, gc.nx,
                       Nzc < 0 ? gc.Nc : Nzc, gf.Nc, betac, betaf, neg, sc_in0, sc_in1, sc_out);
    DS_HIP(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(256) k_fill(double *__restrict__ x, i64 n, double v) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) x[i] = v;
}

int launch_fill(double *x, i64 n, double v, hipStream_t st) {
    if (n <= 0) return 0;
    DS_KLAUNCH(k_fill, dim3(launch_blocks(n, 256, 1 << 16)), dim3(256), 0, st, x, n, v);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_scale_div(double *x, const double *w, i64 n, double sc, hipStream_t st) {
    if (n <= 0) return 0;
    DS_KLAUNCH(k_scale_div, dim3(launch_blocks(n, 256, 1 << 22)), dim3(256), 0, st, x, w, n, sc);
    DS_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Outputs.  a(k) = [w(k) *] (cD * (sig * alpha(k)))  -- sigma of finish(), cScale * D of recoverOrgVar,
// weight of wdot2d/utils/recover_RhoE.m:11, in the order the host applies them;  b(k) = dD * q(k).
//   which 0: rho (ny, nx, nt)    rho0 | (a0(t-1) + a0(t)) / 2 | rho1                      recover_RhoE.m:16-18
//   which 1: Ex  (ny, nx, nt)    x-average of a_bx, first / last time layer doubled, 0 on the x boundary
//   which 2: Ey  (ny, nx, nt)    same from a_by
//   which 3: q0  (ny, nx, nt-1)
//   which 4: bx  (ny, nx, nt-1)  x-average (0 outside), then t-average                     recover_q.m:15-17
//   which 5: by  (ny, nx, nt-1)
// ---------------------------------------------------------------------------------------
// Time slabs: t below is the slab-local layer, g.t0 + t the global one; the density at a slab's first node
// averages over the left neighbour's last cell, whose a(.) arrives ready-made in `a_prev` (k_out_tail).
struct OutArgs {
    const double *q, *alpha, *weight, *rho0, *rho1, *a_prev;
    double sig, cD, dD;
};

__global__ void __launch_bounds__(TILE_Y *TILE_X) k_outputs(Grid g, OutArgs a, int which, double *__restrict__ out) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 t = blockIdx.z;
    if (y >= g.ny || x >= g.nx) return;
    const i64 nt = g.nt, tg = g.t0 + t;
    auto A = [&](i64 k) {
        const double v = a.cD * (a.sig * a.alpha[k]);
        return a.weight ? a.weight[k] * v : v;
    };
    auto B = [&](i64 k) { return a.dD * a.q[k]; };
    const i64 node = y + g.py * (x + g.nx * t);
    double r = 0.0;
    if (which == 0) {
        if (tg == 0) r = a.rho0[y + g.py * x];
        else if (tg == nt - 1) r = a.rho1[y + g.py * x];
        else r = ((t == 0 ? a.a_prev[y + g.py * x] : A(node - g.plane)) + A(node)) / 2;
    } else if (which == 1 || which == 2) {
        const double f = (tg == 0 || tg == nt - 1) ? 2.0 : 1.0;
        if (which == 1) {
            if (x >= 1 && x <= g.nx - 2) {
                const i64 e = g.offBx + g.bxLayer * t + y + g.py * x;
                r = (A(e - g.py) * f + A(e) * f) / 2;
            }
        } else {
            if (y >= 1 && y <= g.ny - 2) {
                const i64 e = g.offBy + g.byLayer * t + y + g.pyb * x;
                r = (A(e - 1) * f + A(e) * f) / 2;
            }
        }
    } else if (which == 3) {
        r = B(node);
    } else if (which == 4) {
        if (x >= 1 && x <= g.nx - 2) {
            const i64 e = g.offBx + g.bxLayer * t + y + g.py * x;
            const double m0 = (B(e - g.py) + B(e)) / 2;
            const double m1 = (B(e - g.py + g.bxLayer) + B(e + g.bxLayer)) / 2;
            r = (m0 + m1) / 2;
        }
    } else {
        if (y >= 1 && y <= g.ny - 2) {
            const i64 e = g.offBy + g.byLayer * t + y + g.pyb * x;
            const double m0 = (B(e - 1) + B(e)) / 2;
            const double m1 = (B(e - 1 + g.byLayer) + B(e + g.byLayer)) / 2;
            r = (m0 + m1) / 2;
        }
    }
    out[node] = r;
}

// a(.) of the slab's last cell layer, for the density at the right neighbour's first node
__global__ void __launch_bounds__(256) k_out_tail(Grid g, OutArgs a, double *__restrict__ out) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.plane) return;
    const i64 k = i + g.plane * (g.ncl - 1);
    const double v = a.cD * (a.sig * a.alpha[k]);
    out[i] = a.weight ? a.weight[k] * v : v;
}

int launch_out_tail(const Grid &g, const double *alpha, const double *weight, double sig, double cD, double *out,
                    hipStream_t st) {
    OutArgs a{nullptr, alpha, weight, nullptr, nullptr, nullptr, sig, cD, 0.0};
    DS_KLAUNCH(k_out_tail, dim3((unsigned)((g.plane + 255) / 256)), dim3(256), 0, st, g, a, out);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_outputs(const Grid &g, const double *q, const double *alpha, const double *weight, const double *rho0,
                   const double *rho1, const double *a_prev, double sig, double cD, double dD, int which, double *out,
                   hipStream_t st) {
    OutArgs a{q, alpha, weight, rho0, rho1, a_prev, sig, cD, dD};
    const i64 layers = (which >= 3) ? g.ncl : g.ntl;
    if (layers <= 0) return 0;
    dim3 grid((unsigned)((g.ny + TILE_Y - 1) / TILE_Y), (unsigned)((g.nx + TILE_X - 1) / TILE_X), (unsigned)layers);
    DS_KLAUNCH(k_outputs, grid, dim3(TILE_Y, TILE_X), 0, st, g, a, which, out);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
