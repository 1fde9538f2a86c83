// Kernels of the accelerated ADMM loop (socp/dot2d/algorithms/solver_socp_accADMM.m, weighted:
// socp/wdot2d/algorithms/solver_wsocp_accADMM.m).  Unlike inPALM, z is a state variable here (it is
// extrapolated, so it is no longer a projection of something that can be regenerated), and every
// variable has an anchor (Halpern) or a previous extrapolation point (theta != 2).
//
// k_acc_cone, per cell, with q = q^+ (the q-step result of this iteration), (z, beta) = current state:
//   beta^+ = beta + z - (BF q^+ + d)                              multiplier step   (:236-238)
//   z^+    = Pi_Q(BF q^+ + d - beta^+)                            z-step            (:248)
//   MODE_RAW   : store z^+, beta^+                    (iterations with a KKT check / theta != 2)
//   MODE_FUSED : Halpern step folded in               (:373-379)
//                  x_new = c1 x0 + c2 ((1-rho) x + rho x^+)   for x = z, beta (x0 = anchor),
//                store z_new, beta_new (other buffers: chunks re-read the cell in front of them) and
//                emit q2 = F* B* (z_new + beta_new), the adjoint sums of the NEXT iteration's q-step (:229)
//   MODE_GATHER: q2 = F* B* (z + beta) only.
//   MODE_RESTART: MODE_FUSED for the iteration whose KKT check changed sigma (:346-358): beta and beta^+ are divided by the
//                factor on the way (k_scale's arithmetic), the anchors are x^+ itself and are STORED here instead of read --
//                one pass in place of two scalings, two anchor copies, two extrapolation passes and the gather pass.
// traffic, MODE_FUSED: z, beta, z0, beta0 in + z, beta out + q + q2 = 8 (60 Nz + 2 Nq) bytes.
// Mapping and gather: identical to the fused inPALM kernel (fused.hip / gather_tile.h).
#include "device_utils.h"
#include "gather_tile.h"
#include "kernels.h"

namespace dotsocp {

enum { ACC_RAW = 0, ACC_FUSED = 1, ACC_GATHER = 2, ACC_RESTART = 3 };

// KKT (MODE_RAW, one slab, iteration with a KKT check): besides storing z^+, beta^+ the pass takes the cell part of the KKT
// sums and the F*B*beta^+ terms of every edge -- exactly what k_kkt_cells<., true> (kkt.hip) does for the inPALM loop: same
// sums, same gather, same split (edges inside the tile summed here with one load of alpha^+, edges on the tile's right /
// upper border left as raw partials in q2 / sx / sy for k_kkt_bnd).
template <int MODE, int XB, bool NT = false, bool KKT = false, bool WEIGHTED = false>
__global__ void __launch_bounds__(64 * XB) k_acc_cone(Grid g, LoopCoef c, AccArgs a) {
    __shared__ double2 xch[2][XB][64];
    const int lane = threadIdx.x, xl = threadIdx.y;
    const i64 y = (i64)blockIdx.x * 64 + lane;
    const i64 x = (i64)blockIdx.y * XB + xl;
    const bool inb = (y < g.ny) && (x < g.nx);
    const i64 yc = inb ? y : 0, xc = inb ? x : 0;
    const i64 t0 = (i64)blockIdx.z * a.TC;
    const i64 t1 = (t0 + a.TC < g.ncl) ? t0 + a.TC : g.ncl;
    const bool lastChunk = (t1 == g.ncl);
    const bool gathers = (MODE != ACC_RAW) || KKT;
    const i64 tstart = (gathers && t0 > 0) ? t0 - 1 : t0;
    const i64 tstop = (gathers && lastChunk) ? t1 + 1 : t1;
    const i64 nxblk = gridDim.y, nyblk = gridDim.x;

    EdgeQuad cur{};
    if (MODE != ACC_GATHER) cur = load_edges(g, a.q, yc, xc, tstart, c.sf);
    GatherCarry gc;
    double S[KKT ? S_COUNT : 1];
    if (KKT) {
#pragma unroll
        for (int i = 0; i < S_COUNT; ++i) S[i] = 0.0;
    }
    auto wgt = [&](i64 idx) { return WEIGHTED ? a.weight[idx] : 1.0; };
    const KktCoef &k = a.kk;
    for (i64 tl = tstart; tl < tstop; ++tl) {
        const bool hasCell = tl < g.ncl;
        const bool own = tl >= t0;
        double w[10];
        if (hasCell) {
            const i64 i = yc + g.py * (xc + g.nx * tl);
            double b[10], zz[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) b[j] = ld_stream<NT>(a.beta_in + j * g.Nc + i);
#pragma unroll
            for (int j = 0; j < 10; ++j) zz[j] = ld_stream<NT>(a.z_in + j * g.Nc + i);
            if (MODE == ACC_GATHER) {
#pragma unroll
                for (int j = 0; j < 10; ++j) w[j] = zz[j] + b[j];
            } else {
                const EdgeQuad nxt = load_edges(g, a.q, yc, xc, tl + 1, c.sf);
                double v[10], bp[10];
                const double q0 = a.q[i];
                build_z2(v, q0, cur, nxt, c.s, c.dF);
                cur = nxt;
                double z2[KKT ? 10 : 1];
                if (KKT) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) z2[j] = v[j];
                }
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    double t = b[j] + zz[j];          // beta + z - z2, left to right (:238)
                    bp[j] = t - v[j];
                }
#pragma unroll
                for (int j = 0; j < 10; ++j) v[j] = v[j] - bp[j];
                proj_row<10>(v);                      // v = z^+
                if (MODE == ACC_RAW) {
                    if (own && inb && !(KKT && a.nostore)) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) st_stream<NT>(a.beta_out + j * g.Nc + i, bp[j]);
#pragma unroll
                        for (int j = 0; j < 10; ++j) st_stream<NT>(a.z_out + j * g.Nc + i, v[j]);
                    }
                    if (KKT) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) w[j] = bp[j];           // the gather below is F*B*beta^+
                    }
                    if (KKT && own && inb) {
                        // the cell sums of k_kkt_cells (kkt.hip) at z = z^+, beta' = beta^+, BF q + d = z2
                        double zs = 0.0, bs = 0.0, rs = 0.0;
#pragma unroll
                        for (int j = 0; j < 10; ++j) {
                            const double r = v[j] - z2[j];
                            zs += v[j] * v[j];
                            bs += bp[j] * bp[j];
                            rs += r * r;
                        }
                        auto xj = [&](int j) { return v[j] - k.sigma * bp[j]; };
                        double nn = xj(1) * xj(1);
#pragma unroll
                        for (int j = 2; j < 10; ++j) nn += xj(j) * xj(j);
                        const double n = sqrt(nn), x0 = xj(0);
                        double cf = (x0 / n + 1.0) * 0.5;
                        cf = (cf > 1.0) ? 1.0 : cf;
                        cf = (cf < 0.0) ? 0.0 : cf;
                        const double p0 = (cf >= 1.0) ? x0 : cf * n;
                        double cs = (v[0] - p0) * (v[0] - p0);
#pragma unroll
                        for (int j = 1; j < 10; ++j) {
                            const double d = v[j] - cf * xj(j);
                            cs += d * d;
                        }
                        S[S_Z2] += zs;
                        S[S_BETA2] += bs;
                        S[S_PRIM2] += rs;
                        S[S_COMPLEM] += cs;
                        const double wc = wgt(i);
                        const double av = a.alpha_p[i];
                        const double rhoT = k.kappa * (wc * av);
                        double sq = 0.0;
#pragma unroll
                        for (int j = 1; j < 9; ++j) {
                            const double e = k.dsE * z2[j];
                            sq += e * e;
                        }
                        double rhoFq = rhoT + k.dsD * q0 + sq / 4.0;
                        rhoFq = (rhoFq < 0.0) ? 0.0 : rhoFq;
                        const double dd = rhoT - rhoFq;
                        S[S_DOTCOMP] += dd * dd;
                        S[S_RHO2] += rhoT * rhoT;
                        S[S_RHOFQ2] += rhoFq * rhoFq;
                        const double q2b = c.s * (bp[9] - bp[0]);
                        S[S_FBBETA2] += q2b * q2b;
                        const double r2 = q2b + wc * av;
                        S[S_DUAL2] += r2 * r2;
                    }
                } else {
                    double zn[10], bn[10];
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        const double z0 = (MODE == ACC_RESTART) ? v[j] : ld_stream<NT>(a.z0 + j * g.Nc + i);
                        double t = a.om_rho * zz[j];
                        t = t + a.rho * v[j];
                        zn[j] = a.c1 * z0 + a.c2 * t;
                    }
                    if (MODE == ACC_RESTART) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) {
                            b[j] = b[j] / a.bdiv;
                            bp[j] = bp[j] / a.bdiv;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        const double b0 = (MODE == ACC_RESTART) ? bp[j] : ld_stream<NT>(a.beta0 + j * g.Nc + i);
                        double t = a.om_rho * b[j];
                        t = t + a.rho * bp[j];
                        bn[j] = a.c1 * b0 + a.c2 * t;
                    }
                    if (own && inb) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) st_stream<NT>(a.z_out + j * g.Nc + i, zn[j]);
#pragma unroll
                        for (int j = 0; j < 10; ++j) st_stream<NT>(a.beta_out + j * g.Nc + i, bn[j]);
                        if (MODE == ACC_RESTART) {
#pragma unroll
                            for (int j = 0; j < 10; ++j) st_stream<NT>(a.z0_out + j * g.Nc + i, v[j]);
#pragma unroll
                            for (int j = 0; j < 10; ++j) st_stream<NT>(a.beta0_out + j * g.Nc + i, bp[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 10; ++j) w[j] = zn[j] + bn[j];
                }
            }
            if (gathers && !KKT && own && inb) a.q2[i] = c.s * (w[9] - w[0]);
        } else {
#pragma unroll
            for (int j = 0; j < 10; ++j) w[j] = 0.0;
        }
        if constexpr (KKT) {
            // edge layer tl of F*B*beta^+ with gather_emit's arithmetic and order; the two sums on the spot inside the tile,
            // raw partials for k_kkt_bnd on its right / upper border (k_kkt_cells<., true>)
            auto dual = [&](i64 e, double acc) {
                const double gb = c.sf * acc;
                S[S_FBBETA2] += gb * gb;
                const double r2 = gb + wgt(e) * a.alpha_p[e];
                S[S_DUAL2] += r2 * r2;
            };
            const bool st = own && inb;
            xch[gc.par][xl][lane] = make_double2(w[1], gc.p3);
            __syncthreads();
            if (st) {
                if (x < g.nx - 1) {
                    const i64 e = g.offBx + g.bxLayer * tl + y + g.py * x;
                    if (xl < XB - 1) {
                        const double2 r = xch[gc.par][xl + 1][lane];
                        double acc = r.x + w[2];
                        acc += r.y;
                        acc += gc.p4;
                        dual(e, acc);
                    } else {
                        a.q2[e] = w[2] + gc.p4;
                    }
                }
                if (xl == 0 && x > 0) a.sx[(tl * nxblk + blockIdx.y) * g.ny + y] = w[1] + gc.p3;
            }
            const double u5 = __shfl_down(w[5], 1, 64), u7 = __shfl_down(gc.p7, 1, 64);
            if (st) {
                if (y < g.ny - 1) {
                    const i64 e = g.offBy + g.byLayer * tl + y + g.pyb * x;
                    if (lane < 63) {
                        double acc = u5 + w[6];
                        acc += u7;
                        acc += gc.p8;
                        dual(e, acc);
                    } else {
                        a.q2[e] = w[6] + gc.p8;
                    }
                }
                if (lane == 0 && y > 0) a.sy[(tl * g.nx + x) * nyblk + blockIdx.x] = w[5] + gc.p7;
            }
            gc.p3 = w[3]; gc.p4 = w[4]; gc.p7 = w[7]; gc.p8 = w[8];
            gc.par ^= 1;
        } else if (gathers) {
            gather_emit<XB>(g, c.sf, xch, gc, w, tl, own && inb, x, y, xl, lane, nxblk, nyblk, blockIdx.y, blockIdx.x,
                            a.q2, a.sx, a.sy);
        }
    }
    if constexpr (KKT) {
        __shared__ double red[XB][S_COUNT];
#pragma unroll
        for (int i = 0; i < S_COUNT; ++i) {
            double v = S[i];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[threadIdx.y][i] = v;
        }
        __syncthreads();
        if (threadIdx.y == 0 && lane < S_COUNT) {
            double v = red[0][lane];
#pragma unroll
            for (int wv = 1; wv < XB; ++wv) v += red[wv][lane];
            const i64 bb = blockIdx.x + (i64)gridDim.x * (blockIdx.y + (i64)gridDim.y * blockIdx.z);
            a.partials[bb * S_COUNT + lane] = v;
        }
    }
}

int launch_acc_cone_kkt(const Grid &g, const LoopCoef &c, const FusedGeom &fg, AccArgs a, const KktWork &w, hipStream_t st) {
    if (g.Nz <= 0) return 0;
    a.TC = fg.TC;
    dim3 grid((unsigned)fg.nyblk, (unsigned)fg.nxblk, (unsigned)fg.chunks);
    dim3 blk(64, 4);
    const i64 region = kkt_partials_needed(g) / 8;            // KKT_REGIONS regions (kkt.hip)
    a.partials = w.partials + 1 * region * S_COUNT;
    const bool nt = stream_nt_enabled();
    if (a.weight) {
        if (nt) DS_KLAUNCH((k_acc_cone<ACC_RAW, 4, true, true, true>), grid, blk, 0, st, g, c, a);
        else DS_KLAUNCH((k_acc_cone<ACC_RAW, 4, false, true, true>), grid, blk, 0, st, g, c, a);
    } else {
        if (nt) DS_KLAUNCH((k_acc_cone<ACC_RAW, 4, true, true, false>), grid, blk, 0, st, g, c, a);
        else DS_KLAUNCH((k_acc_cone<ACC_RAW, 4, false, true, false>), grid, blk, 0, st, g, c, a);
    }
    DS_HIP(hipGetLastError());
    return launch_kkt_bnd_dual(g, c, a.kk, fg, a.q, a.alpha_p, a.weight, a.q2, a.sx, a.sy, w, st);
}

int launch_acc_cone(int mode, const Grid &g, const LoopCoef &c, const FusedGeom &fg, AccArgs a, hipStream_t st) {
    if (g.Nz <= 0) return 0;
    a.TC = fg.TC;
    dim3 grid((unsigned)fg.nyblk, (unsigned)fg.nxblk, (unsigned)fg.chunks);
    dim3 blk(64, 4);
    const bool nt = stream_nt_enabled();
#define ACC_MODE(M)                                                                    \
    case M:                                                                            \
        if (nt) DS_KLAUNCH((k_acc_cone<M, 4, true>), grid, blk, 0, st, g, c, a);       \
        else DS_KLAUNCH((k_acc_cone<M, 4>), grid, blk, 0, st, g, c, a);                \
        break;
    switch (mode) {
        ACC_MODE(ACC_RAW) ACC_MODE(ACC_FUSED) ACC_MODE(ACC_GATHER) ACC_MODE(ACC_RESTART)
#undef ACC_MODE
        default: set_error("bad acc cone mode"); return DOTSOCP_EINVAL;
    }
    DS_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Extrapolation of one state array, element-wise and in place (solver_socp_accADMM.m:369-423).
//   x  : current state (= "Old" of the reference: the two are equal at the top of every iteration)
//   xp : x^+ of this iteration
//   mode 0 (Halpern, :373-379):   x = c1 aux + c2 ((1-rho) x + rho xp)            aux = anchor
//   mode 1 (theta != 2, k == 0):  hat = (1-rho) x + rho xp ; x = (1-c1) x + c1 hat               (:390-402)
//   mode 2 (theta != 2, k  > 0):  x = (1-c1) x + (c1+c2) hat - c2 aux             aux = hatOld  (:404-410)
//   modes 1, 2 store hat into aux when write_aux (:420)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_acc_interp(double *__restrict__ x, const double *__restrict__ xp,
                                                    double *__restrict__ aux, i64 n, AccCoef k, int mode,
                                                    int write_aux) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const double xo = x[i], xn = xp[i];
        double hat = k.om_rho * xo;
        hat = hat + k.rho * xn;
        double r;
        if (mode == 0) {
            r = k.c1 * aux[i] + k.c2 * hat;
        } else {
            r = k.om_c1 * xo;
            if (mode == 1) {
                r = r + k.c1 * hat;
            } else {
                r = r + k.c1c2 * hat;
                r = r - k.c2 * aux[i];
            }
            if (write_aux) aux[i] = hat;
        }
        x[i] = r;
    }
}

// Halpern extrapolation of one array right after a sigma update (:346-358 followed by :373-379 with k = 0): x and x^+ are
// divided by the factor (alpha; div = 1 for phi and q), the anchor is x^+ itself -- stored -- and
// x = c1 x^+ + c2 ((1-rho) x + rho x^+); k_scale's, the anchor copy's and k_acc_interp's arithmetic in one pass.
__global__ void __launch_bounds__(256) k_acc_restart(double *__restrict__ x, const double *__restrict__ xp,
                                                     double *__restrict__ anchor, i64 n, AccCoef k, double div, int use_div) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        double xo = x[i], xn = xp[i];
        if (use_div) {
            xo = xo / div;
            xn = xn / div;
        }
        anchor[i] = xn;
        double hat = k.om_rho * xo;
        hat = hat + k.rho * xn;
        x[i] = k.c1 * xn + k.c2 * hat;
    }
}

int launch_acc_restart(double *x, const double *xp, double *anchor, i64 n, const AccCoef &k, double div, hipStream_t st) {
    if (n <= 0) return 0;
    DS_KLAUNCH(k_acc_restart, dim3(launch_blocks(n, 256, 1 << 22)), dim3(256), 0, st, x, xp, anchor, n, k, div,
               (int)(div != 1.0));
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_acc_interp(double *x, const double *xp, double *aux, i64 n, const AccCoef &k, int mode, int write_aux,
                      hipStream_t st) {
    if (n <= 0) return 0;
    DS_KLAUNCH(k_acc_interp, dim3(launch_blocks(n, 256, 1 << 22)), dim3(256), 0, st, x, xp, aux, n, k, mode,
                       write_aux);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
