// Matrix-free staggered-gradient stencils of the loop: rhs = A'(w.*q - alpha) + c and the
// q-step / alpha-update.  A = D * [Dt; Dx; Dy] are forward differences
// (socp/dot2d/utils/initialize.m:35-39,67-87, scaled by D in solver_dotsocp2d.m:338);
// summation orders follow the column/row order of the reference's sparse products
//  so that results agree with the oracle to the last bit.
#include "device_utils.h"
#include "kernels.h"

#include <cstdlib>

namespace dotsocp {

static inline dim3 tile_grid(const Grid &g, i64 layers) {
    return dim3((unsigned)((g.ny + TILE_Y - 1) / TILE_Y), (unsigned)((g.nx + TILE_X - 1) / TILE_X), (unsigned)layers);
}

// ---------------------------------------------------------------------------------------
// rhs(y,x,t) = sum over the (up to) six staggered neighbours, Neumann: missing ones dropped
// (solver_socp_inPALM.m:194; weighted: solver_wsocp_inPALM.m:200)
// ---------------------------------------------------------------------------------------
template <bool WEIGHTED>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_rhs(RhsArgs a, double *__restrict__ rhs) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = blockIdx.z;
    if (y >= a.g.ny || x >= a.g.nx) return;
    rhs[y + a.g.py * (x + a.g.nx * tl)] = rhs_value<WEIGHTED>(a, y, x, tl);
}

int launch_rhs(const Grid &g, const LoopCoef &c, const double *q, const double *alpha, const double *cvec,
               const double *weight, const double *u0_prev, double *rhs, hipStream_t st) {
    if (g.Nphi <= 0) return 0;
    RhsArgs a{g, c.at, c.ax, c.ay, q, alpha, cvec, weight, u0_prev};
    if (weight)
        DS_KLAUNCH(k_rhs<true>, tile_grid(g, g.ntl), dim3(TILE_Y, TILE_X), 0, st, a, rhs);
    else
        DS_KLAUNCH(k_rhs<false>, tile_grid(g, g.ntl), dim3(TILE_Y, TILE_X), 0, st, a, rhs);
    DS_HIP(hipGetLastError());
    return 0;
}

template <bool WEIGHTED>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_u0_tail(Grid g, const double *__restrict__ q,
                                                             const double *__restrict__ alpha,
                                                             const double *__restrict__ weight,
                                                             double *__restrict__ out) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    if (y >= g.ny || x >= g.nx) return;
    const i64 k = y + g.py * (x + g.nx * (g.ncl - 1));
    out[y + g.py * x] = WEIGHTED ? weight[k] * q[k] - alpha[k] : q[k] - alpha[k];
}

int launch_u0_tail(const Grid &g, const double *q, const double *alpha, const double *weight, double *out,
                   hipStream_t st) {
    if (g.ncl <= 0) return 0;
    if (weight)
        DS_KLAUNCH(k_u0_tail<true>, tile_grid(g, 1), dim3(TILE_Y, TILE_X), 0, st, g, q, alpha, weight, out);
    else
        DS_KLAUNCH(k_u0_tail<false>, tile_grid(g, 1), dim3(TILE_Y, TILE_X), 0, st, g, q, alpha, weight, out);
    DS_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// q-step and alpha update, one staggered entry per thread:
//   tmp = (A phi)_k ; q2 = (F* B* (z + beta))_k
//   q_k = (w_k (tmp + alpha_k) + q2) * diagQInv_k          (solver_socp_inPALM.m:204-206; w: :212)
//   alpha_k += tau (tmp - w_k q_k)                           (:211,214; w: :217)
// diagQInv = 1 ./ oper_q  (socp/dot2d/utils/oper_q.m:13-26, wdot2d/utils/oper_q.m:15-28)
// ---------------------------------------------------------------------------------------
struct WSum {
    const double *z, *b;
    i64 Nz;
    __device__ __forceinline__ double operator()(int j, i64 cell) const { return z[j * Nz + cell] + b[j * Nz + cell]; }
};

// VAR 0: inPALM / ALG2; 1: acc-ADMM (alpha_out = (alpha_in + tmp) - w q); 2: PALM's first q-step (alpha untouched)
template <bool WEIGHTED, int VAR = 0>
__device__ __forceinline__ void q_update(const LoopCoef &c, double tmp, double q2, double diag_c, double dinv, i64 k,
                                         const double *__restrict__ weight, double *__restrict__ q,
                                         double *alpha, const double *alpha_in = nullptr) {
    const double a = (VAR == 1) ? alpha_in[k] : alpha[k];
    double qn, r;
    if (WEIGHTED) {
        const double w = weight[k];
        const double di = 1.0 / (diag_c + w * w);
        qn = (w * (tmp + a) + q2) * di;
        r = tmp - w * qn;
    } else {
        qn = (tmp + a + q2) * dinv;
        r = tmp - qn;
    }
    q[k] = qn;
    if (VAR == 2) return;
    if (VAR == 1) {
        // alpha + tmp_q - w.*q, left to right (solver_socp_accADMM.m:237, solver_wsocp_accADMM.m:243)
        const double t = a + tmp;
        alpha[k] = WEIGHTED ? t - weight[k] * qn : t - qn;
    } else {
        alpha[k] = a + c.tau * r;
    }
}

template <bool WEIGHTED, int SEG>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_qstep(Grid g, LoopCoef c, const double *__restrict__ phi,
                                                           const double *__restrict__ z,
                                                           const double *__restrict__ beta,
                                                           const double *__restrict__ weight,
                                                           const double *__restrict__ tail_bx,
                                                           const double *__restrict__ tail_by,
                                                           double *__restrict__ q, double *__restrict__ alpha) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = blockIdx.z;
    WSum W{z, beta, g.Nc};
    const i64 node = y + g.py * (x + g.nx * tl);
    if (SEG == 0) {
        if (y >= g.ny || x >= g.nx) return;
        double tmp = (-c.at) * phi[node];
        tmp += c.at * phi[node + g.plane];
        const double q2 = c.s * (W(9, node) - W(0, node));
        q_update<WEIGHTED>(c, tmp, q2, c.c1, c.dinv1, node, weight, q, alpha);
    } else {
        const bool tbnd = (g.t0 + tl == 0) || (g.t0 + tl == g.nt - 1);
        const double dc = tbnd ? c.c2 : c.c1;
        const double di = tbnd ? c.dinv2 : c.dinv1;
        if (SEG == 1) {
            if (y >= g.ny || x >= g.nx - 1) return;
            double tmp = (-c.ax) * phi[node];
            tmp += c.ax * phi[node + g.py];
            const double q2 = c.sf * gather_bx(g, W, y, x, tl, tail_bx);
            q_update<WEIGHTED>(c, tmp, q2, dc, di, g.offBx + g.bxLayer * tl + y + g.py * x, weight, q, alpha);
        } else {
            if (y >= g.ny - 1 || x >= g.nx) return;
            double tmp = (-c.ay) * phi[node];
            tmp += c.ay * phi[node + 1];
            const double q2 = c.sf * gather_by(g, W, y, x, tl, tail_by);
            q_update<WEIGHTED>(c, tmp, q2, dc, di, g.offBy + g.byLayer * tl + y + g.pyb * x, weight, q, alpha);
        }
    }
}

template <bool WEIGHTED>
static int launch_qstep_t(const Grid &g, const LoopCoef &c, const double *phi, const double *z, const double *beta,
                          const double *weight, const double *tail_bx, const double *tail_by, double *q,
                          double *alpha, hipStream_t st) {
    dim3 blk(TILE_Y, TILE_X);
    if (g.Nz > 0)
        DS_KLAUNCH((k_qstep<WEIGHTED, 0>), tile_grid(g, g.ncl), blk, 0, st, g, c, phi, z, beta, weight, tail_bx,
                           tail_by, q, alpha);
    if (g.bxLayer > 0)
        DS_KLAUNCH((k_qstep<WEIGHTED, 1>), tile_grid(g, g.ntl), blk, 0, st, g, c, phi, z, beta, weight, tail_bx,
                           tail_by, q, alpha);
    if (g.byLayer > 0)
        DS_KLAUNCH((k_qstep<WEIGHTED, 2>), tile_grid(g, g.ntl), blk, 0, st, g, c, phi, z, beta, weight, tail_bx,
                           tail_by, q, alpha);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_qstep(const Grid &g, const LoopCoef &c, const double *phi, const double *z, const double *beta,
                 const double *weight, const double *tail_bx, const double *tail_by, double *q, double *alpha,
                 hipStream_t st) {
    return weight ? launch_qstep_t<true>(g, c, phi, z, beta, weight, tail_bx, tail_by, q, alpha, st)
                  : launch_qstep_t<false>(g, c, phi, z, beta, weight, tail_bx, tail_by, q, alpha, st);
}

// q-step on the adjoint sums produced by the fused cone kernel (fused.hip): q2 already holds
// sf * sum for tile-interior edges and the own tile's raw partial for tile-boundary edges.
template <bool WEIGHTED, int SEG, int VAR = 0>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_qstep_fused(Grid g, LoopCoef c, FusedGeom fg,
                                                                 const double *__restrict__ phi,
                                                                 const double *__restrict__ q2v,
                                                                 const double *__restrict__ sx,
                                                                 const double *__restrict__ sy,
                                                                 const double *__restrict__ weight,
                                                                 const double *__restrict__ tail_bx,
                                                                 const double *__restrict__ tail_by,
                                                                 double *__restrict__ q, double *alpha,
                                                                 const double *alpha_in) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = blockIdx.z;
    const i64 node = y + g.py * (x + g.nx * tl);
    if (SEG == 3) {
        // one launch for all three kinds of entries: the thread of node (y, x, tl) owns the q0 entry of
        // the cell that starts there and the bx / by edges that leave it; phi(node) is loaded once
        if (y >= g.ny || x >= g.nx) return;
        const double p0 = phi[node];
        if (tl < g.ncl) {
            double tmp = (-c.at) * p0;
            tmp += c.at * phi[node + g.plane];
            q_update<WEIGHTED, VAR>(c, tmp, q2v[node], c.c1, c.dinv1, node, weight, q, alpha, alpha_in);
        }
        const bool tbnd = (g.t0 + tl == 0) || (g.t0 + tl == g.nt - 1);
        const double dc = tbnd ? c.c2 : c.c1;
        const double di = tbnd ? c.dinv2 : c.dinv1;
        if (x < g.nx - 1) {
            const i64 e = g.offBx + g.bxLayer * tl + y + g.py * x;
            double tmp = (-c.ax) * p0;
            tmp += c.ax * phi[node + g.py];
            double q2 = q2v[e];
            if ((x % fg.XB) == fg.XB - 1) q2 = c.sf * (q2 + sx[(tl * fg.nxblk + (x / fg.XB + 1)) * g.ny + y]);
            if (tl == 0 && !g.first) q2 += tail_bx[y + g.py * x];
            q_update<WEIGHTED, VAR>(c, tmp, q2, dc, di, e, weight, q, alpha, alpha_in);
        }
        if (y < g.ny - 1) {
            const i64 e = g.offBy + g.byLayer * tl + y + g.pyb * x;
            double tmp = (-c.ay) * p0;
            tmp += c.ay * phi[node + 1];
            double q2 = q2v[e];
            if ((y & 63) == 63) q2 = c.sf * (q2 + sy[(tl * g.nx + x) * fg.nyblk + (y / 64 + 1)]);
            if (tl == 0 && !g.first) q2 += tail_by[y + g.pyb * x];
            q_update<WEIGHTED, VAR>(c, tmp, q2, dc, di, e, weight, q, alpha, alpha_in);
        }
    } else if (SEG == 0) {
        if (y >= g.ny || x >= g.nx) return;
        double tmp = (-c.at) * phi[node];
        tmp += c.at * phi[node + g.plane];
        q_update<WEIGHTED, VAR>(c, tmp, q2v[node], c.c1, c.dinv1, node, weight, q, alpha, alpha_in);
    } else {
        const bool tbnd = (g.t0 + tl == 0) || (g.t0 + tl == g.nt - 1);
        const double dc = tbnd ? c.c2 : c.c1;
        const double di = tbnd ? c.dinv2 : c.dinv1;
        if (SEG == 1) {
            if (y >= g.ny || x >= g.nx - 1) return;
            const i64 e = g.offBx + g.bxLayer * tl + y + g.py * x;
            double tmp = (-c.ax) * phi[node];
            tmp += c.ax * phi[node + g.py];
            double q2 = q2v[e];
            if ((x % fg.XB) == fg.XB - 1) q2 = c.sf * (q2 + sx[(tl * fg.nxblk + (x / fg.XB + 1)) * g.ny + y]);
            if (tl == 0 && !g.first) q2 += tail_bx[y + g.py * x];      // left slab's part, already times sf
            q_update<WEIGHTED, VAR>(c, tmp, q2, dc, di, e, weight, q, alpha, alpha_in);
        } else {
            if (y >= g.ny - 1 || x >= g.nx) return;
            const i64 e = g.offBy + g.byLayer * tl + y + g.pyb * x;
            double tmp = (-c.ay) * phi[node];
            tmp += c.ay * phi[node + 1];
            double q2 = q2v[e];
            if ((y & 63) == 63) q2 = c.sf * (q2 + sy[(tl * g.nx + x) * fg.nyblk + (y / 64 + 1)]);
            if (tl == 0 && !g.first) q2 += tail_by[y + g.pyb * x];
            q_update<WEIGHTED, VAR>(c, tmp, q2, dc, di, e, weight, q, alpha, alpha_in);
        }
    }
}

template <bool WEIGHTED>
static int launch_qstep_fused_t(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi,
                                const double *q2, const double *sx, const double *sy, const double *weight,
                                const double *tail_bx, const double *tail_by, double *q, double *alpha,
                                hipStream_t st) {
    dim3 blk(TILE_Y, TILE_X);
    DS_KLAUNCH((k_qstep_fused<WEIGHTED, 3>), tile_grid(g, g.ntl), blk, 0, st, g, c, fg, phi, q2, sx, sy, weight,
                       tail_bx, tail_by, q, alpha, (const double *)nullptr);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_qstep_fused(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi, const double *q2,
                       const double *sx, const double *sy, const double *weight, const double *tail_bx,
                       const double *tail_by, double *q_out, double *alpha, hipStream_t st) {
    return weight ? launch_qstep_fused_t<true>(g, c, fg, phi, q2, sx, sy, weight, tail_bx, tail_by, q_out, alpha, st)
                  : launch_qstep_fused_t<false>(g, c, fg, phi, q2, sx, sy, weight, tail_bx, tail_by, q_out, alpha, st);
}


// ---------------------------------------------------------------------------------------
// q-step + alpha update + the NEXT iteration's right-hand side in one pass (fused dataflow):
//   q^{k+1}, alpha^{k+1} as in k_qstep_fused, then rhs = A'(w.*q^{k+1} - alpha^{k+1}) + c (solver_socp_inPALM.m:194
//   of iteration k+1) while u = w.*q - alpha is still in registers: saves re-reading q and alpha (6 of the 8
//   arrays k_rhs streams).  A workgroup owns a 64 (y) x 4 (x) tile of nodes and marches through a chunk of
//   time layers; the thread of node (y, x, tl) owns the q0 entry of the cell that starts there and the bx /
//   by edges that leave it.  u of the t-1 cell is carried in a register, u of the x-1 edge comes through LDS,
//   u of the y-1 edge through a lane shuffle; on a tile / chunk boundary the neighbour's entry is recomputed
//   (reads only -- alpha is ping-ponged, so no other workgroup's writes are observed).  The sum order is the
//   one of rhs_value().  Time-slab mode: the term of the left neighbour's last cell is added by k_rhs_fixup
//   after the u0 exchange.
// ---------------------------------------------------------------------------------------
// MULT 0: alpha + tau (A phi - w q) (inPALM); 1: (alpha + A phi) - w q (acc-ADMM); 2: alpha stays (PALM's first q-step)
// A scaling of alpha that is still pending in memory (sigma update, solver_socp_inPALM.m:312: alpha = alpha / factor) is
// applied on load with k_scale's arithmetic; the q-step writes the scaled values into the ping-pong partner.
struct APend {
    int on;
    double mul, div;
};

template <bool WEIGHTED, int MULT = 0>
__device__ __forceinline__ double q_value(const LoopCoef &c, double tmp, double q2, double diag_c, double dinv, i64 k,
                                          const double *__restrict__ weight, const double *__restrict__ alpha_in,
                                          double &qn, double &an, double &u, const APend &ap = APend{0, 1.0, 1.0}) {
    double a = alpha_in[k];
    if (ap.on) a = a * ap.mul / ap.div;
    if (WEIGHTED) {
        const double w = weight[k];
        const double di = 1.0 / (diag_c + w * w);
        qn = (w * (tmp + a) + q2) * di;
        if (MULT == 2) {
            an = a;
        } else if (MULT == 1) {
            const double t = a + tmp;                 // alpha + tmp_q - w.*q (solver_wsocp_accADMM.m:243)
            an = t - w * qn;
        } else {
            const double r = tmp - w * qn;
            an = a + c.tau * r;
        }
        u = w * qn - an;
    } else {
        qn = (tmp + a + q2) * dinv;
        if (MULT == 2) {
            an = a;
        } else if (MULT == 1) {
            const double t = a + tmp;                 // alpha + tmp_q - q (solver_socp_accADMM.m:237)
            an = t - qn;
        } else {
            const double r = tmp - qn;
            an = a + c.tau * r;
        }
        u = qn - an;
    }
    return a;
}

// q_value on operands that are already in registers (the q-step's load phase): a = alpha_in[k], w = weight[k]
template <bool WEIGHTED, int MULT = 0>
__device__ __forceinline__ double q_calc(const LoopCoef &c, double tmp, double q2, double diag_c, double dinv, double w,
                                         double a, const APend &ap, double &qn, double &an, double &u) {
    if (ap.on) a = a * ap.mul / ap.div;
    if (WEIGHTED) {
        const double di = 1.0 / (diag_c + w * w);
        qn = (w * (tmp + a) + q2) * di;
        if (MULT == 2) {
            an = a;
        } else if (MULT == 1) {
            const double t = a + tmp;                 // alpha + tmp_q - w.*q (solver_wsocp_accADMM.m:243)
            an = t - w * qn;
        } else {
            const double r = tmp - w * qn;
            an = a + c.tau * r;
        }
        u = w * qn - an;
    } else {
        qn = (tmp + a + q2) * dinv;
        if (MULT == 2) {
            an = a;
        } else if (MULT == 1) {
            const double t = a + tmp;                 // alpha + tmp_q - q (solver_socp_accADMM.m:237)
            an = t - qn;
        } else {
            const double r = tmp - qn;
            an = a + c.tau * r;
        }
        u = qn - an;
    }
    return a;
}

// F*B*(BF q + d) of one q entry, with mexBFd's / mexBFdConj's arithmetic: a q0 entry sits in columns 1 and 10 of its cell
// (d cancels), a staggered edge in one column of each of the four cells around it -- two on the first and the last layer
__device__ __forceinline__ double fbbf_cell(const LoopCoef &c, double q0) {
    return c.s * ((c.dF + c.s * q0) - (c.dF - c.s * q0));
}
__device__ __forceinline__ double fbbf_edge(const LoopCoef &c, double e, bool tbnd) {
    const double v = c.sf * e;
    double acc = v + v;
    if (!tbnd) {
        acc += v;
        acc += v;
    }
    return c.sf * acc;
}

struct QRhsArgs {
    const double *phi, *q2v, *sx, *sy, *weight, *tail_bx, *tail_by, *cvec, *alpha_in;
    double *q_out, *alpha_out, *rhs;
    double *u0_tail;   // time slabs, VAR 0 - 2 (optional): raw u0 = w.*q0^+ - alpha0^+ of the last owned cell layer, for the right slab's rhs
    i64 TC, z0, zstride;   // layers per chunk; this launch runs the chunks z0 + blockIdx.z * zstride
    // VAR 2 (acc-ADMM, Halpern step folded in): q_out receives the raw q^+ (the cone pass needs it), the
    // extrapolated q goes to q_state in place and the extrapolated alpha to alpha_out
    double *q_state;
    const double *q_anchor, *alpha_anchor;
    double c1, c2, om_rho, rho;
    APend ap;          // pending scaling of alpha_in (VAR 0)
    int xcd;           // XCD-aware tile order
    // VAR 3 with the gather given as p2 = F*B*((1 + tau) z + beta) (k_cone_fused modes 5 / 6): qk = q^k, and the gather the
    // q-step uses is p2 - tau F*B*(BF q^k + d)
    const double *qk;
    // KKT variant (VAR 0, single slab): per-workgroup partial sums, r = A' alpha^+ - c per node, DOT complementarity scalars
    double *partials, *resid;
    double kappa, dsD;
};

// VAR 0: inPALM / ALG2; 1: acc-ADMM multiplier arithmetic, raw outputs; 2: acc-ADMM with the Halpern step of q and
// alpha folded in (solver_socp_accADMM.m:373-379); 3: PALM's first q-step (q only, solver_socp_PALM.m:196-200).
// The rhs is formed from the raw u = w.*q^+ - alpha^+ in all cases (VAR 3: alpha^+ = alpha)
//
// KKT = true (VAR 0, one slab): the iteration ends with a KKT check (solver_socp_inPALM.m:220-267).  Everything of that
// block that depends on phi^{k+1}, q^{k+1}, alpha^{k+1}, A phi and c only is accumulated here, where those values are in
// registers anyway: ||q||^2, ||alpha||^2, ||A phi||^2, ||A phi - w q||^2, <w q, alpha>, <c, phi>, ||phi||^2,
// ||A' alpha - c||^2 (a second accumulation next to the rhs, alpha of the x-1 / y-1 / t-1 entries travelling beside u) and
// the momentum terms of compute_kkt_dot_complement.m:10-18 for all edges whose two density nodes lie in this tile (the
// edges on the tile's right / upper border are left to k_kkt_bnd).  r = A' alpha - c is also stored per node: after a
// sigma update the right-hand side of the next phi-step is rhs + r - r / factor (launch_rhs_sigma_fix) instead of a new pass.
enum { Q_Q2 = 0, Q_ALPHA2, Q_APHI2, Q_PRIM1, Q_QALPHA, Q_CPHI, Q_PHI2, Q_DUAL1, Q_MRHOB, Q_M2, Q_RHOB2, Q_COUNT };

template <bool WEIGHTED, int VAR, bool KKT = false, int QTX = TILE_X>
__global__ void __launch_bounds__(TILE_Y *QTX, (QTX > TILE_X && !WEIGHTED ? 4 : 1)) k_qstep_rhs(Grid g, LoopCoef c, FusedGeom fg, QRhsArgs a) {
    __shared__ double xch[2][QTX][TILE_Y];
    // phi of the layer the march stands on, with a one-entry halo in x and y: every phi entry is fetched ONCE per tile and
    // layer (own column as the "t + 1" value of the step before, the four halo strips by the border lanes) and the x / y
    // neighbours are read from here -- read from global they cost a second fetch of the whole layer, a step later
    __shared__ double ph[2][QTX + 2][TILE_Y + 2];
    __shared__ double xcha[KKT ? 2 : 1][KKT ? QTX : 1][KKT ? TILE_Y : 1];   // alpha^+ of the bx edge
    __shared__ double xchr[KKT ? 2 : 1][KKT ? QTX : 1][KKT ? TILE_Y : 1];   // density at the node
    double S[Q_COUNT];     // KKT only (dead code otherwise)
    if (KKT) {
#pragma unroll
        for (int i = 0; i < Q_COUNT; ++i) S[i] = 0.0;
    }
    auto wgt = [&](i64 k) { return WEIGHTED ? a.weight[k] : 1.0; };
    // sums every staggered entry contributes to (edge_sums of k_kkt)
    auto entry = [&](double tmp, double qn, double an, double w) {
        const double wq = w * qn;
        S[Q_Q2] += qn * qn;
        S[Q_ALPHA2] += an * an;
        S[Q_APHI2] += tmp * tmp;
        const double r1 = tmp - wq;
        S[Q_PRIM1] += r1 * r1;
        S[Q_QALPHA] += wq * an;
    };
    double a0prev = 0.0, rhoTprev = 0.0;
    const int lane = threadIdx.x, xl = threadIdx.y;
    // XCD-aware tile order (device_utils.h): tiles that are neighbours in y or x run on the same XCD back to back, so
    // what they share -- the cache lines of the by rows (length ny - 1: never line-aligned), the phi row above, the
    // neighbour tile's edge that is recomputed here -- is served by that XCD's L2 instead of a second HBM fetch
    const BlockId blk = block_id(a.xcd != 0);
    const i64 y = (i64)blk.x * TILE_Y + lane;
    const i64 x = (i64)blk.y * QTX + xl;
    const bool inb = (y < g.ny) && (x < g.nx);
    const i64 t0 = ((i64)blk.z * a.zstride + a.z0) * a.TC;
    const i64 t1 = (t0 + a.TC < g.ntl) ? t0 + a.TC : g.ntl;
    auto put = [&](i64 k, double qn, double an, double ain) {
        a.q_out[k] = qn;
        if (VAR == 3) return;                         // PALM's first q-step: alpha is not touched
        if (VAR == 2) {
            double t = a.om_rho * a.q_state[k];
            t = t + a.rho * qn;
            a.q_state[k] = a.c1 * a.q_anchor[k] + a.c2 * t;
            t = a.om_rho * ain;
            t = t + a.rho * an;
            a.alpha_out[k] = a.c1 * a.alpha_anchor[k] + a.c2 * t;
        } else {
            a.alpha_out[k] = an;
        }
    };
    double u0prev = 0.0;
    double p0 = 0.0;
    const bool hasBx0 = inb && (x < g.nx - 1), hasBy0 = inb && (y < g.ny - 1);
    const bool rightCol = hasBx0 && (xl == QTX - 1), topRow = hasBy0 && (lane == TILE_Y - 1);
    if (inb) {
        const i64 node0 = y + g.py * (x + g.nx * t0);
        p0 = a.phi[node0];
        // halo strips of the chunk's first layer
        double hx = 0.0, hl = 0.0, hy = 0.0, hb = 0.0;
        if (rightCol) hx = a.phi[node0 + g.py];
        if (xl == 0 && x >= 1) hl = a.phi[node0 - g.py];
        if (topRow) hy = a.phi[node0 + 1];
        if (lane == 0 && y >= 1) hb = a.phi[node0 - 1];
        ph[0][xl + 1][lane + 1] = p0;
        if (xl == QTX - 1) ph[0][QTX + 1][lane + 1] = hx;
        if (xl == 0) ph[0][0][lane + 1] = hl;
        if (lane == TILE_Y - 1) ph[0][xl + 1][TILE_Y + 1] = hy;
        if (lane == 0) ph[0][xl + 1][0] = hb;
        if (t0 > 0) {       // cell in front of the chunk (owned by the previous chunk): recompute, do not store
            const i64 k = node0 - g.plane;
            double tmp = (-c.at) * a.phi[k];
            tmp += c.at * p0;
            double qn, an;
            double g0f = a.q2v[k];
            if (VAR == 3 && a.qk) g0f = g0f - c.tau * fbbf_cell(c, a.qk[k]);
            q_value<WEIGHTED, (VAR == 3 ? 2 : (VAR != 0 ? 1 : 0))>(c, tmp, g0f, c.c1, c.dinv1, k, a.weight, a.alpha_in, qn, an, u0prev, a.ap);
            if (KKT) {
                a0prev = an;
                rhoTprev = a.kappa * (wgt(k) * an);
            }
        }
    }
    // The march.  Every step is written in three phases -- all loads of the step (from clamped, always valid
    // addresses; the few that exist on tile / slab borders only sit under their condition but are loads and nothing
    // else), then the arithmetic, then the stores -- so that the loads leave together and are waited for once.  (With
    // a load, its use and a store inside one `if` per entry the step was five dependent memory round trips long.)
    constexpr int MULT = (VAR == 3 ? 2 : (VAR != 0 ? 1 : 0));
    const i64 yc = inb ? y : 0, xc = inb ? x : 0;
    const bool hasBx = inb && (x < g.nx - 1), hasBy = inb && (y < g.ny - 1);
    const bool leftTile = inb && (xl == 0) && (x >= 1);          // the bx edge on the left belongs to the tile there
    const bool belowTile = inb && (lane == 0) && (y >= 1);       // the by edge below belongs to the tile there
    const bool sxOwn = hasBx && ((x % fg.XB) == fg.XB - 1), syOwn = hasBy && ((y & 63) == 63);
    int par = 0;
    __syncthreads();                                              // ph[0] is complete
    for (i64 tl = t0; tl < t1; ++tl) {
        const i64 node = yc + g.py * (xc + g.nx * tl);
        const bool tbnd = (g.t0 + tl == 0) || (g.t0 + tl == g.nt - 1);
        const double dc = tbnd ? c.c2 : c.c1;
        const double di = tbnd ? c.dinv2 : c.dinv1;
        const bool hasCell = inb && (tl < g.ncl);
        const bool tails = (tl == 0) && !g.first;                // slab mode: the left neighbour's share of the first layer
        // KKT variant on a slab that is not the first: the sums of its first node / edge layer need the left neighbour's
        // last cell (alpha0 for A' alpha, the density for the momentum terms) -- they are left to a one-layer launch of
        // k_kkt after the exchange; the q0 entries of that layer need no neighbour and stay here
        const bool lay0 = KKT && tails;
        // ---------------- loads ----------------
        const i64 eX = hasBx ? g.offBx + g.bxLayer * tl + yc + g.py * xc : node;
        const i64 eY = hasBy ? g.offBy + g.byLayer * tl + yc + g.pyb * xc : node;
        const i64 k0 = hasCell ? node : 0;                       // q0 entries exist for tl < ncl only
        const i64 nodeT = hasCell ? node + g.plane : node;       // the layer of the next step (this one again at the end)
        const double pTl = a.phi[nodeT];
        double hx = 0.0, hl = 0.0, hy = 0.0, hb = 0.0;           // its halo strips
        if (rightCol) hx = a.phi[nodeT + g.py];
        if (leftTile) hl = a.phi[nodeT - g.py];
        if (topRow) hy = a.phi[nodeT + 1];
        if (belowTile) hb = a.phi[nodeT - 1];
        const double pXl = ph[par][xl + 2][lane + 1];
        const double pYl = ph[par][xl + 1][lane + 2];
        const double al0 = a.alpha_in[k0], alX = a.alpha_in[eX], alY = a.alpha_in[eY];
        double g0 = a.q2v[k0];
        double gX = a.q2v[eX], gY = a.q2v[eY];
        double k0v = 0.0, kXv = 0.0, kYv = 0.0;
        const bool pcorr = (VAR == 3) && (a.qk != nullptr);
        if (pcorr) { k0v = a.qk[k0]; kXv = a.qk[eX]; kYv = a.qk[eY]; }
        const double cv = a.cvec[node];
        double w0 = 1.0, wX = 1.0, wY = 1.0;
        if (WEIGHTED) { w0 = a.weight[k0]; wX = a.weight[eX]; wY = a.weight[eY]; }
        const double sxv = a.sx[sxOwn ? (tl * fg.nxblk + (x / fg.XB + 1)) * g.ny + y : 0];
        const double syv = a.sy[syOwn ? (tl * g.nx + x) * fg.nyblk + (y / 64 + 1) : 0];
        double tXv = 0.0, tYv = 0.0;
        if (tails) {
            if (hasBx) tXv = a.tail_bx[y + g.py * x];
            if (hasBy) tYv = a.tail_by[y + g.pyb * x];
        }
        // neighbour tiles' edges (first column / first row of the tile)
        double pLl = 0.0, alL = 0.0, gL = 0.0, wL = 1.0, sxL = 0.0, tLv = 0.0, kLv = 0.0, kBv = 0.0;
        i64 eL = 0;
        if (leftTile) {
            eL = g.offBx + g.bxLayer * tl + y + g.py * (x - 1);
            pLl = ph[par][0][lane + 1];
            alL = a.alpha_in[eL];
            gL = a.q2v[eL];
            if (pcorr) kLv = a.qk[eL];
            if (WEIGHTED) wL = a.weight[eL];
            if (((x - 1) % fg.XB) == fg.XB - 1) sxL = a.sx[(tl * fg.nxblk + ((x - 1) / fg.XB + 1)) * g.ny + y];
            if (tails) tLv = a.tail_bx[y + g.py * (x - 1)];
        }
        double pBl = 0.0, alB = 0.0, gB = 0.0, wB = 1.0, syB = 0.0, tBv = 0.0;
        i64 eB = 0;
        if (belowTile) {
            eB = g.offBy + g.byLayer * tl + (y - 1) + g.pyb * x;
            pBl = ph[par][xl + 1][0];
            alB = a.alpha_in[eB];
            gB = a.q2v[eB];
            if (pcorr) kBv = a.qk[eB];
            if (WEIGHTED) wB = a.weight[eB];
            if (((y - 1) & 63) == 63) syB = a.sy[(tl * g.nx + x) * fg.nyblk + ((y - 1) / 64 + 1)];
            if (tails) tBv = a.tail_by[(y - 1) + g.pyb * x];
        }
        // ---------------- arithmetic ----------------
        // the adjoint sums of an edge on a tile border are completed from the neighbour tile's partial (k_qstep_fused)
        if (sxOwn) gX = c.sf * (gX + sxv);
        if (syOwn) gY = c.sf * (gY + syv);
        if (tails) { gX += tXv; gY += tYv; }
        if (pcorr) {
            g0 = g0 - c.tau * fbbf_cell(c, k0v);
            gX = gX - c.tau * fbbf_edge(c, kXv, tbnd);
            gY = gY - c.tau * fbbf_edge(c, kYv, tbnd);
        }
        double pT = 0.0, u0 = 0.0, ubx = 0.0, uby = 0.0;
        double a0 = 0.0, abx = 0.0, aby = 0.0;                 // KKT: alpha^+ of the own entries
        double qbx = 0.0, mbx = 0.0, qby = 0.0, mby = 0.0;     // KKT: q^+ and momentum kappa (w alpha^+) of the own edges
        double rhoT = 0.0;                                     // KKT: density of the cell that starts at this node
        double q0n = 0.0, a0n = 0.0, ain0 = 0.0, qXn = 0.0, aXn = 0.0, ainX = 0.0, qYn = 0.0, aYn = 0.0, ainY = 0.0;
        if (hasCell) {
            pT = pTl;
            double tmp = (-c.at) * p0;
            tmp += c.at * pT;
            ain0 = q_calc<WEIGHTED, MULT>(c, tmp, g0, c.c1, c.dinv1, w0, al0, a.ap, q0n, a0n, u0);
            if (KKT) {
                entry(tmp, q0n, a0n, w0);
                a0 = a0n;
                rhoT = a.kappa * (w0 * a0n);
            }
        }
        if (hasBx) {
            double tmp = (-c.ax) * p0;
            tmp += c.ax * pXl;
            ainX = q_calc<WEIGHTED, MULT>(c, tmp, gX, dc, di, wX, alX, a.ap, qXn, aXn, ubx);
            if (KKT) {
                abx = aXn;
                qbx = qXn;
                if (!lay0) {
                    entry(tmp, qXn, aXn, wX);
                    mbx = a.kappa * (wX * aXn);
                    S[Q_M2] += mbx * mbx;
                }
            }
        }
        if (hasBy) {
            double tmp = (-c.ay) * p0;
            tmp += c.ay * pYl;
            ainY = q_calc<WEIGHTED, MULT>(c, tmp, gY, dc, di, wY, alY, a.ap, qYn, aYn, uby);
            if (KKT) {
                aby = aYn;
                qby = qYn;
                if (!lay0) {
                    entry(tmp, qYn, aYn, wY);
                    mby = a.kappa * (wY * aYn);
                    S[Q_M2] += mby * mby;
                }
            }
        }
        double ubx_l = 0.0, abx_l = 0.0, uby_b = 0.0, aby_b = 0.0;
        if (leftTile) {
            double q2 = gL;
            if (((x - 1) % fg.XB) == fg.XB - 1) q2 = c.sf * (q2 + sxL);
            if (tails) q2 += tLv;
            if (pcorr) q2 = q2 - c.tau * fbbf_edge(c, kLv, tbnd);
            double tmp = (-c.ax) * pLl;
            tmp += c.ax * p0;
            double qn, an;
            q_calc<WEIGHTED, MULT>(c, tmp, q2, dc, di, wL, alL, a.ap, qn, an, ubx_l);
            abx_l = an;
        }
        if (belowTile) {
            double q2 = gB;
            if (((y - 1) & 63) == 63) q2 = c.sf * (q2 + syB);
            if (tails) q2 += tBv;
            if (pcorr) q2 = q2 - c.tau * fbbf_edge(c, kBv, tbnd);
            double tmp = (-c.ay) * pBl;
            tmp += c.ay * p0;
            double qn, an;
            q_calc<WEIGHTED, MULT>(c, tmp, q2, dc, di, wB, alB, a.ap, qn, an, uby_b);
            aby_b = an;
        }
        // ---------------- stores ----------------
        if (hasCell) {
            put(node, q0n, a0n, ain0);
            if (VAR != 3 && a.u0_tail && tl == g.ncl - 1 && !g.last) a.u0_tail[y + g.py * x] = u0;
        }
        if (hasBx) put(eX, qXn, aXn, ainX);
        if (hasBy) put(eY, qYn, aYn, ainY);
        // density at the node: mean of the two cells that meet there in time, zero outside (movmean's padding)
        const double rhoN = (rhoTprev + rhoT) / 2.0;
        xch[par][xl][lane] = ubx;
        ph[par ^ 1][xl + 1][lane + 1] = pTl;
        if (xl == QTX - 1) ph[par ^ 1][QTX + 1][lane + 1] = hx;
        if (xl == 0) ph[par ^ 1][0][lane + 1] = hl;
        if (lane == TILE_Y - 1) ph[par ^ 1][xl + 1][TILE_Y + 1] = hy;
        if (lane == 0) ph[par ^ 1][xl + 1][0] = hb;
        if (KKT) {
            xcha[par][xl][lane] = abx;
            xchr[par][xl][lane] = rhoN;
        }
        __syncthreads();
        double uby_m = __shfl_up(uby, 1, 64);
        double aby_m = KKT ? __shfl_up(aby, 1, 64) : 0.0;
        const double rhoU = KKT ? __shfl_down(rhoN, 1, 64) : 0.0;
        if (inb) {
            double ubx_m = 0.0, abx_m = 0.0;
            if (x >= 1) {
                if (xl > 0) {
                    ubx_m = xch[par][xl - 1][lane];
                    if (KKT) abx_m = xcha[par][xl - 1][lane];
                } else {            // edge owned by the tile to the left
                    ubx_m = ubx_l;
                    abx_m = abx_l;
                }
            }
            if (belowTile) {        // edge owned by the tile below
                uby_m = uby_b;
                aby_m = aby_b;
            }
            double r = 0.0;
            if (tl >= 1) r += c.at * u0prev;
            if (tl < g.ncl) r += (-c.at) * u0;
            if (x >= 1) r += c.ax * ubx_m;
            if (x <= g.nx - 2) r += (-c.ax) * ubx;
            if (y >= 1) r += c.ay * uby_m;
            if (y <= g.ny - 2) r += (-c.ay) * uby;
            a.rhs[node] = r + cv;
            if (KKT && !lay0) {
                double ra = 0.0;                       // A' alpha^+ in the order of k_kkt's node part
                if (tl >= 1) ra += c.at * a0prev;
                if (tl < g.ncl) ra += (-c.at) * a0;
                if (x >= 1) ra += c.ax * abx_m;
                if (x <= g.nx - 2) ra += (-c.ax) * abx;
                if (y >= 1) ra += c.ay * aby_m;
                if (y <= g.ny - 2) ra += (-c.ay) * aby;
                ra = ra - cv;
                a.resid[node] = ra;
                S[Q_DUAL1] += ra * ra;
                S[Q_CPHI] += cv * p0;
                S[Q_PHI2] += p0 * p0;
                // compute_kkt_dot_complement.m:10-18: momentum against mean density times b, edges inside the tile
                if (x < g.nx - 1 && xl < QTX - 1) {
                    const double rm = (rhoN + xchr[par][xl + 1][lane]) / 2.0;
                    const double rb = a.dsD * (rm * qbx);
                    const double d = mbx - rb;
                    S[Q_MRHOB] += d * d;
                    S[Q_RHOB2] += rb * rb;
                }
                if (y < g.ny - 1 && lane < TILE_Y - 1) {
                    const double rm = (rhoN + rhoU) / 2.0;
                    const double rb = a.dsD * (rm * qby);
                    const double d = mby - rb;
                    S[Q_MRHOB] += d * d;
                    S[Q_RHOB2] += rb * rb;
                }
            }
        }
        u0prev = u0;
        if (KKT) {
            a0prev = a0;
            rhoTprev = rhoT;
        }
        p0 = pT;
        par ^= 1;
    }
    if (KKT) {
        // workgroup reduction as in k_kkt: wavefront shuffles, LDS across the four wavefronts, one partial row per workgroup
        __shared__ double red[QTX][Q_COUNT];
        static const int slot[Q_COUNT] = {S_Q2, S_ALPHA2, S_APHI2, S_PRIM1, S_QALPHA, S_CPHI, S_PHI2, S_DUAL1, S_MRHOB, S_M2, S_RHOB2};
#pragma unroll
        for (int i = 0; i < Q_COUNT; ++i) {
            double v = S[i];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[xl][i] = v;
        }
        __syncthreads();
        if (xl == 0 && lane < S_COUNT) {
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < Q_COUNT; ++i)
                if (slot[i] == lane) {
                    v = red[0][i];
#pragma unroll
                    for (int wv = 1; wv < QTX; ++wv) v += red[wv][i];
                }
            // one row per tile and CHUNK (a slab's q-step runs as several launches over disjoint sets of chunks)
            const i64 b = blk.x + (i64)gridDim.x * (blk.y + (i64)gridDim.y * ((i64)blk.z * a.zstride + a.z0));
            a.partials[b * S_COUNT + lane] = v;
        }
    }
}

static int launch_qstep_rhs_var(int var, const Grid &g, const LoopCoef &c, const FusedGeom &fg, QRhsArgs a, hipStream_t st,
                                i64 z0 = 0, i64 zcount = -1, i64 zstride = 1);

int launch_qstep_rhs(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi, const double *q2,
                     const double *sx, const double *sy, const double *weight, const double *tail_bx,
                     const double *tail_by, const double *cvec, double *q_out, const double *alpha_in, double *alpha_out,
                     double *rhs, hipStream_t st, i64 z0, i64 zcount, i64 zstride, const QStepExtra *ex) {
    QRhsArgs a{};
    a.phi = phi; a.q2v = q2; a.sx = sx; a.sy = sy; a.weight = weight; a.tail_bx = tail_bx; a.tail_by = tail_by;
    a.cvec = cvec; a.alpha_in = alpha_in; a.q_out = q_out; a.alpha_out = alpha_out; a.rhs = rhs;
    a.ap = APend{0, 1.0, 1.0};
    if (ex) {
        a.ap = APend{ex->apend, ex->amul, ex->adiv};
        a.partials = ex->partials;
        a.resid = ex->resid;
        a.u0_tail = ex->u0_tail;
        a.kappa = ex->kappa;
        a.dsD = ex->dsD;
    }
    return launch_qstep_rhs_var(0, g, c, fg, a, st, z0, zcount, zstride);
}

// blocks of the q-step launch: one row of partial sums each in the KKT variant
i64 qstep_rhs_blocks(const Grid &g, const FusedGeom &fg) { return fg.nyblk * fg.nxblk * qstep_rhs_chunks(g, fg); }

// After a sigma update (alpha, c <- / factor, solver_socp_inPALM.m:312-314) the right-hand side A'(w.*q - alpha) + c the
// q-step left behind becomes  A'(w.*q) - (A' alpha - c) / factor = (rhs + r) - r / factor  with the r = A' alpha - c the
// KKT variant of the q-step stored; c is divided on the way (alpha stays pending: APend).
__global__ void __launch_bounds__(256) k_rhs_sigma_fix(double *__restrict__ rhs, const double *__restrict__ r,
                                                        double *__restrict__ cvec, i64 n, double factor) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const double rv = r[i];
        rhs[i] = (rhs[i] + rv) - rv / factor;
        cvec[i] = cvec[i] / factor;
    }
}

int launch_rhs_sigma_fix(double *rhs, const double *r, double *cvec, i64 n, double factor, hipStream_t st) {
    if (n <= 0) return 0;
    DS_KLAUNCH(k_rhs_sigma_fix, dim3(launch_blocks(n, 256, 1 << 22)), dim3(256), 0, st, rhs, r, cvec, n, factor);
    DS_HIP(hipGetLastError());
    return 0;
}

// var 1 / 2: the acc-ADMM flavours (see k_qstep_rhs); `acc` carries the Halpern weights and the extra arrays of var 2
int launch_qstep_rhs_acc(int var, const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi,
                         const double *q2, const double *sx, const double *sy, const double *weight, const double *cvec,
                         double *q_raw, const double *alpha_in, double *alpha_out, double *rhs, double *q_state,
                         const double *q_anchor, const double *alpha_anchor, const AccCoef &k, hipStream_t st,
                         const double *tail_bx, const double *tail_by, double *u0_tail) {
    QRhsArgs a{};
    a.phi = phi; a.q2v = q2; a.sx = sx; a.sy = sy; a.weight = weight; a.cvec = cvec;
    a.tail_bx = tail_bx; a.tail_by = tail_by; a.u0_tail = u0_tail;
    a.alpha_in = alpha_in; a.q_out = q_raw; a.alpha_out = alpha_out; a.rhs = rhs;
    a.q_state = q_state; a.q_anchor = q_anchor; a.alpha_anchor = alpha_anchor;
    a.c1 = k.c1; a.c2 = k.c2; a.om_rho = k.om_rho; a.rho = k.rho;
    return launch_qstep_rhs_var(var, g, c, fg, a, st);
}

i64 qstep_rhs_chunks(const Grid &g, const FusedGeom &fg, i64 *TCout) {
    // short chunks of time layers (measured at 1024x1024x128: 3.45 ms with 8-layer chunks, 4.2 ms with one chunk per
    // tile -- the march is latency-bound per workgroup); each extra chunk recomputes one cell
    const i64 tiles = fg.nyblk * fg.nxblk;
    const i64 target = 32768;
    i64 chunks = (target + tiles - 1) / tiles;
    i64 TC = (g.ntl + chunks - 1) / chunks;
    if (TC < 8) TC = 8;
    // a slab of a time-slab decomposition: at least four chunks, so that the two in the middle -- which need neither
    // neighbour -- can run while the phi head and the adjoint tails travel (Solver::step)
    if (!(g.first && g.last) && cone_split_enabled() && g.ntl >= 12) {
        // ... the LAST chunk -- the only one that waits for the phi head of the right neighbour -- about a quarter of the
        // slab, the chunks in front of it up to eight layers each (16 layers: 6 + 6 + 4)
        const i64 tail = (g.ntl / 4 < 4) ? 4 : g.ntl / 4;
        const i64 body = g.ntl - tail, nb = (body + 7) / 8;
        TC = (body + nb - 1) / nb;
    }
    if (TC > g.ntl) TC = g.ntl;
    if (TC < 1) TC = 1;
    if (TCout) *TCout = TC;
    return (g.ntl + TC - 1) / TC;
}

static int launch_qstep_rhs_var(int var, const Grid &g, const LoopCoef &c, const FusedGeom &fg, QRhsArgs a, hipStream_t st,
                                i64 z0, i64 zcount, i64 zstride) {
    i64 TC = 1;
    const i64 chunks = qstep_rhs_chunks(g, fg, &TC);
    if (zcount < 0) zcount = chunks - z0;
    if (z0 < 0 || zcount <= 0 || zstride < 1 || z0 + (zcount - 1) * zstride >= chunks) return 0;
    a.TC = TC;
    a.z0 = z0;
    a.zstride = zstride;
    a.xcd = 1;
    dim3 grid((unsigned)fg.nyblk, (unsigned)fg.nxblk, (unsigned)zcount);
    dim3 blk(TILE_Y, TILE_X);
    // the plain inPALM instance may run on tiles twice as wide (the recomputed x - 1 edge and the phi halo columns cost
    // half as much); the KKT variant keeps the tile of k_kkt_bnd, which finishes the edges on ITS tile borders
    // (1024 x 1024 x 128: 18.35 -> 17.6 GB per launch by the PMC counters, same time; small grids keep the narrow tile:
    // they need the workgroup count more than the bytes)
    const char *qe = getenv("DOTSOCP_QTX");                   // read per launch: the tests switch it inside one process
    const int qtx_env = qe ? atoi(qe) : 0;
    const int qtx = qtx_env ? qtx_env : ((fg.nyblk * fg.nxblk * zcount >= 8192 && !a.weight) ? 2 * TILE_X : TILE_X);
    if (var == 0 && !a.partials && qtx == 2 * TILE_X) {
        dim3 grid2((unsigned)fg.nyblk, (unsigned)((g.nx + 2 * TILE_X - 1) / (2 * TILE_X)), (unsigned)zcount);
        dim3 blk2(TILE_Y, 2 * TILE_X);
        if (a.weight) DS_KLAUNCH((k_qstep_rhs<true, 0, false, 2 * TILE_X>), grid2, blk2, 0, st, g, c, fg, a);
        else DS_KLAUNCH((k_qstep_rhs<false, 0, false, 2 * TILE_X>), grid2, blk2, 0, st, g, c, fg, a);
        DS_HIP(hipGetLastError());
        return 0;
    }
#define QRHS_LAUNCH(W, V) DS_KLAUNCH((k_qstep_rhs<W, V>), grid, blk, 0, st, g, c, fg, a)
    if (var == 0 && a.partials) {          // iteration with a KKT check
        if (a.weight) DS_KLAUNCH((k_qstep_rhs<true, 0, true>), grid, blk, 0, st, g, c, fg, a);
        else DS_KLAUNCH((k_qstep_rhs<false, 0, true>), grid, blk, 0, st, g, c, fg, a);
    } else if (a.weight) {
        if (var == 0) QRHS_LAUNCH(true, 0); else if (var == 1) QRHS_LAUNCH(true, 1); else QRHS_LAUNCH(true, 2);
    } else {
        if (var == 0) QRHS_LAUNCH(false, 0); else if (var == 1) QRHS_LAUNCH(false, 1);
        else if (var == 2) QRHS_LAUNCH(false, 2); else QRHS_LAUNCH(false, 3);
    }
#undef QRHS_LAUNCH
    DS_HIP(hipGetLastError());
    return 0;
}

// time-slab mode: rhs(:, :, first layer) += (D/ht) * u0 of the left neighbour's last cell
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_rhs_fixup(Grid g, double at, const double *__restrict__ u0_prev,
                                                               double *__restrict__ rhs) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    if (y >= g.ny || x >= g.nx) return;
    const i64 i = y + g.py * x;
    rhs[i] = rhs[i] + at * u0_prev[i];
}

int launch_rhs_fixup(const Grid &g, const LoopCoef &c, const double *u0_prev, double *rhs, hipStream_t st) {
    DS_KLAUNCH(k_rhs_fixup, tile_grid(g, 1), dim3(TILE_Y, TILE_X), 0, st, g, c.at, u0_prev, rhs);
    DS_HIP(hipGetLastError());
    return 0;
}

// PALM's first q-step (solver_socp_PALM.m:196-200): q_out = (A phi + alpha + q2) .* diagQInv, alpha untouched,
// plus the rhs of the phi-step that follows it (:204), A'(q_out - alpha) + c
int launch_qstep_palm_first(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi, const double *q2,
                            const double *sx, const double *sy, const double *cvec, double *q_out, const double *alpha,
                            double *rhs, hipStream_t st, const double *tail_bx, const double *tail_by, const double *qk) {
    QRhsArgs a{};
    a.phi = phi; a.q2v = q2; a.sx = sx; a.sy = sy; a.cvec = cvec; a.tail_bx = tail_bx; a.tail_by = tail_by;
    a.alpha_in = alpha; a.q_out = q_out; a.rhs = rhs; a.qk = qk;
    return launch_qstep_rhs_var(3, g, c, fg, a, st);
}

// tmp_q = A phi in q layout (solver_socp_PALM.m:137): forward differences times D/h, like the q-step
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_grad(Grid g, LoopCoef c, const double *__restrict__ phi,
                                                          double *__restrict__ out) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = blockIdx.z;
    if (y >= g.ny || x >= g.nx) return;
    const i64 node = y + g.py * (x + g.nx * tl);
    const double p0 = phi[node];
    if (tl < g.ncl) {
        double tmp = (-c.at) * p0;
        tmp += c.at * phi[node + g.plane];
        out[node] = tmp;
    }
    if (x < g.nx - 1) {
        double tmp = (-c.ax) * p0;
        tmp += c.ax * phi[node + g.py];
        out[g.offBx + g.bxLayer * tl + y + g.py * x] = tmp;
    }
    if (y < g.ny - 1) {
        double tmp = (-c.ay) * p0;
        tmp += c.ay * phi[node + 1];
        out[g.offBy + g.byLayer * tl + y + g.pyb * x] = tmp;
    }
}

int launch_grad(const Grid &g, const LoopCoef &c, const double *phi, double *out, hipStream_t st) {
    DS_KLAUNCH(k_grad, tile_grid(g, g.ntl), dim3(TILE_Y, TILE_X), 0, st, g, c, phi, out);
    DS_HIP(hipGetLastError());
    return 0;
}

// Time-slab mode: the fused cone kernel leaves, in the halo layer (index ncl) of q2 and of the side
// buffers, the adjoint sums that the LAST owned cell contributes to the first edge layer of the
// right neighbour.  This kernel completes them (tile-boundary edges) into two contiguous planes.
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_tail_finalize(Grid g, LoopCoef c, FusedGeom fg,
                                                                   const double *__restrict__ q2v,
                                                                   const double *__restrict__ sx,
                                                                   const double *__restrict__ sy,
                                                                   double *__restrict__ tail_bx,
                                                                   double *__restrict__ tail_by) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = g.ncl;
    if (y < g.ny && x < g.nx - 1) {
        double v = q2v[g.offBx + g.bxLayer * tl + y + g.py * x];
        if ((x % fg.XB) == fg.XB - 1) v = c.sf * (v + sx[(tl * fg.nxblk + (x / fg.XB + 1)) * g.ny + y]);
        tail_bx[y + g.py * x] = v;
    }
    if (y < g.ny - 1 && x < g.nx) {
        double v = q2v[g.offBy + g.byLayer * tl + y + g.pyb * x];
        if ((y & 63) == 63) v = c.sf * (v + sy[(tl * g.nx + x) * fg.nyblk + (y / 64 + 1)]);
        tail_by[y + g.pyb * x] = v;
    }
}

int launch_tail_finalize(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *q2, const double *sx,
                         const double *sy, double *tail_bx, double *tail_by, hipStream_t st) {
    DS_KLAUNCH(k_tail_finalize, tile_grid(g, 1), dim3(TILE_Y, TILE_X), 0, st, g, c, fg, q2, sx, sy, tail_bx,
                       tail_by);
    DS_HIP(hipGetLastError());
    return 0;
}

// Time-slab mode, KKT block: what the right neighbour needs from this slab's LAST cell layer --
// alpha0, w.*alpha0 and the raw partial adjoint sums of beta (cone columns 4,5 / 8,9).
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_kkt_tail(Grid g, const double *__restrict__ alpha,
                                                              const double *__restrict__ beta,
                                                              const double *__restrict__ weight,
                                                              double *__restrict__ a0, double *__restrict__ a0w,
                                                              double *__restrict__ bt_bx, double *__restrict__ bt_by) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = g.ncl - 1;
    if (y < g.ny && x < g.nx) {
        const i64 cidx = y + g.py * (x + g.nx * tl);
        const double a = alpha[cidx];
        a0[y + g.py * x] = a;
        a0w[y + g.py * x] = weight ? weight[cidx] * a : a;
    }
    if (y < g.ny && x < g.nx - 1) {
        double acc = beta[3 * g.Nc + y + g.py * ((x + 1) + g.nx * tl)];
        acc += beta[4 * g.Nc + y + g.py * (x + g.nx * tl)];
        bt_bx[y + g.py * x] = acc;
    }
    if (y < g.ny - 1 && x < g.nx) {
        double acc = beta[7 * g.Nc + (y + 1) + g.py * (x + g.nx * tl)];
        acc += beta[8 * g.Nc + y + g.py * (x + g.nx * tl)];
        bt_by[y + g.pyb * x] = acc;
    }
}

int launch_kkt_tail(const Grid &g, const double *alpha, const double *beta, const double *weight, double *a0,
                    double *a0w, double *bt_bx, double *bt_by, hipStream_t st) {
    DS_KLAUNCH(k_kkt_tail, tile_grid(g, 1), dim3(TILE_Y, TILE_X), 0, st, g, alpha, beta, weight, a0, a0w, bt_bx,
                       bt_by);
    DS_HIP(hipGetLastError());
    return 0;
}

// One-process-per-slab transposes of the Poisson solve: the slab rows [t][col] <-> the send / receive staging
// area in which the columns of every peer's pencil are contiguous, [peer][t][col - cut(peer)], in ONE launch
// (cut[j] .. cut[j+1] are the columns of pencil j; the area of peer j starts at cut[j] * ntl).
template <bool PACK>
__global__ void __launch_bounds__(256) k_pencil_pack(PencilCuts pc, i64 plane, i64 ntl, double *__restrict__ slab,
                                                      double *__restrict__ stage) {
    const i64 col = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 t = blockIdx.y;
    if (col >= plane) return;
    int j = (int)((col * pc.world) / plane);                 // first guess, then walk to the owning pencil
    while (j > 0 && col < pc.cut[j]) --j;
    while (j < pc.world - 1 && col >= pc.cut[j + 1]) ++j;
    const i64 c0 = pc.cut[j], w = pc.cut[j + 1] - c0;
    const i64 si = c0 * ntl + t * w + (col - c0);
    if (PACK) stage[si] = slab[t * plane + col];
    else slab[t * plane + col] = stage[si];
}

int launch_pencil_pack(bool pack, const PencilCuts &pc, i64 plane, i64 ntl, double *slab, double *stage, hipStream_t st) {
    if (plane * ntl <= 0) return 0;
    dim3 grid((unsigned)((plane + 255) / 256), (unsigned)ntl);
    if (pack) DS_KLAUNCH(k_pencil_pack<true>, grid, dim3(256), 0, st, pc, plane, ntl, slab, stage);
    else DS_KLAUNCH(k_pencil_pack<false>, grid, dim3(256), 0, st, pc, plane, ntl, slab, stage);
    DS_HIP(hipGetLastError());
    return 0;
}

// x = x * mul / div  (left to right, like `alpha * dScale2 / cScale2^2`, solver_socp_inPALM.m:170-178,312-314)
__global__ void __launch_bounds__(256) k_scale(double *__restrict__ x, i64 n, double mul, double div, int use_mul) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        double v = x[i];
        if (use_mul) v = v * mul;
        x[i] = v / div;
    }
}

int launch_scale(double *x, i64 n, double mul, double div, hipStream_t st) {
    if (n <= 0) return 0;
    DS_KLAUNCH(k_scale, dim3(launch_blocks(n, 256, 1 << 22)), dim3(256), 0, st, x, n, mul, div,
                       (int)(mul != 1.0));
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
