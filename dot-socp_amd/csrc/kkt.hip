// KKT block of the loop (solver_socp_inPALM.m:220-267, compute_kkt_dot_complement.m:2-18;
// weighted: solver_wsocp_inPALM.m:232-272, wdot2d/utils/compute_kkt_dot_complement.m:2-19):
// every global sum the residuals need, produced by ONE pass over the state.
// Each thread owns a (y, x) column of a time chunk and visits, per layer, its node, its cell and
// its two staggered edges; 20 running sums stay in registers, are reduced over the wavefront
// with cross-lane shuffles, over the workgroup through LDS, and written as per-block partials;
// a second kernel adds the partials in a fixed order (deterministic, no atomics).
#include "device_utils.h"
#include "gather_tile.h"
#include "kernels.h"

#include <cstdlib>

namespace dotsocp {

struct WBeta {
    const double *b;
    i64 Nz;
    __device__ __forceinline__ double operator()(int j, i64 cell) const { return b[j * Nz + cell]; }
};

// PART selects which entries a launch visits (1 node, 2 cell + its q0 entry, 4 bx edge, 8 by edge):
// four lean launches keep the register count low and the occupancy high.
// NODUAL (acc-ADMM's folded check: the cone pass has taken ||F*B*beta||^2 and ||F*B*beta + w.*alpha||^2 of every entry): the
// F*B*beta terms are left out and beta is not read; PART 16 then is the q0 entries' share of the staggered sums.
template <bool WEIGHTED, int PART, bool NODUAL = false>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_kkt(Grid g, LoopCoef c, KktCoef k, i64 chunk,
                                                         const double *__restrict__ phi, const double *__restrict__ q,
                                                         const double *__restrict__ alpha, const double *__restrict__ z,
                                                         const double *__restrict__ beta,
                                                         const double *__restrict__ cvec,
                                                         const double *__restrict__ weight,
                                                         KktHalo halo, double *__restrict__ partials,
                                                         double *__restrict__ resid) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const bool inb = (y < g.ny) && (x < g.nx);
    const i64 tbeg = (i64)blockIdx.z * chunk;
    const i64 tend = (tbeg + chunk < g.ntl) ? tbeg + chunk : g.ntl;
    double S[S_COUNT];
#pragma unroll
    for (int i = 0; i < S_COUNT; ++i) S[i] = 0.0;
    WBeta WB{beta, g.Nc};
    auto wgt = [&](i64 idx) { return WEIGHTED ? weight[idx] : 1.0; };
    // rhoT of a cell = kappa * (w .* alpha)_0 ; 0 outside the time range (zero padding of movmean)
    auto rhoT_at = [&](i64 yy, i64 xx, i64 tl) -> double {
        if (tl >= g.ncl) return 0.0;
        if (tl < 0) {
            if (g.first) return 0.0;
            return k.kappa * halo.a0w_prev[yy + g.py * xx];
        }
        const i64 cidx = yy + g.py * (xx + g.nx * tl);
        return k.kappa * (wgt(cidx) * alpha[cidx]);
    };
    auto rho_node = [&](i64 yy, i64 xx, i64 tl) { return (rhoT_at(yy, xx, tl - 1) + rhoT_at(yy, xx, tl)) / 2.0; };
    auto edge_sums = [&](i64 idx, double tmp, double q2b) {
        const double w = wgt(idx);
        const double qv = q[idx], av = alpha[idx];
        const double wq = w * qv, wa = w * av;
        S[S_Q2] += qv * qv;
        S[S_ALPHA2] += av * av;
        S[S_APHI2] += tmp * tmp;
        const double r1 = tmp - wq;
        S[S_PRIM1] += r1 * r1;
        if (!NODUAL) {
            S[S_FBBETA2] += q2b * q2b;
            const double r2 = q2b + wa;
            S[S_DUAL2] += r2 * r2;
        }
        S[S_QALPHA] += wq * av;
    };
    if (inb) {
        for (i64 tl = tbeg; tl < tend; ++tl) {
            const i64 node = y + g.py * (x + g.nx * tl);
            const i64 bxo = g.offBx + g.bxLayer * tl, byo = g.offBy + g.byLayer * tl;
            // ---- node: A' alpha - c, <c, phi>, ||phi||^2 ----
            if (PART & 1) {
                double r = 0.0;
                if (tl >= 1)
                    r += c.at * alpha[node - g.plane];
                else if (!g.first)
                    r += c.at * halo.a0_prev[y + g.py * x];
                if (tl < g.ncl) r += (-c.at) * alpha[node];
                if (x >= 1) r += c.ax * alpha[bxo + y + g.py * (x - 1)];
                if (x <= g.nx - 2) r += (-c.ax) * alpha[bxo + y + g.py * x];
                if (y >= 1) r += c.ay * alpha[byo + (y - 1) + g.pyb * x];
                if (y <= g.ny - 2) r += (-c.ay) * alpha[byo + y + g.pyb * x];
                const double cv = cvec[node], pv = phi[node];
                r = r - cv;
                // A' alpha - c of the node for launch_rhs_sigma_fix.  The right-hand side it corrects is the q-step's, which
                // on a slab's first layer still lacks the left neighbour's cell (k_rhs_fixup adds it later, from the
                // scaled alpha): the same term is left out here
                if (resid) resid[node] = (tl == 0 && !g.first) ? r - c.at * halo.a0_prev[y + g.py * x] : r;
                S[S_DUAL1] += r * r;
                S[S_CPHI] += cv * pv;
                S[S_PHI2] += pv * pv;
            }
            // ---- cell ----
            if ((PART & 2) && tl < g.ncl) {
                const EdgeQuad ea = load_edges(g, q, y, x, tl, c.sf);
                const EdgeQuad eb = load_edges(g, q, y, x, tl + 1, c.sf);
                const double q0 = q[node];
                double z2[10], zz[10], p[10];
                build_z2(z2, q0, ea, eb, c.s, c.dF);
                double zs = 0.0, bs = 0.0, rs = 0.0;
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const double zv = z[j * g.Nc + node], bv = beta[j * g.Nc + node];
                    zz[j] = zv;
                    p[j] = zv - k.sigma * bv;
                    zs += zv * zv;
                    bs += bv * bv;
                    const double d = zv - z2[j];
                    rs += d * d;
                }
                proj_row<10>(p);
                double cs = 0.0;
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const double d = zz[j] - p[j];
                    cs += d * d;
                }
                S[S_Z2] += zs;
                S[S_BETA2] += bs;
                S[S_PRIM2] += rs;
                S[S_COMPLEM] += cs;
                const double rhoT = rhoT_at(y, x, tl);
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
                // q0 entry of q
                double tmp = (-c.at) * phi[node];
                tmp += c.at * phi[node + g.plane];
                edge_sums(node, tmp, c.s * (WB(9, node) - WB(0, node)));
            }
            if ((PART & 16) && tl < g.ncl) {            // q0 entry alone (NODUAL)
                double tmp = (-c.at) * phi[node];
                tmp += c.at * phi[node + g.plane];
                edge_sums(node, tmp, 0.0);
            }
            // ---- bx edge (y, x+1/2, t) ----
            if ((PART & 4) && x < g.nx - 1) {
                const i64 idx = bxo + y + g.py * x;
                double tmp = (-c.ax) * phi[node];
                tmp += c.ax * phi[node + g.py];
                const double q2b = NODUAL ? 0.0 : c.sf * gather_bx(g, WB, y, x, tl, halo.btail_bx);
                edge_sums(idx, tmp, q2b);
                const double rm = (rho_node(y, x, tl) + rho_node(y, x + 1, tl)) / 2.0;
                const double rb = k.dsD * (rm * q[idx]);
                const double m = k.kappa * (wgt(idx) * alpha[idx]);
                const double d = m - rb;
                S[S_MRHOB] += d * d;
                S[S_M2] += m * m;
                S[S_RHOB2] += rb * rb;
            }
            // ---- by edge (y+1/2, x, t) ----
            if ((PART & 8) && y < g.ny - 1) {
                const i64 idx = byo + y + g.pyb * x;
                double tmp = (-c.ay) * phi[node];
                tmp += c.ay * phi[node + 1];
                const double q2b = NODUAL ? 0.0 : c.sf * gather_by(g, WB, y, x, tl, halo.btail_by);
                edge_sums(idx, tmp, q2b);
                const double rm = (rho_node(y, x, tl) + rho_node(y + 1, x, tl)) / 2.0;
                const double rb = k.dsD * (rm * q[idx]);
                const double m = k.kappa * (wgt(idx) * alpha[idx]);
                const double d = m - rb;
                S[S_MRHOB] += d * d;
                S[S_M2] += m * m;
                S[S_RHOB2] += rb * rb;
            }
        }
    }
    // ---- workgroup reduction: wavefront shuffles, then LDS across the 4 wavefronts ----
    __shared__ double red[TILE_X][S_COUNT];
    const int lane = threadIdx.x;   // blockDim.x == 64 == one wavefront per threadIdx.y
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
        for (int wv = 1; wv < TILE_X; ++wv) v += red[wv][lane];
        const i64 b = blockIdx.x + (i64)gridDim.x * (blockIdx.y + (i64)gridDim.y * blockIdx.z);
        partials[b * S_COUNT + lane] = v;
    }
}

// Fused path: the pending multiplier step and the cell part of the KKT sums in ONE pass
// (instead of materialising beta and z and reading them back): per cell, with q_old = q^k,
// q = q^{k+1}, beta_in = beta^k,
//   z      = Pi_Q(BF q_old + d - beta_in)          (:199, recomputed)
//   beta'  = beta_in + tau (z - (BF q + d))        (:212-215)  -> beta_out
//   sums of z^2, beta'^2, (z - BFq - d)^2, (z - Pi_Q(z - sigma beta'))^2, the DOT complementarity terms
//   of compute_kkt_dot_complement.m:2-9, and the q0 entries' share of the edge sums.
// t-marching like k_cone_fused; z is not written (MODE_Z of k_cone_fused regenerates it on demand
// from the kept beta_in).
//
// EDGES = true (one slab, iterations whose q-step ran in its KKT variant): the kernel also forms the adjoint gather
// F* B* beta' that ||F*B*beta||^2 and ||F*B*beta + w.*alpha||^2 (:225,233-236) need -- beta' is in registers here, so the
// edge passes of k_kkt never read beta again.  The gather is the cone kernel's (gather_tile.h: x-neighbours through
// LDS, y-neighbours by lane shuffle, "t+1" entries of the previous cell carried in registers); for an edge inside the
// tile the two sums are taken on the spot (one load of alpha), for an edge on the tile's right / upper border the raw
// partial sums go to q2 / sx / sy exactly as in the cone pass and k_kkt_bnd completes them.  The q0 entries' share of
// ||q||^2, ||alpha||^2, ... is left to the q-step in this mode.
//
// NORMS = true: only ||z||^2 and ||beta'||^2 of the iterate whose multiplier step is still pending, nothing stored (the
// rescale block's every-100-iterations check, solver_socp_inPALM.m:139-149, when no KKT check precedes it).
template <bool WEIGHTED, bool EDGES, bool NORMS = false, bool NT = false>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_kkt_cells(Grid g, LoopCoef c, KktCoef k, FusedArgs a,
                                                               const double *__restrict__ phi,
                                                               const double *__restrict__ alpha,
                                                               const double *__restrict__ weight,
                                                               double *__restrict__ partials) {
    __shared__ double2 xch[EDGES ? 2 : 1][EDGES ? TILE_X : 1][EDGES ? 64 : 1];
    const int lane = threadIdx.x, xl = threadIdx.y;
    const i64 y = (i64)blockIdx.x * TILE_Y + lane;
    const i64 x = (i64)blockIdx.y * TILE_X + xl;
    const bool inb = (y < g.ny) && (x < g.nx);
    const i64 yc = inb ? y : 0, xc = inb ? x : 0;      // clamped coordinates keep out-of-tile lanes harmless
    const i64 t0 = (i64)blockIdx.z * a.TC;
    const i64 t1 = (t0 + a.TC < g.ncl) ? t0 + a.TC : g.ncl;
    const bool lastChunk = (t1 == g.ncl);
    // the gather needs the "t+1" entries of the cell in front of the chunk (recomputed, not stored) and one virtual
    // step behind the last cell that emits the final edge layer
    const i64 tstart = (EDGES && t0 > 0) ? t0 - 1 : t0;
    const i64 tstop = (EDGES && lastChunk) ? t1 + 1 : t1;
    const i64 nxblk = gridDim.y, nyblk = gridDim.x;
    double S[S_COUNT];
#pragma unroll
    for (int i = 0; i < S_COUNT; ++i) S[i] = 0.0;
    auto wgt = [&](i64 idx) { return WEIGHTED ? weight[idx] : 1.0; };
    double p3 = 0.0, p4 = 0.0, p7 = 0.0, p8 = 0.0;     // beta' columns 4, 5, 8, 9 of the previous cell
    int par = 0;
    if (inb || EDGES) {
        EdgeQuad cur = load_edges(g, a.q, yc, xc, tstart, c.sf);
        EdgeQuad curo = load_edges(g, a.q_old, yc, xc, tstart, c.sf);
        for (i64 tl = tstart; tl < tstop; ++tl) {
            const bool hasCell = tl < g.ncl;
            const bool own = (tl >= t0) && inb;
            double b[10];
            if (hasCell) {
                const i64 i = yc + g.py * (xc + g.nx * tl);
                const EdgeQuad nxt = load_edges(g, a.q, yc, xc, tl + 1, c.sf);
                const EdgeQuad nxto = load_edges(g, a.q_old, yc, xc, tl + 1, c.sf);
                const double q0 = a.q[i];
                double v[10], zo[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) b[j] = ld_stream<NT>(a.beta_in + j * g.Nc + i);
                if (a.bpend) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) b[j] = b[j] * a.bmul / a.bdiv;
                    if (a.bpend > 1) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) b[j] = b[j] * a.bmul2 / a.bdiv2;
                    }
                }
                build_z2(v, q0, cur, nxt, c.s, c.dF);
                build_z2(zo, a.q_old[i], curo, nxto, c.s, c.dF);
#pragma unroll
                for (int j = 0; j < 10; ++j) zo[j] = zo[j] - b[j];
                proj_row<10>(zo);
                double zs = 0.0, bs = 0.0, rs = 0.0;
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const double r = zo[j] - v[j];
                    b[j] = b[j] + c.tau * r;
                    zs += zo[j] * zo[j];
                    bs += b[j] * b[j];
                    rs += r * r;
                }
                cur = nxt;
                curo = nxto;
                if (own && NORMS) {
                    S[S_Z2] += zs;
                    S[S_BETA2] += bs;
                }
                if (own && !NORMS) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) st_stream<NT>(a.beta_out + j * g.Nc + i, b[j]);
                    // ||z - Pi_Q(z - sigma beta')||^2 (:240-241) with proj_row's arithmetic, the projected row never
                    // stored: x = z - sigma beta' is cheap to form twice, ten registers are not
                    auto xj = [&](int j) { return zo[j] - k.sigma * b[j]; };
                    double nn = xj(1) * xj(1);
#pragma unroll
                    for (int j = 2; j < 10; ++j) nn += xj(j) * xj(j);
                    const double n = sqrt(nn), x0 = xj(0);
                    double cf = (x0 / n + 1.0) * 0.5;
                    cf = (cf > 1.0) ? 1.0 : cf;
                    cf = (cf < 0.0) ? 0.0 : cf;
                    const double p0 = (cf >= 1.0) ? x0 : cf * n;
                    double cs = (zo[0] - p0) * (zo[0] - p0);
#pragma unroll
                    for (int j = 1; j < 10; ++j) {
                        const double d = zo[j] - cf * xj(j);
                        cs += d * d;
                    }
                    S[S_Z2] += zs;
                    S[S_BETA2] += bs;
                    S[S_PRIM2] += rs;
                    S[S_COMPLEM] += cs;
                    const double w = wgt(i);
                    const double av = alpha[i];
                    const double rhoT = k.kappa * (w * av);
                    double sq = 0.0;
#pragma unroll
                    for (int j = 1; j < 9; ++j) {
                        const double e = k.dsE * v[j];
                        sq += e * e;
                    }
                    double rhoFq = rhoT + k.dsD * q0 + sq / 4.0;
                    rhoFq = (rhoFq < 0.0) ? 0.0 : rhoFq;
                    const double dd = rhoT - rhoFq;
                    S[S_DOTCOMP] += dd * dd;
                    S[S_RHO2] += rhoT * rhoT;
                    S[S_RHOFQ2] += rhoFq * rhoFq;
                    // the q0 entry of q (same terms as the staggered edges in k_kkt)
                    const double q2b = c.s * (b[9] - b[0]);
                    const double wa = w * av;
                    S[S_FBBETA2] += q2b * q2b;
                    const double r2 = q2b + wa;
                    S[S_DUAL2] += r2 * r2;
                    if (!EDGES) {
                        double tmp = (-c.at) * phi[i];
                        tmp += c.at * phi[i + g.plane];
                        const double wq = w * q0;
                        S[S_Q2] += q0 * q0;
                        S[S_ALPHA2] += av * av;
                        S[S_APHI2] += tmp * tmp;
                        const double r1 = tmp - wq;
                        S[S_PRIM1] += r1 * r1;
                        S[S_QALPHA] += wq * av;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 10; ++j) b[j] = 0.0;
            }
            if (EDGES) {
                // edge layer tl of F* B* beta' (gather_emit's arithmetic and order)
                auto dual = [&](i64 e, double acc) {
                    const double gb = c.sf * acc;
                    S[S_FBBETA2] += gb * gb;
                    const double r2 = gb + wgt(e) * alpha[e];
                    S[S_DUAL2] += r2 * r2;
                };
                xch[par][xl][lane] = make_double2(b[1], p3);
                __syncthreads();
                // edge layers this launch sums over: the slab's own ones, except the first of a slab that is not the
                // first (the left neighbour's tail is missing here: k_kkt's one-layer launch takes that layer)
                const bool sumLayer = (tl < g.ntl) && !(tl == 0 && !g.first);
                if (own) {
                    if (x < g.nx - 1) {
                        const i64 e = g.offBx + g.bxLayer * tl + y + g.py * x;
                        if (xl < TILE_X - 1) {
                            if (sumLayer) {
                            const double2 r = xch[par][xl + 1][lane];
                            double acc = r.x + b[2];
                            acc += r.y;
                            acc += p4;
                            dual(e, acc);
                            }
                        } else {
                            a.q2[e] = b[2] + p4;             // partial; the right tile's part arrives in sx
                        }
                    }
                    if (xl == 0 && x > 0) a.sx[(tl * nxblk + blockIdx.y) * g.ny + y] = b[1] + p3;
                }
                const double u5 = __shfl_down(b[5], 1, 64), u7 = __shfl_down(p7, 1, 64);
                if (own) {
                    if (y < g.ny - 1) {
                        const i64 e = g.offBy + g.byLayer * tl + y + g.pyb * x;
                        if (lane < 63) {
                            if (sumLayer) {
                                double acc = u5 + b[6];
                                acc += u7;
                                acc += p8;
                                dual(e, acc);
                            }
                        } else {
                            a.q2[e] = b[6] + p8;             // partial; the upper tile's part arrives in sy
                        }
                    }
                    if (lane == 0 && y > 0) a.sy[(tl * g.nx + x) * nyblk + blockIdx.x] = b[5] + p7;
                }
                p3 = b[3]; p4 = b[4]; p7 = b[7]; p8 = b[8];
                par ^= 1;
            }
        }
    }
    __shared__ double red[TILE_X][S_COUNT];
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
        for (int wv = 1; wv < TILE_X; ++wv) v += red[wv][lane];
        const i64 bb = blockIdx.x + (i64)gridDim.x * (blockIdx.y + (i64)gridDim.y * blockIdx.z);
        partials[bb * S_COUNT + lane] = v;
    }
}

// sum of squares of n doubles, one partial row (slot `slot` of S_COUNT) per workgroup
__global__ void __launch_bounds__(256) k_sumsq(const double *__restrict__ x, i64 n, int slot, double *__restrict__ partials) {
    double v = 0.0;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) v += x[i] * x[i];
    __shared__ double red[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < S_COUNT)
        partials[(i64)blockIdx.x * S_COUNT + threadIdx.x] = ((int)threadIdx.x == slot) ? ((red[0] + red[1]) + red[2]) + red[3] : 0.0;
}

// The edges on tile borders (one slab).  DIR 0: bx edges (y, x+1/2, t) with x % TILE_X == TILE_X - 1 -- their right density
// node and the x+1 cells of their adjoint gather live in the next tile; DIR 1: by edges (y+1/2, x, t) with y % 64 == 63.
// Per edge: the momentum terms the q-step's KKT variant left out, and ||F*B*beta||^2, ||F*B*beta + w.*alpha||^2 from the raw
// partial sums k_kkt_cells<., true> left in q2 / sx / sy.  alpha, q: the new iterates in memory.
#define BND_TC 8
template <bool WEIGHTED, int DIR, bool NOMOM = false>
__global__ void __launch_bounds__(256) k_kkt_bnd(Grid g, LoopCoef c, KktCoef k, FusedGeom fg, const double *__restrict__ q,
                                                  const double *__restrict__ alpha, const double *__restrict__ weight,
                                                  const double *__restrict__ q2, const double *__restrict__ sx,
                                                  const double *__restrict__ sy, double *__restrict__ partials) {
    auto wgt = [&](i64 idx) { return WEIGHTED ? weight[idx] : 1.0; };
    auto rhoT_at = [&](i64 yy, i64 xx, i64 tl) -> double {
        if (tl < 0 || tl >= g.ncl) return 0.0;
        const i64 cidx = yy + g.py * (xx + g.nx * tl);
        return k.kappa * (wgt(cidx) * alpha[cidx]);
    };
    auto rho_node = [&](i64 yy, i64 xx, i64 tl) { return (rhoT_at(yy, xx, tl - 1) + rhoT_at(yy, xx, tl)) / 2.0; };
    double sM = 0.0, sR = 0.0, sF = 0.0, sD = 0.0;
    // DIR 0: threads along y, blockIdx.y = border column; DIR 1: threads along x, blockIdx.y = border row;
    // blockIdx.z: a chunk of BND_TC edge layers
    const i64 u = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 tbeg = (i64)blockIdx.z * BND_TC, tend = (tbeg + BND_TC < g.ntl) ? tbeg + BND_TC : g.ntl;
    for (i64 tl = tbeg; tl < tend; ++tl) {
    if (tl == 0 && !g.first) continue;            // time slabs: that layer belongs to k_kkt's one-layer launch
    if (DIR == 0) {
        const i64 y = u, x = (i64)blockIdx.y * TILE_X + (TILE_X - 1);
        if (y < g.ny && x < g.nx - 1) {
            const i64 e = g.offBx + g.bxLayer * tl + y + g.py * x;
            if (!NOMOM) {
                const double rm = (rho_node(y, x, tl) + rho_node(y, x + 1, tl)) / 2.0;
                const double rb = k.dsD * (rm * q[e]);
                const double m = k.kappa * (wgt(e) * alpha[e]);
                const double d = m - rb;
                sM += d * d;
                sR += rb * rb;
            }
            const double gb = c.sf * (q2[e] + sx[(tl * fg.nxblk + (x / TILE_X + 1)) * g.ny + y]);
            sF += gb * gb;
            const double r2 = gb + wgt(e) * alpha[e];
            sD += r2 * r2;
        }
    } else {
        const i64 x = u, y = (i64)blockIdx.y * 64 + 63;
        if (x < g.nx && y < g.ny - 1) {
            const i64 e = g.offBy + g.byLayer * tl + y + g.pyb * x;
            if (!NOMOM) {
                const double rm = (rho_node(y, x, tl) + rho_node(y + 1, x, tl)) / 2.0;
                const double rb = k.dsD * (rm * q[e]);
                const double m = k.kappa * (wgt(e) * alpha[e]);
                const double d = m - rb;
                sM += d * d;
                sR += rb * rb;
            }
            const double gb = c.sf * (q2[e] + sy[(tl * g.nx + x) * fg.nyblk + (y / 64 + 1)]);
            sF += gb * gb;
            const double r2 = gb + wgt(e) * alpha[e];
            sD += r2 * r2;
        }
    }
    }
    __shared__ double red[4][4];
    double v4[4] = {sM, sR, sF, sD};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double v = v4[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < S_COUNT) {
        const int s = threadIdx.x;
        const int i = (s == S_MRHOB) ? 0 : (s == S_RHOB2) ? 1 : (s == S_FBBETA2) ? 2 : (s == S_DUAL2) ? 3 : -1;
        double v = 0.0;
        if (i >= 0) v = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        const i64 bb = blockIdx.x + (i64)gridDim.x * (blockIdx.y + (i64)gridDim.y * blockIdx.z);
        partials[bb * S_COUNT + s] = v;
    }
}

// Fixed-order reduction of the per-workgroup partial sums, in two levels (the q-step's KKT variant alone leaves tens of
// thousands of rows): block (s, j) adds sum s over the j-th slice of rows, a second launch adds the KKT_SLICES slices.
__global__ void __launch_bounds__(256) k_kkt_final(const double *__restrict__ partials, i64 nblocks,
                                                    double *__restrict__ sums) {
    const int s = blockIdx.x;
    const i64 per = (nblocks + gridDim.y - 1) / gridDim.y;
    const i64 b0 = (i64)blockIdx.y * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
    double v = 0.0;
    for (i64 b = b0 + threadIdx.x; b < b1; b += 256) v += partials[b * S_COUNT + s];
    __shared__ double red[256];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[(i64)blockIdx.y * S_COUNT + s] = red[0];
}

__global__ void __launch_bounds__(64) k_kkt_final2(const double *__restrict__ slices, int nslices, double *__restrict__ sums) {
    const int s = threadIdx.x;
    if (s >= S_COUNT) return;
    double v = 0.0;
    for (int j = 0; j < nslices; ++j) v += slices[(i64)j * S_COUNT + s];
    sums[s] = v;
}

static void kkt_geometry(const Grid &g, dim3 &grid, i64 &chunk) {
    const i64 tiles = ((g.ny + TILE_Y - 1) / TILE_Y) * ((g.nx + TILE_X - 1) / TILE_X);
    i64 chunks = (4096 + tiles - 1) / tiles;
    if (chunks < 1) chunks = 1;
    if (chunks > g.ntl) chunks = g.ntl;
    chunk = (g.ntl + chunks - 1) / chunks;
    chunks = (g.ntl + chunk - 1) / chunk;
    grid = dim3((unsigned)((g.ny + TILE_Y - 1) / TILE_Y), (unsigned)((g.nx + TILE_X - 1) / TILE_X), (unsigned)chunks);
}

static i64 bnd_blocks(const Grid &g, const FusedGeom &fg, int dir) {
    const i64 tch = (g.ntl + BND_TC - 1) / BND_TC;
    if (dir == 0) return ((g.ny + 255) / 256) * fg.nxblk * tch;
    return ((g.nx + 255) / 256) * fg.nyblk * tch;
}

static i64 kkt_region_blocks(const Grid &g) {
    dim3 grid;
    i64 chunk;
    kkt_geometry(g, grid, chunk);
    FusedGeom fg;
    fused_geometry(g, fg);
    i64 m = (i64)grid.x * grid.y * grid.z;
    const i64 cand[] = {fg.nyblk * fg.nxblk * fg.chunks, qstep_rhs_blocks(g, fg), bnd_blocks(g, fg, 0), bnd_blocks(g, fg, 1)};
    for (i64 v : cand) m = v > m ? v : m;
    return m;
}

// KKT_REGIONS regions of per-workgroup partial sums: 0-3 node / cell / bx / by launches (folded path: q-step / cells / the two
// border launches), 4-6 the one-layer launches of the folded path on time slabs (launch_norms: one region per launch); a region is as long as the largest grid
// that writes to it, entries no launch writes stay zero (buffer cleared before every use)
#define KKT_REGIONS 8
i64 kkt_partials_needed(const Grid &g) { return KKT_REGIONS * kkt_region_blocks(g); }

// parts: bit mask of 1 node, 2 cell (+ q0 entries, needs a stored z), 4 bx edges, 8 by edges
int launch_kkt(const Grid &g, const LoopCoef &c, const KktCoef &k, const double *phi, const double *q,
               const double *alpha, const double *z, const double *beta, const double *cvec, const double *weight,
               const KktHalo &halo, const KktWork &w, int parts, hipStream_t st, bool layer0, double *resid) {
    dim3 grid;
    i64 chunk;
    kkt_geometry(g, grid, chunk);
    const i64 region = kkt_region_blocks(g);
    if (KKT_REGIONS * region > w.maxBlocks) {
        set_error("kkt workspace too small");
        return DOTSOCP_EINVAL;
    }
    // layer0 (folded path on a time slab): the first node / edge layer only, partial sums in regions 4-6
    const int base = layer0 ? 4 : 0;
    if (layer0) { chunk = 1; grid.z = 1; }
    const dim3 blk(TILE_Y, TILE_X);
#define KKT_PART(W, P, slot)                                                                                     \
    if (parts & P)                                                                                               \
    DS_KLAUNCH((k_kkt<W, P>), grid, blk, 0, st, g, c, k, chunk, phi, q, alpha, z, beta, cvec, weight, halo, \
                       w.partials + ((layer0 && (slot) > 0 ? (slot) - 1 : (slot)) + base) * region * S_COUNT,    \
                       (P == 1) ? resid : nullptr)
    if (weight) { KKT_PART(true, 1, 0); KKT_PART(true, 2, 1); KKT_PART(true, 4, 2); KKT_PART(true, 8, 3); }
    else { KKT_PART(false, 1, 0); KKT_PART(false, 2, 1); KKT_PART(false, 4, 2); KKT_PART(false, 8, 3); }
#undef KKT_PART
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_kkt_nodual(const Grid &g, const LoopCoef &c, const KktCoef &k, const double *phi, const double *q,
                      const double *alpha, const double *cvec, const double *weight, const KktWork &w, hipStream_t st) {
    dim3 grid;
    i64 chunk;
    kkt_geometry(g, grid, chunk);
    const i64 region = kkt_region_blocks(g);
    if (KKT_REGIONS * region > w.maxBlocks) {
        set_error("kkt workspace too small");
        return DOTSOCP_EINVAL;
    }
    const dim3 blk(TILE_Y, TILE_X);
    const KktHalo halo{nullptr, nullptr, nullptr, nullptr};
    const double *none = nullptr;
#define KKT_ND(W, P, reg)                                                                                                 \
    DS_KLAUNCH((k_kkt<W, P, true>), grid, blk, 0, st, g, c, k, chunk, phi, q, alpha, none, none, cvec, weight, halo, \
               w.partials + (i64)(reg) * region * S_COUNT, (double *)nullptr)
    if (weight) { KKT_ND(true, 1, 0); KKT_ND(true, 16, 4); KKT_ND(true, 4, 5); KKT_ND(true, 8, 6); }
    else { KKT_ND(false, 1, 0); KKT_ND(false, 16, 4); KKT_ND(false, 4, 5); KKT_ND(false, 8, 6); }
#undef KKT_ND
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_kkt_bnd_dual(const Grid &g, const LoopCoef &c, const KktCoef &k, const FusedGeom &fg, const double *q,
                        const double *alpha, const double *weight, const double *q2, const double *sx, const double *sy,
                        const KktWork &w, hipStream_t st) {
    const i64 region = kkt_region_blocks(g);
    for (int dir = 0; dir < 2; ++dir) {
        const i64 len = dir == 0 ? g.ny : g.nx, lines = dir == 0 ? fg.nxblk : fg.nyblk;
        if (len <= 0 || lines <= 0) continue;
        dim3 gb((unsigned)((len + 255) / 256), (unsigned)lines, (unsigned)((g.ntl + BND_TC - 1) / BND_TC));
        double *pb = w.partials + (2 + dir) * region * S_COUNT;
#define BND(W, D) DS_KLAUNCH((k_kkt_bnd<W, D, true>), gb, dim3(256), 0, st, g, c, k, fg, q, alpha, weight, q2, sx, sy, pb)
        if (weight) { if (dir == 0) BND(true, 0); else BND(true, 1); }
        else { if (dir == 0) BND(false, 0); else BND(false, 1); }
#undef BND
        DS_HIP(hipGetLastError());
    }
    return 0;
}

// fused path: pending multiplier step + cell sums in one pass (region 1); a.beta_out must differ from a.beta_in.
// edges: the iteration's q-step ran in its KKT variant (region 0): this launch adds the F*B*beta' terms of all edges inside
// the tiles and a second one (regions 2, 3) everything that is left on the tile borders; a.q2 / sx / sy are scratch then.
int launch_kkt_cells_update(const Grid &g, const LoopCoef &c, const KktCoef &k, const FusedGeom &fg, FusedArgs a,
                            const double *phi, const double *alpha, const double *weight, const KktWork &w,
                            hipStream_t st, bool edges, const double *q_new) {
    if (g.Nz <= 0) return 0;
    const i64 region = kkt_region_blocks(g);
    a.TC = fg.TC;
    dim3 grid((unsigned)fg.nyblk, (unsigned)fg.nxblk, (unsigned)fg.chunks);
    double *part = w.partials + 1 * region * S_COUNT;
    const dim3 blk(TILE_Y, TILE_X);
    if (!edges) {
        if (weight) DS_KLAUNCH((k_kkt_cells<true, false>), grid, blk, 0, st, g, c, k, a, phi, alpha, weight, part);
        else DS_KLAUNCH((k_kkt_cells<false, false>), grid, blk, 0, st, g, c, k, a, phi, alpha, weight, part);
        DS_HIP(hipGetLastError());
        return 0;
    }
    const bool nt = stream_nt_enabled();
    if (weight) {
        if (nt) DS_KLAUNCH((k_kkt_cells<true, true, false, true>), grid, blk, 0, st, g, c, k, a, phi, alpha, weight, part);
        else DS_KLAUNCH((k_kkt_cells<true, true>), grid, blk, 0, st, g, c, k, a, phi, alpha, weight, part);
    } else {
        if (nt) DS_KLAUNCH((k_kkt_cells<false, true, false, true>), grid, blk, 0, st, g, c, k, a, phi, alpha, weight, part);
        else DS_KLAUNCH((k_kkt_cells<false, true>), grid, blk, 0, st, g, c, k, a, phi, alpha, weight, part);
    }
    DS_HIP(hipGetLastError());
    for (int dir = 0; dir < 2; ++dir) {
        const i64 len = dir == 0 ? g.ny : g.nx, lines = dir == 0 ? fg.nxblk : fg.nyblk;
        if (len <= 0 || lines <= 0) continue;
        dim3 gb((unsigned)((len + 255) / 256), (unsigned)lines, (unsigned)((g.ntl + BND_TC - 1) / BND_TC));
        double *pb = w.partials + (2 + dir) * region * S_COUNT;
#define BND(W, D) DS_KLAUNCH((k_kkt_bnd<W, D>), gb, dim3(256), 0, st, g, c, k, fg, q_new, alpha, weight, a.q2, a.sx, a.sy, pb)
        if (weight) { if (dir == 0) BND(true, 0); else BND(true, 1); }
        else { if (dir == 0) BND(false, 0); else BND(false, 1); }
#undef BND
        DS_HIP(hipGetLastError());
    }
    return 0;
}

// ||phi||^2, ||q||^2, ||alpha||^2 (owned entries), ||z||^2, ||beta||^2 of the iterate with its multiplier step pending:
// the five norms of the rescale block without materialising anything (regions 0-4 of the partial sums)
int launch_norms(const Grid &g, const LoopCoef &c, const KktCoef &k, const FusedGeom &fg, FusedArgs a, const double *phi,
                 const double *alpha, const double *weight, const KktWork &w, hipStream_t st) {
    const i64 region = kkt_region_blocks(g);
    a.TC = fg.TC;
    if (g.Nz > 0) {
        dim3 grid((unsigned)fg.nyblk, (unsigned)fg.nxblk, (unsigned)fg.chunks);
        double *part = w.partials + 1 * region * S_COUNT;
        if (weight) DS_KLAUNCH((k_kkt_cells<true, false, true>), grid, dim3(TILE_Y, TILE_X), 0, st, g, c, k, a, phi, alpha, weight, part);
        else DS_KLAUNCH((k_kkt_cells<false, false, true>), grid, dim3(TILE_Y, TILE_X), 0, st, g, c, k, a, phi, alpha, weight, part);
    }
    // q-layout arrays: q0 cells, then the owned bx and by layers (a slab's halo layers are not its own)
    struct R { const double *x; i64 n; int slot, reg; };
    const R rs[] = {{phi, g.Nphi, S_PHI2, 0},
                    {a.q, g.Nz, S_Q2, 2}, {a.q + g.offBx, g.bxLayer * g.ntl, S_Q2, 3}, {a.q + g.offBy, g.byLayer * g.ntl, S_Q2, 4},
                    {alpha, g.Nz, S_ALPHA2, 5}, {alpha + g.offBx, g.bxLayer * g.ntl, S_ALPHA2, 6},
                    {alpha + g.offBy, g.byLayer * g.ntl, S_ALPHA2, 7}};
    for (const R &r : rs) {
        if (r.n <= 0) continue;
        i64 blocks = (r.n + 256 * 64 - 1) / (256 * 64);
        if (blocks > region) blocks = region;
        double *part = w.partials + r.reg * region * S_COUNT;
        DS_KLAUNCH(k_sumsq, dim3((unsigned)blocks), dim3(256), 0, st, r.x, r.n, r.slot, part);
    }
    DS_HIP(hipGetLastError());
    return 0;
}

double *kkt_qstep_partials(const Grid &g, const KktWork &w) {
    (void)g;
    return w.partials;       // region 0 (the node launch of the unfolded path uses it otherwise)
}

int launch_kkt_final(const Grid &g, const KktWork &w, hipStream_t st) {
    // w.sums: [S_COUNT] result followed by [KKT_SLICES][S_COUNT] intermediate sums
    DS_KLAUNCH(k_kkt_final, dim3(S_COUNT, KKT_SLICES), dim3(256), 0, st, w.partials,
                       KKT_REGIONS * kkt_region_blocks(g), w.sums + S_COUNT);
    DS_KLAUNCH(k_kkt_final2, dim3(1), dim3(64), 0, st, w.sums + S_COUNT, KKT_SLICES, w.sums);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
