// KKT block of the loop (solver_socp_inPALM.m:220-267, compute_kkt_dot_complement.m:2-18;
// weighted: solver_wsocp_inPALM.m:232-272, wdot2d/utils/compute_kkt_dot_complement.m:2-19):
// every global sum the residuals need, produced by ONE pass over the state.
// Each thread owns a (y, x) column of a time chunk and visits, per layer, its node, its cell and
// its two staggered edges; 20 running sums stay in registers, are reduced over the wavefront
// with cross-lane shuffles, over the workgroup through LDS, and written as per-block partials;
// a second kernel adds the partials in a fixed order (deterministic, no atomics).
#include "device_utils.h"
#include "kernels.h"

namespace dotsocp {

struct WBeta {
    const double *b;
    i64 Nz;
    __device__ __forceinline__ double operator()(int j, i64 cell) const { return b[j * Nz + cell]; }
};

// PART selects which entries a launch visits (1 node, 2 cell + its q0 entry, 4 bx edge, 8 by edge):
// four lean launches keep the register count low and the occupancy high.
template <bool WEIGHTED, int PART>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_kkt(Grid g, LoopCoef c, KktCoef k, i64 chunk,
                                                         const double *__restrict__ phi, const double *__restrict__ q,
                                                         const double *__restrict__ alpha, const double *__restrict__ z,
                                                         const double *__restrict__ beta,
                                                         const double *__restrict__ cvec,
                                                         const double *__restrict__ weight,
                                                         KktHalo halo, double *__restrict__ partials) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const bool inb = (y < g.ny) && (x < g.nx);
    const i64 tbeg = (i64)blockIdx.z * chunk;
    const i64 tend = (tbeg + chunk < g.ntl) ? tbeg + chunk : g.ntl;
    double S[S_COUNT];
#pragma unroll
    for (int i = 0; i < S_COUNT; ++i) S[i] = 0.0;
    WBeta WB{beta, g.Nz};
    auto wgt = [&](i64 idx) { return WEIGHTED ? weight[idx] : 1.0; };
    // rhoT of a cell = kappa * (w .* alpha)_0 ; 0 outside the time range (zero padding of movmean)
    auto rhoT_at = [&](i64 yy, i64 xx, i64 tl) -> double {
        if (tl >= g.ncl) return 0.0;
        if (tl < 0) {
            if (g.first) return 0.0;
            return k.kappa * halo.a0w_prev[yy + g.ny * xx];
        }
        const i64 cidx = yy + g.ny * (xx + g.nx * tl);
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
        S[S_FBBETA2] += q2b * q2b;
        const double r2 = q2b + wa;
        S[S_DUAL2] += r2 * r2;
        S[S_QALPHA] += wq * av;
    };
    if (inb) {
        for (i64 tl = tbeg; tl < tend; ++tl) {
            const i64 node = y + g.ny * (x + g.nx * tl);
            const i64 bxo = g.offBx + g.bxLayer * tl, byo = g.offBy + g.byLayer * tl;
            // ---- node: A' alpha - c, <c, phi>, ||phi||^2 ----
            if (PART & 1) {
                double r = 0.0;
                if (tl >= 1)
                    r += c.at * alpha[node - g.plane];
                else if (!g.first)
                    r += c.at * halo.a0_prev[y + g.ny * x];
                if (tl < g.ncl) r += (-c.at) * alpha[node];
                if (x >= 1) r += c.ax * alpha[bxo + y + g.ny * (x - 1)];
                if (x <= g.nx - 2) r += (-c.ax) * alpha[bxo + y + g.ny * x];
                if (y >= 1) r += c.ay * alpha[byo + (y - 1) + (g.ny - 1) * x];
                if (y <= g.ny - 2) r += (-c.ay) * alpha[byo + y + (g.ny - 1) * x];
                const double cv = cvec[node], pv = phi[node];
                r = r - cv;
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
                    const double zv = z[j * g.Nz + node], bv = beta[j * g.Nz + node];
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
            // ---- bx edge (y, x+1/2, t) ----
            if ((PART & 4) && x < g.nx - 1) {
                const i64 idx = bxo + y + g.ny * x;
                double tmp = (-c.ax) * phi[node];
                tmp += c.ax * phi[node + g.ny];
                const double q2b = c.sf * gather_bx(g, WB, y, x, tl, halo.btail_bx);
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
                const i64 idx = byo + y + (g.ny - 1) * x;
                double tmp = (-c.ay) * phi[node];
                tmp += c.ay * phi[node + 1];
                const double q2b = c.sf * gather_by(g, WB, y, x, tl, halo.btail_by);
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
template <bool WEIGHTED>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_kkt_cells(Grid g, LoopCoef c, KktCoef k, FusedArgs a,
                                                               const double *__restrict__ phi,
                                                               const double *__restrict__ alpha,
                                                               const double *__restrict__ weight,
                                                               double *__restrict__ partials) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const bool inb = (y < g.ny) && (x < g.nx);
    const i64 t0 = (i64)blockIdx.z * a.TC;
    const i64 t1 = (t0 + a.TC < g.ncl) ? t0 + a.TC : g.ncl;
    double S[S_COUNT];
#pragma unroll
    for (int i = 0; i < S_COUNT; ++i) S[i] = 0.0;
    if (inb) {
        EdgeQuad cur = load_edges(g, a.q, y, x, t0, c.sf);
        EdgeQuad curo = load_edges(g, a.q_old, y, x, t0, c.sf);
        for (i64 tl = t0; tl < t1; ++tl) {
            const i64 i = y + g.ny * (x + g.nx * tl);
            const EdgeQuad nxt = load_edges(g, a.q, y, x, tl + 1, c.sf);
            const EdgeQuad nxto = load_edges(g, a.q_old, y, x, tl + 1, c.sf);
            const double q0 = a.q[i];
            double b[10], v[10], zo[10], p[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) b[j] = a.beta_in[j * g.Nz + i];
            if (a.bpend) {
#pragma unroll
                for (int j = 0; j < 10; ++j) b[j] = b[j] * a.bmul / a.bdiv;
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
                a.beta_out[j * g.Nz + i] = b[j];
                p[j] = zo[j] - k.sigma * b[j];
                zs += zo[j] * zo[j];
                bs += b[j] * b[j];
                rs += r * r;
            }
            proj_row<10>(p);
            double cs = 0.0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const double d = zo[j] - p[j];
                cs += d * d;
            }
            S[S_Z2] += zs;
            S[S_BETA2] += bs;
            S[S_PRIM2] += rs;
            S[S_COMPLEM] += cs;
            const double w = WEIGHTED ? weight[i] : 1.0;
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
            double tmp = (-c.at) * phi[i];
            tmp += c.at * phi[i + g.plane];
            const double q2b = c.s * (b[9] - b[0]);
            const double wq = w * q0, wa = w * av;
            S[S_Q2] += q0 * q0;
            S[S_ALPHA2] += av * av;
            S[S_APHI2] += tmp * tmp;
            const double r1 = tmp - wq;
            S[S_PRIM1] += r1 * r1;
            S[S_FBBETA2] += q2b * q2b;
            const double r2 = q2b + wa;
            S[S_DUAL2] += r2 * r2;
            S[S_QALPHA] += wq * av;
            cur = nxt;
            curo = nxto;
        }
    }
    __shared__ double red[TILE_X][S_COUNT];
    const int lane = threadIdx.x;
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

__global__ void __launch_bounds__(256) k_kkt_final(const double *__restrict__ partials, i64 nblocks,
                                                    double *__restrict__ sums) {
    const int s = blockIdx.x;
    double v = 0.0;
    for (i64 b = threadIdx.x; b < nblocks; b += 256) v += partials[b * S_COUNT + s];
    __shared__ double red[256];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[s] = red[0];
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

static i64 kkt_region_blocks(const Grid &g) {
    dim3 grid;
    i64 chunk;
    kkt_geometry(g, grid, chunk);
    FusedGeom fg;
    fused_geometry(g, fg);
    const i64 a = (i64)grid.x * grid.y * grid.z, b = fg.nyblk * fg.nxblk * fg.chunks;
    return a > b ? a : b;
}

// four regions (node, cell, bx, by launches) of per-workgroup partial sums; a region is as long as
// the largest grid that writes to it, entries no launch writes stay zero (buffer cleared once)
i64 kkt_partials_needed(const Grid &g) { return 4 * kkt_region_blocks(g); }

// parts: bit mask of 1 node, 2 cell (+ q0 entries, needs a stored z), 4 bx edges, 8 by edges
int launch_kkt(const Grid &g, const LoopCoef &c, const KktCoef &k, const double *phi, const double *q,
               const double *alpha, const double *z, const double *beta, const double *cvec, const double *weight,
               const KktHalo &halo, const KktWork &w, int parts, hipStream_t st) {
    dim3 grid;
    i64 chunk;
    kkt_geometry(g, grid, chunk);
    const i64 region = kkt_region_blocks(g);
    if (4 * region > w.maxBlocks) {
        set_error("kkt workspace too small");
        return DOTSOCP_EINVAL;
    }
    const dim3 blk(TILE_Y, TILE_X);
#define KKT_PART(W, P, slot)                                                                                     \
    if (parts & P)                                                                                               \
    hipLaunchKernelGGL((k_kkt<W, P>), grid, blk, 0, st, g, c, k, chunk, phi, q, alpha, z, beta, cvec, weight, halo, \
                       w.partials + (slot) * region * S_COUNT)
    if (weight) { KKT_PART(true, 1, 0); KKT_PART(true, 2, 1); KKT_PART(true, 4, 2); KKT_PART(true, 8, 3); }
    else { KKT_PART(false, 1, 0); KKT_PART(false, 2, 1); KKT_PART(false, 4, 2); KKT_PART(false, 8, 3); }
#undef KKT_PART
    DS_HIP(hipGetLastError());
    return 0;
}

// fused path: pending multiplier step + cell sums in one pass (region 1); a.beta_out must differ from a.beta_in
int launch_kkt_cells_update(const Grid &g, const LoopCoef &c, const KktCoef &k, const FusedGeom &fg, FusedArgs a,
                            const double *phi, const double *alpha, const double *weight, const KktWork &w,
                            hipStream_t st) {
    if (g.Nz <= 0) return 0;
    const i64 region = kkt_region_blocks(g);
    a.TC = fg.TC;
    dim3 grid((unsigned)fg.nyblk, (unsigned)fg.nxblk, (unsigned)fg.chunks);
    double *part = w.partials + 1 * region * S_COUNT;
    if (weight)
        hipLaunchKernelGGL(k_kkt_cells<true>, grid, dim3(TILE_Y, TILE_X), 0, st, g, c, k, a, phi, alpha, weight, part);
    else
        hipLaunchKernelGGL(k_kkt_cells<false>, grid, dim3(TILE_Y, TILE_X), 0, st, g, c, k, a, phi, alpha, weight, part);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_kkt_final(const Grid &g, const KktWork &w, hipStream_t st) {
    hipLaunchKernelGGL(k_kkt_final, dim3(S_COUNT), dim3(256), 0, st, w.partials, 4 * kkt_region_blocks(g), w.sums);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
