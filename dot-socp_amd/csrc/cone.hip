// Cone kernels: SOC projection, B F q + d, its adjoint, and the fused per-cell kernels of the
// inPALM loop.  fp64, SoA planes (plane j of cell i at j*Nz + i), y-fastest coalescing:
// a wavefront covers 64 consecutive y of one (x, t) column, so every plane access is one
// 512-byte coalesced segment.  Bandwidth-bound: no MFMA, no LDS needed for correctness --
// x/y neighbours of the staggered q are served by L1/L2 (they are 1/10 of the traffic).
//
// Reference semantics: SURVEY.md section 8a rows a1-a3, a6; call sites
// socp/dot2d/algorithms/solver_socp_inPALM.m:133,187,199,205,212-215,225,240,242.
#include "device_utils.h"
#include "kernels.h"

namespace dotsocp {

// --------------------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------------------

// --------------------------------------------------------------------------------------
// mexProjSoc: generic M x K
// --------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) k_proj_soc(double *__restrict__ out, const double *__restrict__ in, i64 M) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (i64)gridDim.x * blockDim.x) {
        double v[K];
#pragma unroll
        for (int j = 0; j < K; ++j) v[j] = in[j * M + i];
        proj_row<K>(v);
#pragma unroll
        for (int j = 0; j < K; ++j) out[j * M + i] = v[j];
    }
}

// any K >= 2: three passes over the row like the original (norm, coefficient, scale)
__global__ void __launch_bounds__(256) k_proj_soc_any(double *__restrict__ out, const double *__restrict__ in, i64 M,
                                                      i64 K) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (i64)gridDim.x * blockDim.x) {
        double x1 = in[i];
        double nn = 0.0;
        for (i64 j = 1; j < K; ++j) {
            double t = in[j * M + i];
            nn += t * t;
        }
        const double n = sqrt(nn);
        double c = (x1 / n + 1.0) * 0.5;
        c = (c > 1.0) ? 1.0 : c;
        c = (c < 0.0) ? 0.0 : c;
        for (i64 j = 1; j < K; ++j) out[j * M + i] = c * in[j * M + i];
        out[i] = (c >= 1.0) ? x1 : c * n;
    }
}

int launch_proj_soc(double *out, const double *in, i64 M, i64 K, hipStream_t st) {
    if (M <= 0) return 0;
    int blocks = launch_blocks(M, 256, 1 << 16);
    if (K == 10)
        DS_KLAUNCH(k_proj_soc<10>, dim3(blocks), dim3(256), 0, st, out, in, M);
    else if (K == 6)
        DS_KLAUNCH(k_proj_soc<6>, dim3(blocks), dim3(256), 0, st, out, in, M);
    else
        DS_KLAUNCH(k_proj_soc_any, dim3(blocks), dim3(256), 0, st, out, in, M, K);
    DS_HIP(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------
// mexBFd / mexBFdConj (operator level; materialise the Nz x 10 matrix)
// --------------------------------------------------------------------------------------
static inline dim3 cell_grid(const Grid &g, i64 layers) {
    return dim3((unsigned)((g.ny + TILE_Y - 1) / TILE_Y), (unsigned)((g.nx + TILE_X - 1) / TILE_X), (unsigned)layers);
}

__global__ void __launch_bounds__(TILE_Y *TILE_X) k_bfd(Grid g, double *__restrict__ z, const double *__restrict__ q,
                                                         double s, double sf, double dF) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = blockIdx.z;
    if (y >= g.ny || x >= g.nx) return;
    const i64 i = y + g.py * (x + g.nx * tl);
    const double q0 = q[i];
    z[i] = dF - s * q0;
    z[9 * g.Nc + i] = dF + s * q0;
    const double *bx = q + g.offBx;
    const double *by = q + g.offBy;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const i64 tt = tl + dt;
        if (x >= 1) z[(1 + 2 * dt) * g.Nc + i] = sf * bx[g.bxLayer * tt + y + g.py * (x - 1)];
        if (x <= g.nx - 2) z[(2 + 2 * dt) * g.Nc + i] = sf * bx[g.bxLayer * tt + y + g.py * x];
        if (y >= 1) z[(5 + 2 * dt) * g.Nc + i] = sf * by[g.byLayer * tt + (y - 1) + g.pyb * x];
        if (y <= g.ny - 2) z[(6 + 2 * dt) * g.Nc + i] = sf * by[g.byLayer * tt + y + g.pyb * x];
    }
}

int launch_bfd(const Grid &g, double *z, const double *q, double s, double dF, hipStream_t st) {
    if (g.Nz <= 0) return 0;
    DS_KLAUNCH(k_bfd, cell_grid(g, g.ncl), dim3(TILE_Y, TILE_X), 0, st, g, z, q, s, s / sqrt(2.0), dF);
    DS_HIP(hipGetLastError());
    return 0;
}

// seg: 0 = q0 (cells), 1 = bx edges, 2 = by edges
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_bfd_conj(Grid g, double *__restrict__ q, const double *__restrict__ w,
                                                              double s, double sf, int seg,
                                                              const double *__restrict__ tail_bx,
                                                              const double *__restrict__ tail_by) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = blockIdx.z;
    WPlain W{w, g.Nc};
    if (seg == 0) {
        if (y >= g.ny || x >= g.nx) return;
        const i64 i = y + g.py * (x + g.nx * tl);
        q[i] = s * (w[9 * g.Nc + i] - w[i]);
    } else if (seg == 1) {
        if (y >= g.ny || x >= g.nx - 1) return;
        q[g.offBx + g.bxLayer * tl + y + g.py * x] = sf * gather_bx(g, W, y, x, tl, tail_bx);
    } else {
        if (y >= g.ny - 1 || x >= g.nx) return;
        q[g.offBy + g.byLayer * tl + y + g.pyb * x] = sf * gather_by(g, W, y, x, tl, tail_by);
    }
}

int launch_bfd_conj(const Grid &g, double *q, const double *w, double s, hipStream_t st, const double *tail_bx,
                    const double *tail_by) {
    const double sf = s / sqrt(2.0);
    if (!g.first && (!tail_bx || !tail_by)) {
        set_error("bfd_conj on a time slab needs the left neighbour's tails");
        return DOTSOCP_EINVAL;
    }
    const dim3 blk(TILE_Y, TILE_X);
    if (g.Nz > 0) DS_KLAUNCH(k_bfd_conj, cell_grid(g, g.ncl), blk, 0, st, g, q, w, s, sf, 0, tail_bx, tail_by);
    if (g.bxLayer > 0) DS_KLAUNCH(k_bfd_conj, cell_grid(g, g.ntl), blk, 0, st, g, q, w, s, sf, 1, tail_bx, tail_by);
    if (g.byLayer > 0) DS_KLAUNCH(k_bfd_conj, cell_grid(g, g.ntl), blk, 0, st, g, q, w, s, sf, 2, tail_bx, tail_by);
    DS_HIP(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------
// Fused per-cell kernels of the loop: t-marching (each thread owns one (y, x) column and
// walks a chunk of time cells, carrying the t+1 edge layer in registers so every q entry is
// fetched once per chunk).
// --------------------------------------------------------------------------------------
#define MARCH 8   // time cells per block

// MODE 0: z = Pi_Q(B F q + d - beta)                       (solver_socp_inPALM.m:199)
// MODE 1: beta += tau * (z - (B F q + d))                  (solver_socp_inPALM.m:212-215)
template <int MODE>
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_cone_march(Grid g, LoopCoef c, const double *__restrict__ q,
                                                                const double *zin, const double *betain,
                                                                double *zout, double *betaout) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    if (y >= g.ny || x >= g.nx) return;
    const i64 tbeg = (i64)blockIdx.z * MARCH;
    const i64 tend = (tbeg + MARCH < g.ncl) ? tbeg + MARCH : g.ncl;
    EdgeQuad cur = load_edges(g, q, y, x, tbeg, c.sf);
    for (i64 tl = tbeg; tl < tend; ++tl) {
        const i64 i = y + g.py * (x + g.nx * tl);
        const EdgeQuad nxt = load_edges(g, q, y, x, tl + 1, c.sf);
        double v[10];
        build_z2(v, q[i], cur, nxt, c.s, c.dF);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = v[j] - betain[j * g.Nc + i];
            proj_row<10>(v);
#pragma unroll
            for (int j = 0; j < 10; ++j) zout[j * g.Nc + i] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const double r = zin[j * g.Nc + i] - v[j];
                betaout[j * g.Nc + i] = betain[j * g.Nc + i] + c.tau * r;
            }
        }
        cur = nxt;
    }
}

static inline dim3 march_grid(const Grid &g) {
    return dim3((unsigned)((g.ny + TILE_Y - 1) / TILE_Y), (unsigned)((g.nx + TILE_X - 1) / TILE_X),
                (unsigned)((g.ncl + MARCH - 1) / MARCH));
}

int launch_cone_proj(const Grid &g, const LoopCoef &c, const double *q, const double *beta, double *z,
                     hipStream_t st) {
    if (g.Nz <= 0) return 0;
    DS_KLAUNCH(k_cone_march<0>, march_grid(g), dim3(TILE_Y, TILE_X), 0, st, g, c, q, (const double *)nullptr,
                       beta, z, (double *)nullptr);
    DS_HIP(hipGetLastError());
    return 0;
}

int launch_beta_update(const Grid &g, const LoopCoef &c, const double *q, const double *z, double *beta,
                       hipStream_t st) {
    if (g.Nz <= 0) return 0;
    DS_KLAUNCH(k_cone_march<1>, march_grid(g), dim3(TILE_Y, TILE_X), 0, st, g, c, q, z, (const double *)beta,
                       (double *)nullptr, beta);
    DS_HIP(hipGetLastError());
    return 0;
}

// Partial adjoint sums of the last owned cell layer: what the first bx / by layer of the right
// neighbour slab is missing (columns 4,5 resp. 8,9 of cell t-1/2).
__global__ void __launch_bounds__(TILE_Y *TILE_X) k_gather_tail(Grid g, const double *__restrict__ z,
                                                                 const double *__restrict__ beta,
                                                                 double *__restrict__ tail_bx,
                                                                 double *__restrict__ tail_by) {
    const i64 y = (i64)blockIdx.x * TILE_Y + threadIdx.x;
    const i64 x = (i64)blockIdx.y * TILE_X + threadIdx.y;
    const i64 tl = g.ncl - 1;
    auto w = [&](int j, i64 cell) { return z[j * g.Nc + cell] + beta[j * g.Nc + cell]; };
    if (y < g.ny && x < g.nx - 1) {
        double acc = w(3, y + g.py * ((x + 1) + g.nx * tl));
        acc += w(4, y + g.py * (x + g.nx * tl));
        tail_bx[y + g.py * x] = acc;
    }
    if (y < g.ny - 1 && x < g.nx) {
        double acc = w(7, (y + 1) + g.py * (x + g.nx * tl));
        acc += w(8, y + g.py * (x + g.nx * tl));
        tail_by[y + g.pyb * x] = acc;
    }
}

int launch_gather_tail(const Grid &g, const double *z, const double *beta, double *tail_bx, double *tail_by,
                       hipStream_t st) {
    if (g.ncl <= 0) return 0;
    DS_KLAUNCH(k_gather_tail, cell_grid(g, 1), dim3(TILE_Y, TILE_X), 0, st, g, z, beta, tail_bx, tail_by);
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
