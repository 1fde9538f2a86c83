// Device-side helpers shared by cone.hip, stencil.hip and kkt.hip.
#pragma once
#include "common.h"

namespace dotsocp {

#define TILE_Y 64
#define TILE_X 4

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in the order of their linear index.
// Where neighbouring tiles share cache lines -- the strided DCT axes (a tile row is 64 bytes wide for n = 1024)
// and every tile kernel on grids whose row length is not a multiple of 16 doubles (the 2^k+1 grids) -- the tile
// index is permuted such that neighbouring tiles run on the SAME XCD back to back: the second one hits in L2
// instead of fetching the shared line from HBM again.  Bijection on [0, nb).
__device__ __forceinline__ i64 xcd_tile(unsigned b, unsigned nb) {
    const unsigned full = nb >> 3, rem = nb & 7u, r = b & 7u, q = b >> 3;
    return (i64)r * full + (r < rem ? r : rem) + q;
}

// (x, y, z) block coordinates after that permutation of the linear block index (x fastest)
struct BlockId {
    unsigned x, y, z;
};
__device__ __forceinline__ BlockId block_id(bool remap) {
    BlockId b{blockIdx.x, blockIdx.y, blockIdx.z};
    if (remap) {
        const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
        const i64 L = xcd_tile(b.x + gridDim.x * (b.y + gridDim.y * b.z), nb);
        b.x = (unsigned)(L % gridDim.x);
        b.y = (unsigned)((L / gridDim.x) % gridDim.y);
        b.z = (unsigned)(L / ((i64)gridDim.x * gridDim.y));
    }
    return b;
}

// Workgroup barrier that leaves the vector-memory counter alone: LDS hand-off only.  The pipelined kernels keep LDS-DMA
// loads and global stores in flight across barriers (counted s_waitcnt vmcnt(N) of their own), which a fence would drain.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one wave-instruction: 64 lanes x 16 B from each lane's global address to LDS [lds_dst + 16 * lane] (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Streams that are read or written exactly once per pass and are far larger than the caches (the cone multipliers: 10 Nz
// doubles in, 10 Nz out): non-temporal accesses keep them from displacing the lines neighbouring tiles share (q edges,
// tile-border partial sums).  Cone kernel at 1024 x 1024 x 128: 5.87 -> 5.50 ms.  (The q-step's alpha / q2 / q streams:
// no effect, measured; its loads of neighbour tiles' entries rely on the L2.  The pipelined DCT kernels with `nt` on
// their LDS-DMA loads and stores: 2.50 -> 2.60 ms at 1024-point lines, 3.1 -> 5.3 ms at 2048 -- tiles narrower than a
// 128-byte line share every line with a neighbour workgroup.)
template <bool NT>
__device__ __forceinline__ double ld_stream(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st_stream(double *p, double v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
bool stream_nt_enabled();          // fused.hip: DOTSOCP_NT=0 switches the non-temporal flavours off

// Row projection onto {x1 >= ||x_2..K||}; literal restatement of mexProjSoc's arithmetic
// (SURVEY.md 8a a1): n = ||x_2..K||, c = clamp((x1/n + 1)/2, 0, 1) with NaN passing through,
// x_j <- c x_j, x_1 <- (c >= 1) ? x_1 : c n.
template <int K>
__device__ __forceinline__ void proj_row(double (&v)[K]) {
    double nn = v[1] * v[1];
#pragma unroll
    for (int j = 2; j < K; ++j) nn += v[j] * v[j];
    const double n = sqrt(nn);
    double c = (v[0] / n + 1.0) * 0.5;
    c = (c > 1.0) ? 1.0 : c;
    c = (c < 0.0) ? 0.0 : c;
#pragma unroll
    for (int j = 1; j < K; ++j) v[j] = c * v[j];
    v[0] = (c >= 1.0) ? v[0] : c * n;
}

// The four staggered edge values around cell column (y, x) at time layer tt (pre-multiplied
// by sf); edges outside the domain contribute exact zeros (mexBFd never writes those slots).
struct EdgeQuad {
    double xm, xp, ym, yp;
};

__device__ __forceinline__ EdgeQuad load_edges(const Grid &g, const double *__restrict__ q, i64 y, i64 x,
                                               i64 tt, double sf) {
    // All four loads are issued unconditionally from clamped (always valid) addresses and the out-of-domain ones are
    // replaced by zeros afterwards: straight-line code, the loads leave together and are waited for once.  (A load
    // under its own `if` is followed by its own wait -- four serialised memory latencies per step of a t-marching
    // kernel.)  Grids without bx (nx = 1: the 1-D problems) or by entries read the q0 block instead.
    const double *bx = (g.nx >= 2) ? q + g.offBx + g.bxLayer * tt : q;
    const double *by = (g.ny >= 2) ? q + g.offBy + g.byLayer * tt : q;
    const i64 nxe = (g.nx >= 2) ? g.nx - 2 : 0, nye = (g.ny >= 2) ? g.ny - 2 : 0;   // last edge index
    const i64 rowy = (g.ny >= 2) ? g.pyb : 0;                                        // by row stride
    const i64 xm = (x >= 1) ? x - 1 : 0, xp = (x <= nxe) ? x : nxe;
    const i64 ym = (y >= 1) ? y - 1 : 0, yp = (y <= nye) ? y : nye;
    const double vxm = bx[y + g.py * xm], vxp = bx[y + g.py * xp];
    const double vym = by[ym + rowy * x], vyp = by[yp + rowy * x];
    EdgeQuad e;
    e.xm = (x >= 1) ? sf * vxm : 0.0;
    e.xp = (x <= g.nx - 2) ? sf * vxp : 0.0;
    e.ym = (y >= 1) ? sf * vym : 0.0;
    e.yp = (y <= g.ny - 2) ? sf * vyp : 0.0;
    return e;
}

__device__ __forceinline__ void build_z2(double (&v)[10], double q0, const EdgeQuad &a, const EdgeQuad &b,
                                         double s, double dF) {
    v[0] = dF - s * q0;
    v[1] = a.xm; v[2] = a.xp; v[3] = b.xm; v[4] = b.xp;
    v[5] = a.ym; v[6] = a.yp; v[7] = b.ym; v[8] = b.yp;
    v[9] = dF + s * q0;
}

// Adjoint gather for one staggered edge.  `w(j, cell)` is supplied by a functor so the same
// code serves w = z (operator level), w = z + beta (q-step) and w = beta (KKT).
template <class W>
__device__ __forceinline__ double gather_bx(const Grid &g, const W &w, i64 y, i64 xe, i64 tl,
                                            const double *__restrict__ tail_bx) {
    double acc = 0.0;
    if (tl < g.ncl) {
        acc += w(1, y + g.py * ((xe + 1) + g.nx * tl));
        acc += w(2, y + g.py * (xe + g.nx * tl));
    }
    if (tl >= 1) {
        acc += w(3, y + g.py * ((xe + 1) + g.nx * (tl - 1)));
        acc += w(4, y + g.py * (xe + g.nx * (tl - 1)));
    } else if (!g.first) {
        acc += tail_bx[y + g.py * xe];
    }
    return acc;
}

template <class W>
__device__ __forceinline__ double gather_by(const Grid &g, const W &w, i64 ye, i64 x, i64 tl,
                                            const double *__restrict__ tail_by) {
    double acc = 0.0;
    if (tl < g.ncl) {
        acc += w(5, (ye + 1) + g.py * (x + g.nx * tl));
        acc += w(6, ye + g.py * (x + g.nx * tl));
    }
    if (tl >= 1) {
        acc += w(7, (ye + 1) + g.py * (x + g.nx * (tl - 1)));
        acc += w(8, ye + g.py * (x + g.nx * (tl - 1)));
    } else if (!g.first) {
        acc += tail_by[ye + g.pyb * x];
    }
    return acc;
}

struct WPlain {
    const double *w;
    i64 Nz;
    __device__ __forceinline__ double operator()(int j, i64 cell) const { return w[j * Nz + cell]; }
};

// rhs = A'(w.*q - alpha) + c at node (y, x, tl): sum over the (up to) six staggered neighbours in the
// column order of the reference's sparse product, Neumann: missing ones dropped
// (solver_socp_inPALM.m:194; weighted: solver_wsocp_inPALM.m:200).  Shared by k_rhs and by the
// first DCT pass of the Poisson solve, which forms rhs on the fly.
struct RhsArgs {
    Grid g;
    double at, ax, ay;
    const double *q, *alpha, *cvec, *weight, *u0_prev;
};

template <bool WEIGHTED>
__device__ __forceinline__ double rhs_value(const RhsArgs &a, i64 y, i64 x, i64 tl) {
    const Grid &g = a.g;
    auto u = [&](i64 k) { return WEIGHTED ? a.weight[k] * a.q[k] - a.alpha[k] : a.q[k] - a.alpha[k]; };
    const i64 node = y + g.py * (x + g.nx * tl);
    double r = 0.0;
    if (tl >= 1)
        r += a.at * u(node - g.plane);
    else if (!g.first)
        r += a.at * a.u0_prev[y + g.py * x];
    if (tl < g.ncl) r += (-a.at) * u(node);
    const i64 bxo = g.offBx + g.bxLayer * tl;
    if (x >= 1) r += a.ax * u(bxo + y + g.py * (x - 1));
    if (x <= g.nx - 2) r += (-a.ax) * u(bxo + y + g.py * x);
    const i64 byo = g.offBy + g.byLayer * tl;
    if (y >= 1) r += a.ay * u(byo + (y - 1) + g.pyb * x);
    if (y <= g.ny - 2) r += (-a.ay) * u(byo + y + g.pyb * x);
    return r + a.cvec[node];
}

}  // namespace dotsocp
