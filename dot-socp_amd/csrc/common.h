// Shared host/device helpers for libdotsocp (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/dotsocp.h"

typedef long long i64;

namespace dotsocp {

void set_error(const char *fmt, ...);

#define DS_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            dotsocp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,                \
                               hipGetErrorString(e__));                                     \
            return DOTSOCP_EHIP;                                                            \
        }                                                                                   \
    } while (0)

#define DS_CHECK(expr)                                                                      \
    do {                                                                                    \
        int r__ = (expr);                                                                   \
        if (r__ != 0) return r__;                                                           \
    } while (0)

#define DS_ARG(cond, msg)                                                                   \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            dotsocp::set_error("invalid argument: %s", msg);                                \
            return DOTSOCP_EINVAL;                                                          \
        }                                                                                   \
    } while (0)

// Staggered grid geometry of one time slab.  Index order everywhere: y fastest, then x,
// then t (MATLAB column-major, socp/dot2d/utils/initialize.m:18-25).
//
// A slab owns time nodes [t0, t0+ntl) (phi, c, bx, by) and the staggered cells
// [t0, t0+ncl) (q0, alpha0, z, beta); ncl = ntl except on the last slab (ntl-1).
// Every slab but the last keeps ONE extra halo layer of bx / by (and of phi) behind its
// own layers -- the first layer of the right neighbour -- so that B F q + d of its last
// cell and the forward time difference of phi can be formed locally (`halo` = 1).
// For the single slab (t0 = 0, ntl = nt, halo = 0) the local layout of q is exactly the
// reference's q = [q0; bx; by].
//
// Rows may be PITCHED: every array indexed by (y, x[, t]) -- phi, c, q0, bx, by, the ten cone planes -- stores its rows
// `py` doubles apart (py >= ny, a multiple of 16 = 128 bytes), by rows `pyb` apart (unpitched: ny - 1, the reference's
// layout; pitched: py).  A wavefront's 64 consecutive y then start on a cache-line boundary on the 2^k+1 grids the
// reference's multilevel driver runs (rows of 1025 doubles straddle five lines instead of four and both ends of every
// store are partial lines).  Only strides change: extents, loops and sums run over y < ny; the pad entries are zero
// (weights: one) and are never written.  Host arrays keep the reference layout (converted in upload / download).
struct Grid {
    i64 ny, nx, nt;        // global dims
    i64 py, pyb;           // row pitch of node / q0 / bx rows, of by rows
    i64 t0, ntl, ncl;      // slab: first node, #nodes, #cells
    int halo;              // 1: an extra bx/by/phi layer is stored behind the owned ones
    int first, last;       // slab touches the global t = 0 / t = nt-1 boundary
    i64 plane;             // py*nx
    i64 bxLayer, byLayer;  // py*(nx-1), pyb*nx
    i64 Nphi, Nz;          // owned nodes / cells
    i64 Nc;                // doubles between the ten columns of a cone array (z, beta): Nz, or Nz + a pad that keeps the columns
                           // off each other's DRAM banks (Nz * 8 bytes is a multiple of every power of two in sight)
    i64 offBx, offBy, NqAlloc;  // local q layout: [q0 | bx (ntl+halo layers) | by (ntl+halo layers)]
    i64 NphiAlloc;         // plane*(ntl+halo)

    __host__ __device__ void set(i64 ny_, i64 nx_, i64 nt_, i64 t0_, i64 ntl_, i64 py_ = 0, i64 cpad_ = 0) {
        ny = ny_; nx = nx_; nt = nt_; t0 = t0_; ntl = ntl_;
        py = (py_ > ny_) ? py_ : ny_;
        pyb = (py > ny) ? py : ny - 1;
        first = (t0 == 0);
        last = (t0 + ntl == nt);
        ncl = last ? ntl - 1 : ntl;
        halo = last ? 0 : 1;
        plane = py * nx;
        bxLayer = py * (nx - 1);
        byLayer = pyb * nx;
        Nphi = plane * ntl;
        Nz = plane * ncl;
        Nc = Nz + (cpad_ > 0 ? cpad_ : 0);
        offBx = Nz;
        offBy = offBx + bxLayer * (ntl + halo);
        NqAlloc = offBy + byLayer * (ntl + halo);
        NphiAlloc = plane * (ntl + halo);
    }
};

inline int launch_blocks(i64 work, int threads, i64 cap = 1 << 20) {
    i64 b = (work + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace dotsocp
