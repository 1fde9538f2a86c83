// Adjoint gather (F* B* w) inside a t-marching 64 x XB tile kernel: shared by the fused inPALM cone
// kernel (fused.hip) and the acc-ADMM cone kernel (acc.hip).
//
// Called once per time step tl with the ten cone entries w[] of the cell (y, x, tl) held by this
// thread (all zero on the virtual step tl == ncl).  Emits edge layer tl of bx and by:
//   bx(y, x+1/2, tl) = sf * (((w1(x+1) + w2(x)) + w3'(x+1)) + w4'(x)),   ' = cell tl-1 (carried in registers)
//   by(y+1/2, x, tl) = sf * (((w5(y+1) + w6(y)) + w7'(y+1)) + w8'(y))
// x-neighbours are exchanged through LDS (double-buffered, one barrier per step), y-neighbours
// through cross-lane shuffles.  Edges on a tile boundary get the own tile's raw partial in q2 and
// the neighbour tile's raw partial in the side buffers sx / sy; the q-step completes them.
#pragma once
#include "common.h"

namespace dotsocp {

struct GatherCarry {
    double p3 = 0.0, p4 = 0.0, p7 = 0.0, p8 = 0.0;   // "t+1" cone entries of the previous cell
    int par = 0;
};

// the two halves of gather_emit: the LDS hand-off in front of the workgroup barrier, everything else behind it (a kernel
// with two gathers per step posts both, waits once, finishes both)
template <int XB>
__device__ __forceinline__ void gather_post(double2 (&xch)[2][XB][64], const GatherCarry &gc, const double (&w)[10], int xl,
                                            int lane) {
    xch[gc.par][xl][lane] = make_double2(w[1], gc.p3);
}

template <int XB>
__device__ __forceinline__ void gather_finish(const Grid &g, double sf, double2 (&xch)[2][XB][64], GatherCarry &gc,
                                            const double (&w)[10], i64 tl, bool store, i64 x, i64 y, int xl,
                                            int lane, i64 nxblk, i64 nyblk, unsigned bx_blk, unsigned by_blk,
                                            double *__restrict__ q2, double *__restrict__ sx,
                                            double *__restrict__ sy) {
    if (store) {
        if (x < g.nx - 1) {
            const i64 e = g.offBx + g.bxLayer * tl + y + g.py * x;
            if (xl < XB - 1) {
                const double2 r = xch[gc.par][xl + 1][lane];
                double acc = r.x + w[2];
                acc += r.y;
                acc += gc.p4;
                q2[e] = sf * acc;
            } else {
                q2[e] = w[2] + gc.p4;                        // partial; the right tile adds its part via sx
            }
        }
        if (xl == 0 && x > 0) sx[(tl * nxblk + bx_blk) * g.ny + y] = w[1] + gc.p3;
    }
    const double u5 = __shfl_down(w[5], 1, 64), u7 = __shfl_down(gc.p7, 1, 64);
    if (store) {
        if (y < g.ny - 1) {
            const i64 e = g.offBy + g.byLayer * tl + y + g.pyb * x;
            if (lane < 63) {
                double acc = u5 + w[6];
                acc += u7;
                acc += gc.p8;
                q2[e] = sf * acc;
            } else {
                q2[e] = w[6] + gc.p8;                        // partial; the upper tile adds its part via sy
            }
        }
        if (lane == 0 && y > 0) sy[(tl * g.nx + x) * nyblk + by_blk] = w[5] + gc.p7;
    }
    gc.p3 = w[3]; gc.p4 = w[4]; gc.p7 = w[7]; gc.p8 = w[8];
    gc.par ^= 1;
}

template <int XB>
__device__ __forceinline__ void gather_emit(const Grid &g, double sf, double2 (&xch)[2][XB][64], GatherCarry &gc,
                                            const double (&w)[10], i64 tl, bool store, i64 x, i64 y, int xl,
                                            int lane, i64 nxblk, i64 nyblk, unsigned bx_blk, unsigned by_blk,
                                            double *__restrict__ q2, double *__restrict__ sx,
                                            double *__restrict__ sy) {
    gather_post<XB>(xch, gc, w, xl, lane);
    __syncthreads();
    gather_finish<XB>(g, sf, xch, gc, w, tl, store, x, y, xl, lane, nxblk, nyblk, bx_blk, by_blk, q2, sx, sy);
}

}  // namespace dotsocp
