// The fused cone kernel of the inPALM loop: deferred multiplier update + cone projection +
// adjoint gather in ONE pass over beta ("minimal fused dataflow"):
//
//   MODE_B (steady state), per cell, with q_old = q^{k-1}, q = q^k, beta_in = beta^{k-1}:
//     z^k      = Pi_Q(BF q_old + d - beta_in)          recomputed, z is never stored        (:199 of it. k-1)
//     beta^k   = beta_in + tau (z^k - (BF q + d))      multiplier step of iteration k-1     (:212-215)
//     z^{k+1}  = Pi_Q(BF q + d - beta^k)               z-step of iteration k                (:199)
//     q2       = F* B* (z^{k+1} + beta^k)              adjoint gather for the q-step        (:205)
//   traffic: beta in + beta out + q_old + q + q2 = 8 (20 Nz + 3 Nq) bytes; the beta streams non-temporal (device_utils.h:
//   ld_stream / st_stream; the cell entries of q and the gather's stores the same way: no further gain, measured).
//   MODE_A: same without the deferred update (beta already current): 8 (10 Nz + 2 Nq) bytes.
//   MODE_M (2): materialise -- beta update + write z (needed by the rescale block / outputs).
//   MODE_Z (3): z = Pi_Q(BF q_old + d - beta_in) only (z of the last iteration from the kept beta^k).
//   MODE_P (4): PALM's extra q-step (solver_socp_PALM.m:196-200): deferred beta update as in MODE_B, then
//               q2 = F* B* (z^k + beta^k) with the recomputed z^k -- no new projection.
//   MODE_P2 (5), MODE_P1 (6): PALM with ONE pass over beta per iteration (solver_palm.hip).  Mode 6 is MODE_A, mode 5 is MODE_B
//               with the new projection taken at a third array, q3 = q~^k (PALM projects at BF q~ + d, :209, and updates the
//               multiplier at BF q + d, :224); both emit a second gather p2 = F*B*((1 + tau) z^{k+1} + beta^k), from which
//               the next iteration's first q-step forms F*B*(z^{k+1} + beta^{k+1}) = p2 - tau F*B*(BF q^{k+1} + d) without a
//               pass over beta (F*B*BF is diagonal, F*B*d = 0).
//   (line numbers: socp/dot2d/algorithms/solver_socp_inPALM.m)
//
// Mapping: a workgroup owns a 64 (y) x XB (x) tile of cell columns and MARCHES through a chunk
// of time cells; one wavefront = 64 consecutive y of one x column, so every plane access is a
// coalesced 512-byte segment.  The t+1 edge layer of q is carried in registers (each q entry is
// fetched once per chunk), the cell's own "t+1" cone entries are carried to the next step in
// registers, x-neighbour cone entries are exchanged through LDS (double-buffered, one barrier
// per step) and y-neighbour entries through cross-lane shuffles.  Edges on a tile boundary get
// a partial sum in q2 plus the neighbour tile's partial in a small side buffer (sx / sy); the
// q-step adds the two.  Chunks after the first recompute the cell in front of them (reads only:
// beta is ping-ponged, so no other workgroup's writes are observed).
#include "device_utils.h"
#include "gather_tile.h"
#include "kernels.h"

#include <cstdlib>

namespace dotsocp {

template <int MODE, int XB, bool NT = false>
__global__ void __launch_bounds__(64 * XB) k_cone_fused(Grid g, LoopCoef c, FusedArgs a) {
    constexpr bool DUAL = (MODE >= 5);              // second gather (PALM)
    constexpr bool UPD = (MODE != 0 && MODE != 6);  // deferred multiplier step in front of the projection
    constexpr bool PROJ = (MODE < 2 || MODE >= 5);  // new projection + gather
    __shared__ double2 xch[2][XB][64];
    __shared__ double2 xchp[DUAL ? 2 : 1][DUAL ? XB : 1][DUAL ? 64 : 1];
    const int lane = threadIdx.x, xl = threadIdx.y;
    const BlockId blk = block_id(a.xcd != 0);
    const i64 y = (i64)blk.x * 64 + lane;
    const i64 x = (i64)blk.y * XB + xl;
    const bool inb = (y < g.ny) && (x < g.nx);
    const i64 yc = inb ? y : 0, xc = inb ? x : 0;      // clamped coordinates keep out-of-tile lanes harmless
    const i64 t0 = ((i64)blk.z + a.z0) * a.TC;
    const i64 t1 = (t0 + a.TC < g.ncl) ? t0 + a.TC : g.ncl;
    const bool lastChunk = (t1 == g.ncl);
    constexpr bool GATHER = (PROJ || MODE == 4);
    // chunks launched one after the other hand the "t + 1" cone entries of their last cell to the next chunk through
    // a.carry (four planes of one layer) instead of having it recompute that cell (reads of twenty planes)
    const bool carried = GATHER && !DUAL && t0 > 0 && a.carry_in != nullptr;
    const i64 tstart = (GATHER && t0 > 0 && !carried) ? t0 - 1 : t0;
    const i64 nxblk = gridDim.y, nyblk = gridDim.x;

    EdgeQuad cur = load_edges(g, a.q, yc, xc, tstart, c.sf), curo, cur3;
    if (UPD) curo = load_edges(g, a.q_old, yc, xc, tstart, c.sf);
    if (MODE == 5) cur3 = load_edges(g, a.q3, yc, xc, tstart, c.sf);
    GatherCarry gc, gcp;
    if (carried) {
        const i64 ci = yc + g.py * xc;
        gc.p3 = a.carry_in[ci]; gc.p4 = a.carry_in[g.plane + ci];
        gc.p7 = a.carry_in[2 * g.plane + ci]; gc.p8 = a.carry_in[3 * g.plane + ci];
    }
    // one extra virtual step (tl == ncl, no cell) on the last chunk emits the final edge layer
    const i64 tstop = (GATHER && lastChunk) ? t1 + 1 : t1;
    for (i64 tl = tstart; tl < tstop; ++tl) {
        const bool hasCell = tl < g.ncl;
        const bool own = tl >= t0;                      // false only for the recomputed cell in front of the chunk
        double w[10], wp[10];
        if (hasCell) {
            const i64 i = yc + g.py * (xc + g.nx * tl);
            const EdgeQuad nxt = load_edges(g, a.q, yc, xc, tl + 1, c.sf);
            double b[10], v[10];
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
            build_z2(v, a.q[i], cur, nxt, c.s, c.dF);
            if (UPD) {
                const EdgeQuad nxto = load_edges(g, a.q_old, yc, xc, tl + 1, c.sf);
                double zo[10];
                build_z2(zo, a.q_old[i], curo, nxto, c.s, c.dF);
#pragma unroll
                for (int j = 0; j < 10; ++j) zo[j] = zo[j] - b[j];
                proj_row<10>(zo);
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const double r = zo[j] - v[j];
                    b[j] = b[j] + c.tau * r;
                }
                if (own && inb) {
                    if (MODE != 3) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) st_stream<NT>(a.beta_out + j * g.Nc + i, b[j]);
                    }
                    if (MODE == 2 || MODE == 3) {
#pragma unroll
                        for (int j = 0; j < 10; ++j) st_stream<NT>(a.z_out + j * g.Nc + i, zo[j]);
                    }
                }
                curo = nxto;
                if (MODE == 4) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) w[j] = zo[j] + b[j];
                    if (own && inb) a.q2[i] = c.s * (w[9] - w[0]);
                }
            }
            cur = nxt;
            if (PROJ) {
                if (MODE == 5) {          // the projection's argument is BF q3 + d
                    const EdgeQuad nxt3 = load_edges(g, a.q3, yc, xc, tl + 1, c.sf);
                    build_z2(v, a.q3[i], cur3, nxt3, c.s, c.dF);
                    cur3 = nxt3;
                }
#pragma unroll
                for (int j = 0; j < 10; ++j) v[j] = v[j] - b[j];
                proj_row<10>(v);
#pragma unroll
                for (int j = 0; j < 10; ++j) w[j] = v[j] + b[j];
                if (own && inb) a.q2[i] = c.s * (w[9] - w[0]);
                if (DUAL) {
#pragma unroll
                    for (int j = 0; j < 10; ++j) wp[j] = w[j] + c.tau * v[j];
                    if (own && inb) a.p2[i] = c.s * (wp[9] - wp[0]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 10; ++j) w[j] = 0.0;
            if (DUAL) {
#pragma unroll
                for (int j = 0; j < 10; ++j) wp[j] = 0.0;
            }
        }
        if constexpr (GATHER && !DUAL)   // adjoint gather for edge layer tl (gather_tile.h)
            gather_emit<XB>(g, c.sf, xch, gc, w, tl, own && inb, x, y, xl, lane, nxblk, nyblk, blk.y, blk.x,
                            a.q2, a.sx, a.sy);
        if constexpr (DUAL) {            // two gathers, one barrier
            gather_post<XB>(xch, gc, w, xl, lane);
            gather_post<XB>(xchp, gcp, wp, xl, lane);
            __syncthreads();
            gather_finish<XB>(g, c.sf, xch, gc, w, tl, own && inb, x, y, xl, lane, nxblk, nyblk, blk.y, blk.x,
                              a.q2, a.sx, a.sy);
            gather_finish<XB>(g, c.sf, xchp, gcp, wp, tl, own && inb, x, y, xl, lane, nxblk, nyblk, blk.y, blk.x,
                              a.p2, a.sxp, a.syp);
        }
    }
    if (GATHER && !DUAL && !lastChunk && a.carry_out != nullptr && inb) {
        const i64 ci = y + g.py * x;
        a.carry_out[ci] = gc.p3; a.carry_out[g.plane + ci] = gc.p4;
        a.carry_out[2 * g.plane + ci] = gc.p7; a.carry_out[3 * g.plane + ci] = gc.p8;
    }
}

bool stream_nt_enabled() {
    const char *e = getenv("DOTSOCP_NT");          // read per launch: the tests switch it inside one process
    return !(e && atoi(e) == 0);
}

bool tile_xcd_remap(const Grid &g) {
    return (g.py % 16) != 0;
}

bool cone_split_enabled() { return true; }

int fused_geometry(const Grid &g, FusedGeom &fg) {
    fg.XB = 4;
    fg.nyblk = (g.ny + 63) / 64;
    fg.nxblk = (g.nx + fg.XB - 1) / fg.XB;
    const i64 tiles = fg.nyblk * fg.nxblk;
    // enough workgroups to fill 256 CUs several times over; each extra chunk costs one recomputed cell
    const i64 target = 2048;
    i64 chunks = (target + tiles - 1) / tiles;
    if (chunks < 1) chunks = 1;
    // a slab of a time-slab decomposition: at least two chunks (see cone_split_enabled)
    if (!(g.first && g.last) && cone_split_enabled() && chunks < 2 && g.ncl >= 12) chunks = 2;
    i64 TC = (g.ncl + chunks - 1) / chunks;
    if (TC < 8) TC = 8;
    if (TC > g.ncl) TC = g.ncl;
    if (TC < 1) TC = 1;
    fg.TC = TC;
    fg.chunks = (g.ncl + TC - 1) / TC;
    fg.sx_len = (g.ncl + 1) * fg.nxblk * g.ny;
    fg.sy_len = (g.ncl + 1) * g.nx * fg.nyblk;
    return 0;
}

int launch_cone_fused(int mode, const Grid &g, const LoopCoef &c, const FusedGeom &fg, FusedArgs a,
                      hipStream_t st, i64 z0, i64 zcount) {
    if (g.Nz <= 0) return 0;
    if (zcount < 0) zcount = fg.chunks - z0;
    if (z0 < 0 || zcount <= 0 || z0 + zcount > fg.chunks) return 0;
    a.TC = fg.TC;
    a.z0 = (int)z0;
    a.xcd = tile_xcd_remap(g) ? 1 : 0;
    dim3 grid((unsigned)fg.nyblk, (unsigned)fg.nxblk, (unsigned)zcount);
    dim3 blk(64, 4);
    const bool nt = stream_nt_enabled();
#define CONE_MODE(M)                                                                     \
    case M:                                                                              \
        if (nt) DS_KLAUNCH((k_cone_fused<M, 4, true>), grid, blk, 0, st, g, c, a);       \
        else DS_KLAUNCH((k_cone_fused<M, 4>), grid, blk, 0, st, g, c, a);                \
        break;
    switch (mode) {
        CONE_MODE(0) CONE_MODE(1) CONE_MODE(2) CONE_MODE(3) CONE_MODE(4) CONE_MODE(5) CONE_MODE(6)
#undef CONE_MODE
        default: set_error("bad fused mode"); return DOTSOCP_EINVAL;
    }
    DS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dotsocp
