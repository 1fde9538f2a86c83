// Host-side launchers of the gfx950 kernels (definitions in *.hip).
#pragma once
#include "common.h"
#include "defer.h"

namespace dotsocp {

// Scalars of one inPALM iteration (solver_socp_inPALM.m:53-59,96-97,194-215).
struct LoopCoef {
    double s;      // scaleBF = E / D
    double sf;     // s / sqrt(2)
    double dF;     // scaleD = E / dScale
    double at, ax, ay;   // D/ht, D/hx, D/hy  (entries of D * grad, initialize.m:67-87)
    double tau;
    double c1, c2;       // (1+) 2 s^2, (1+) s^2  -- diagonal of I + s^2 F*B*BF (oper_q.m:14-23); without the 1 when weighted
    double dinv1, dinv2; // 1/c1, 1/c2 (unweighted diagQInv)
};

// Number of partial-sum slots produced by the KKT kernels.
enum {
    S_Q2 = 0, S_Z2, S_APHI2, S_ALPHA2, S_BETA2, S_FBBETA2, S_PRIM1, S_PRIM2, S_DUAL1, S_DUAL2,
    S_COMPLEM, S_DOTCOMP, S_RHO2, S_RHOFQ2, S_MRHOB, S_M2, S_RHOB2, S_QALPHA, S_CPHI, S_PHI2,
    S_COUNT
};

struct KktCoef {
    double sigma;
    double kappa;      // sigma * cScale * D      (compute_kkt_dot_complement.m:2)
    double dsD;        // dScale / D
    double dsE;        // dScale / E
};

// ---------------- cone.hip ----------------
int launch_proj_soc(double *out, const double *in, i64 M, i64 K, hipStream_t st);
int launch_bfd(const Grid &g, double *z, const double *q, double s, double dF, hipStream_t st);
// tail_bx / tail_by (time slabs, not the first): raw partial sums w4(x+1) + w5(x) resp. w8(y+1) + w9(y) of the left
// neighbour's last cell layer (launch_kkt_tail's bt_bx / bt_by)
int launch_bfd_conj(const Grid &g, double *q, const double *w, double s, hipStream_t st, const double *tail_bx = nullptr,
                    const double *tail_by = nullptr);
// z = Pi_Q(B F q + d - beta), B F q + d regenerated from q on the fly (solver_socp_inPALM.m:199)
int launch_cone_proj(const Grid &g, const LoopCoef &c, const double *q, const double *beta, double *z,
                     hipStream_t st);
// beta += tau * (z - (B F q + d))    (solver_socp_inPALM.m:212-215)
int launch_beta_update(const Grid &g, const LoopCoef &c, const double *q, const double *z, double *beta,
                       hipStream_t st);
// partial adjoint sums of the LAST owned cell layer for the right neighbour (time-slab mode)
int launch_gather_tail(const Grid &g, const double *z, const double *beta, double *tail_bx, double *tail_by,
                       hipStream_t st);

// ---------------- fused.hip ----------------
// Tile geometry of the fused cone kernel; the q-step needs it to find the edges whose adjoint
// sum is split between q2 (own tile's part) and the side buffers sx / sy (neighbour tile's part).
struct FusedGeom {
    int XB;               // tile width in x (columns per workgroup); tile height in y is 64
    i64 nyblk, nxblk;     // tiles in y / x
    i64 TC, chunks;       // time cells per chunk, number of chunks
    i64 sx_len, sy_len;   // side buffer lengths (doubles)
};
struct FusedArgs {
    const double *q_old;    // q^{k-1} (modes 1, 2)
    const double *q;        // q^k
    const double *beta_in;
    double *beta_out;       // modes 1, 2 (mode 1: must differ from beta_in when chunks > 1)
    double *z_out;          // mode 2
    double *q2;             // modes 0, 1: adjoint sums, q layout
    double *sx, *sy;        // modes 0, 1: tile-boundary partial sums
    const double *q3;       // mode 5: q~^k, the argument of the new projection (q then is q^k of the multiplier step)
    double *p2, *sxp, *syp; // modes 5, 6: second gather, F*B*((1 + tau) z + beta), layout of q2 / sx / sy
    i64 TC;
    // pending scaling of beta_in (sigma update / rescale block, solver_socp_inPALM.m:176,313), applied on
    // load exactly like k_scale would have: b = b * bmul / bdiv
    int bpend;              // number of pending operations (0, 1 or 2: a sigma update followed by a rescale)
    double bmul, bdiv;
    double bmul2, bdiv2;    // second pending operation, applied after the first
    int xcd;                // permute the tile order so that y-neighbouring tiles share an XCD (device_utils.h)
    int z0;                 // first chunk of this launch (launch_cone_fused can launch a range of chunks)
    // chunks launched ONE PER LAUNCH in ascending order on one stream (time slabs): the last cell's "t + 1" cone entries
    // w3, w4, w7, w8 travel to the next chunk through four layer planes instead of that chunk recomputing the cell
    const double *carry_in; // read by a chunk that is not the first (nullptr: recompute the cell in front)
    double *carry_out;      // written by a chunk that is not the last (may be the buffer carry_in points to)
};
// rows that are not a multiple of 16 doubles (128 bytes): neighbouring tiles share cache lines (2^k+1 grids)
bool tile_xcd_remap(const Grid &g);
int fused_geometry(const Grid &g, FusedGeom &fg);
// mode 0: projection + gather; 1: deferred beta update + projection + gather; 2: materialise beta and z;
// 3: z only, from (q_old, beta_in); 4: deferred beta update + gather of (z^k + beta^k) (PALM's first q-step)
// [z0, z0 + zcount) = the chunks to launch (zcount < 0: all from z0 on); chunks are independent of each other
int launch_cone_fused(int mode, const Grid &g, const LoopCoef &c, const FusedGeom &fg, FusedArgs a,
                      hipStream_t st, i64 z0 = 0, i64 zcount = -1);
// time-slab mode: split every slab's cone pass into >= 2 chunks so that the chunks in front of the last one --
// which alone reads the q halo -- can start before the halo has arrived
bool cone_split_enabled();

// ---------------- acc.hip (acc-ADMM loop) ----------------
struct KktWork;
struct AccArgs {
    const double *q;                 // q^+ of this iteration (modes 0, 1)
    const double *z_in, *beta_in;    // current state
    const double *z0, *beta0;        // Halpern anchors (mode 1)
    double *z0_out, *beta0_out;      // mode 3: the anchors, written
    double bdiv;                     // mode 3: factor of the sigma update
    // mode 0 on an iteration that ends with a KKT check (one slab; launch_acc_cone_kkt): the cell part of the KKT sums and the
    // F*B*beta^+ terms of all edges are taken while z^+, beta^+ are in registers (k_kkt_cells<., true>'s sums and gather)
    KktCoef kk;
    const double *alpha_p, *weight;  // alpha^+ of the q-step, the weight field (or nullptr)
    double *partials;
    int nostore;                     // ... and z^+, beta^+ are not written (the pass after the block recomputes them)
    double *z_out, *beta_out;        // mode 0: z^+, beta^+; mode 1: new state (buffers other than the inputs)
    double *q2, *sx, *sy;            // modes 1, 2: adjoint sums for the next q-step
    i64 TC;
    double c1, c2, om_rho, rho;      // Halpern weights: 1/(k+2), (k+1)/(k+2), 1 - rho, rho
};
struct AccCoef {
    double c1, c2, om_rho, rho;      // as above (Halpern) or c1 = theta/(2(k+theta)), c2 = k/(k+theta)
    double om_c1, c1c2;              // 1 - c1, c1 + c2 (theta != 2)
};
// mode 0: multiplier + z-step, raw outputs; 1: + Halpern step + gather; 2: gather of (z + beta) only;
// 3: mode 1 after a sigma update (beta, beta^+ divided by a.bdiv, anchors = x^+ stored)
int launch_acc_cone(int mode, const Grid &g, const LoopCoef &c, const FusedGeom &fg, AccArgs a, hipStream_t st);
// mode 0 + KKT sums: region 1 of the partial sums (cells, edges inside the tiles) and regions 2, 3 (edges on tile borders:
// F*B*beta^+ only); a.q2 / sx / sy are scratch
int launch_acc_cone_kkt(const Grid &g, const LoopCoef &c, const FusedGeom &fg, AccArgs a, const KktWork &w, hipStream_t st);
// Halpern step right after a sigma update: x, x^+ divided by div, anchor = x^+ stored, x extrapolated with k = 0
int launch_acc_restart(double *x, const double *xp, double *anchor, i64 n, const AccCoef &k, double div, hipStream_t st);
// element-wise extrapolation of one state array (modes: see acc.hip)
int launch_acc_interp(double *x, const double *xp, double *aux, i64 n, const AccCoef &k, int mode, int write_aux,
                      hipStream_t st);

// ---------------- transfer.hip (driver steps on the device) ----------------
// gf: the fine slab; phic / betac hold the coarse layers tc0, tc0 + 1, ... (a single coarse slab: its own arrays, tc0 = 0;
// Nzc = doubles per cone column in betac, < 0: gc.Nz); gc is read for ny, nx only
int launch_prolong_phi(const Grid &gf, const Grid &gc, const double *phic, double *phif, double sc_in, double sc_out,
                       hipStream_t st, i64 tc0 = 0);
int launch_prolong_beta(const Grid &gf, const Grid &gc, const double *betac, double *betaf, double *neg, double sc_in0,
                        double sc_in1, double sc_out, hipStream_t st, i64 tc0 = 0, i64 Nzc = -1);
int launch_scale_div(double *x, const double *w, i64 n, double sc, hipStream_t st);
int launch_fill(double *x, i64 n, double v, hipStream_t st);
// slab-aware: `out` holds the slab's own layers (ntl node layers resp. ncl cell layers); a_prev: launch_out_tail of the
// left neighbour (nullptr on the first slab)
int launch_outputs(const Grid &g, const double *q, const double *alpha, const double *weight, const double *rho0,
                   const double *rho1, const double *a_prev, double sig, double cD, double dD, int which, double *out,
                   hipStream_t st);
int launch_out_tail(const Grid &g, const double *alpha, const double *weight, double sig, double cD, double *out,
                    hipStream_t st);

// ---------------- stencil.hip ----------------
// q-step + alpha update reading the precomputed adjoint sums q2 (+ side buffers) of the fused kernel;
// writes q^{k+1} into q_out (q^k stays intact for the deferred beta update).
int launch_qstep_fused(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi,
                       const double *q2, const double *sx, const double *sy, const double *weight,
                       const double *tail_bx, const double *tail_by, double *q_out, double *alpha,
                       hipStream_t st);
// q-step + alpha update (alpha_in -> alpha_out, distinct buffers) + rhs of the next iteration's phi-step
// ex (optional): a scaling of alpha_in that is still pending in memory (applied on load), and -- partials != nullptr, one
// slab only -- the KKT variant: the sums of the KKT block that need only phi, q^+, alpha^+, A phi and c are accumulated in
// the same pass (one row of S_COUNT partial sums per workgroup at `partials`), r = A' alpha^+ - c goes to `resid`
struct QStepExtra {
    int apend;
    double amul, adiv;
    double *partials, *resid;
    double kappa, dsD;       // KktCoef
    double *u0_tail;         // time slabs: w.*q0^+ - alpha0^+ of the last owned cell layer goes here too (the right slab's rhs)
};
int launch_qstep_rhs(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi, const double *q2,
                     const double *sx, const double *sy, const double *weight, const double *tail_bx,
                     const double *tail_by, const double *cvec, double *q_out, const double *alpha_in, double *alpha_out,
                     double *rhs, hipStream_t st, i64 z0 = 0, i64 zcount = -1, i64 zstride = 1,
                     const QStepExtra *ex = nullptr);
i64 qstep_rhs_blocks(const Grid &g, const FusedGeom &fg);
// rhs <- (rhs + r) - r / factor, c <- c / factor  (sigma update without a new pass over q and alpha)
int launch_rhs_sigma_fix(double *rhs, const double *r, double *cvec, i64 n, double factor, hipStream_t st);
// chunks of time layers of that launch (z0 + i * zstride, i < zcount, selects chunks; chunks are independent of each other: only
// chunk 0 reads the adjoint tails of the left neighbour slab and only the last one the phi halo of the right one -- and
// writes the u0 tail)
i64 qstep_rhs_chunks(const Grid &g, const FusedGeom &fg, i64 *TC = nullptr);
// acc-ADMM: q-step + multiplier + next rhs (var 1: raw q^+, alpha^+; var 2: raw q^+ plus the Halpern step of q in
// place in q_state and of alpha into alpha_out, buffers other than alpha_in)
int launch_qstep_rhs_acc(int var, const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi,
                         const double *q2, const double *sx, const double *sy, const double *weight, const double *cvec,
                         double *q_raw, const double *alpha_in, double *alpha_out, double *rhs, double *q_state,
                         const double *q_anchor, const double *alpha_anchor, const AccCoef &k, hipStream_t st,
                         const double *tail_bx = nullptr, const double *tail_by = nullptr, double *u0_tail = nullptr);
int launch_rhs_fixup(const Grid &g, const LoopCoef &c, const double *u0_prev, double *rhs, hipStream_t st);
// PALM (solver_socp_PALM.m:196-200,137): first q-step without the alpha update; tmp_q = A phi in q layout
int launch_qstep_palm_first(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *phi, const double *q2,
                            const double *sx, const double *sy, const double *cvec, double *q_out, const double *alpha,
                            double *rhs, hipStream_t st, const double *tail_bx = nullptr, const double *tail_by = nullptr,
                            const double *qk = nullptr);   // qk: q2 / sx / sy hold k_cone_fused's second gather (modes 5, 6)
int launch_grad(const Grid &g, const LoopCoef &c, const double *phi, double *out, hipStream_t st);
// time-slab mode: complete the adjoint sums of the last owned cell for the right neighbour (values times sf)
int launch_tail_finalize(const Grid &g, const LoopCoef &c, const FusedGeom &fg, const double *q2, const double *sx,
                         const double *sy, double *tail_bx, double *tail_by, hipStream_t st);
// time-slab mode, KKT block: alpha0, w.*alpha0 and raw adjoint partial sums of beta of the last owned cell layer
int launch_kkt_tail(const Grid &g, const double *alpha, const double *beta, const double *weight, double *a0,
                    double *a0w, double *bt_bx, double *bt_by, hipStream_t st);
// rhs = A'(w.*q - alpha) + c   (solver_socp_inPALM.m:194, solver_wsocp_inPALM.m:200)
int launch_rhs(const Grid &g, const LoopCoef &c, const double *q, const double *alpha, const double *cvec,
               const double *weight, const double *u0_prev, double *rhs, hipStream_t st);
// q-step + alpha update (solver_socp_inPALM.m:204-206,211,214; solver_wsocp_inPALM.m:210-217)
int launch_qstep(const Grid &g, const LoopCoef &c, const double *phi, const double *z, const double *beta,
                 const double *weight, const double *tail_bx, const double *tail_by, double *q,
                 double *alpha, hipStream_t st);
// u0 = w.*q0 - alpha0 of the last owned cell layer (halo for the right neighbour's rhs)
int launch_u0_tail(const Grid &g, const double *q, const double *alpha, const double *weight, double *out,
                   hipStream_t st);
int launch_scale(double *x, i64 n, double mul, double div, hipStream_t st);   // x = x * mul / div
// slab rows <-> per-peer contiguous staging of the slab <-> pencil transposes (one process per slab)
#define DS_MAX_WORLD 64
struct PencilCuts {
    int world;
    i64 cut[DS_MAX_WORLD + 1];
};
int launch_pencil_pack(bool pack, const PencilCuts &pc, i64 plane, i64 ntl, double *slab, double *stage, hipStream_t st);

// ---------------- tri.hip: time-slab Poisson solve by partitioned tridiagonal systems (no transposes) ----------------
#define TRI_EXTRA 256     // room for the whole (0, 0) line of a slab in a message: slabs of at most 256 time nodes
int launch_tri_local(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, const PencilCuts &pc,
                     const double *r, double *send, hipStream_t st);
int launch_tri_reduced(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, const PencilCuts &pc,
                       int rank, i64 l0, i64 nl, const i64 *slab_n, const double *recv, double *back, double *zero_work,
                       hipStream_t st, const double *own_recv = nullptr, double *own_back = nullptr);
int launch_tri_final(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, const PencilCuts &pc,
                     const double *back, double *x, hipStream_t st);
// the SINGLE slab's t-axis solve by the same elimination, in place on x = [g.plane][nt] (no transform along t; any nt <= 512)
bool tsolve_tri_supported(i64 nt);
bool tsolve_tri_preferred(i64 nt, bool pow2, i64 plane);      // faster than the transform pass(es) along t?
int launch_tsolve_tri(const Grid &g, i64 nt, double kscale, const double *cy, const double *cx, double *x, hipStream_t st);
// up to DS_MAX_WORLD messages copied by ONE launch on the receiving slab's stream: message m = count[m] doubles from
// src[m] (this or a peer device) to dst[m] -- the exchanges between the slabs of one process (one launch per receiver
// instead of one event-ordered copy per message)
struct GatherMsgs {
    const double *src[DS_MAX_WORLD];
    double *dst[DS_MAX_WORLD];
    i64 count[DS_MAX_WORLD];
    int n;
};
int launch_gather_msgs(const GatherMsgs &m, hipStream_t st);

// ---------------- kkt.hip ----------------
#define KKT_SLICES 64
struct KktWork {
    double *partials;   // [maxBlocks][S_COUNT]
    i64 maxBlocks;
    double *sums;       // [S_COUNT] device, followed by [KKT_SLICES][S_COUNT] intermediate sums (launch_kkt_final)
};
// Halo layers from the LEFT neighbour slab's last cell (all nullptr on the first slab):
// alpha0, w.*alpha0, and the partial adjoint sums of beta (columns 4,5 / 8,9).
struct KktHalo {
    const double *a0_prev, *a0w_prev, *btail_bx, *btail_by;
};
i64 kkt_partials_needed(const Grid &g);
// parts: bit mask of 1 node, 2 cell (+ q0 entries; needs a stored z), 4 bx edges, 8 by edges
// layer0: only the slab's first node / edge layer (the folded path on time slabs leaves it to this launch; parts 1|4|8),
// partial sums in their own regions; resid (part 1): A' alpha - c per visited node
int launch_kkt(const Grid &g, const LoopCoef &c, const KktCoef &k, const double *phi, const double *q,
               const double *alpha, const double *z, const double *beta, const double *cvec,
               const double *weight, const KktHalo &halo, const KktWork &w, int parts, hipStream_t st,
               bool layer0 = false, double *resid = nullptr);
// the node / q0-entry / edge sums that need no multiplier (parts 1, 16, 4, 8 of k_kkt without the F*B*beta terms): regions
// 0, 4, 5, 6 of the partial sums -- what is left when the cone pass has taken the cell sums (launch_acc_cone_kkt)
int launch_kkt_nodual(const Grid &g, const LoopCoef &c, const KktCoef &k, const double *phi, const double *q,
                      const double *alpha, const double *cvec, const double *weight, const KktWork &w, hipStream_t st);
// tile-border edges: F*B*beta terms from the raw partial sums in q2 / sx / sy (regions 2, 3)
int launch_kkt_bnd_dual(const Grid &g, const LoopCoef &c, const KktCoef &k, const FusedGeom &fg, const double *q,
                        const double *alpha, const double *weight, const double *q2, const double *sx, const double *sy,
                        const KktWork &w, hipStream_t st);
// fused path: pending multiplier step (beta_in -> beta_out, distinct buffers) + the cell part of the sums
// edges (one slab, after a q-step in its KKT variant): also the F*B*beta' sums of every edge and the momentum terms of
// the edges on tile borders (a.q2 / a.sx / a.sy are scratch; q_new = q^{k+1})
int launch_kkt_cells_update(const Grid &g, const LoopCoef &c, const KktCoef &k, const FusedGeom &fg, FusedArgs a,
                            const double *phi, const double *alpha, const double *weight, const KktWork &w,
                            hipStream_t st, bool edges = false, const double *q_new = nullptr);
// the rescale block's five norms of an iterate whose multiplier step is pending (a.q_old, a.q, a.beta_in [+ pending ops]):
// S_PHI2, S_Q2, S_ALPHA2, S_Z2, S_BETA2 as partial sums (clear w.partials first, launch_kkt_final afterwards)
int launch_norms(const Grid &g, const LoopCoef &c, const KktCoef &k, const FusedGeom &fg, FusedArgs a, const double *phi,
                 const double *alpha, const double *weight, const KktWork &w, hipStream_t st);
// partial sums of the q-step's KKT variant: region 0 of w.partials
double *kkt_qstep_partials(const Grid &g, const KktWork &w);
int launch_kkt_final(const Grid &g, const KktWork &w, hipStream_t st);

// ---------------- dct.hip ----------------
struct DctPlan;   // twiddles / dense matrices for one axis length
DctPlan *dct_plan_create(i64 n);
void dct_plan_destroy(DctPlan *p);
// Orthonormal DCT-II (inverse=0) / DCT-III (inverse=1) along one axis of an [n0][n1][n2] array
// (n0 fastest), src -> dst.  axis = 0, 1 or 2.  src == dst is allowed for power-of-two and prime-factor lengths.
// pitch0 > n0: the rows of both arrays are pitch0 doubles apart ([pitch0][n1][n2] with n0 valid entries per row);
// power-of-two n0 is never pitched.
int launch_dct_axis(const DctPlan *p, const double *src, double *dst, i64 n0, i64 n1, i64 n2, int axis,
                    int inverse, hipStream_t st, i64 pitch0 = 0);
// Power-of-two nt only: DCT-II along t, spectral division, DCT-III along t in ONE pass over a
// pencil [nl][nt]: the columns line0 .. line0+nl-1 of the ny*nx = nplane (y, x) columns (y fastest),
// all nt time nodes; src -> dst (may alias).
bool dct_plan_is_pow2(const DctPlan *p);
bool dct_plan_has_tsolve(const DctPlan *p);   // fused forward / divide / inverse pass along t available
int launch_dct_t_solve(const DctPlan *p, const double *src, double *dst, i64 ny, i64 nplane, i64 line0, i64 nl,
                       i64 nt, double kscale, const double *cy, const double *cx, const double *ct, hipStream_t st,
                       i64 pitch0 = 0);   // pitch0 > ny: whole layers (line0 = 0, nl = nplane) with pitched rows
// same pencil, non-power-of-two nt: spectral division only (between two dense DCT passes)
int launch_spectral_divide_pencil(double *data, i64 ny, i64 nplane, i64 line0, i64 nl, i64 nt, double kscale,
                                  const double *cy, const double *cx, const double *ct, hipStream_t st);
// data[i] /= kscale * lambda(i)  with lambda the DCT eigenvalues of initialize_FFTkernel.m:6-15
// for global dims (ny, nx, nt); the local block covers x in [x0, x0+nxl) (pencil mode) and all y, t.
int launch_spectral_divide(double *data, i64 ny, i64 nx, i64 nt, i64 x0, i64 nxl, double kscale,
                           const double *cy, const double *cx, const double *ct, hipStream_t st, i64 pitch0 = 0);

}  // namespace dotsocp
