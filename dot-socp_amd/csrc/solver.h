// Device-resident inPALM/ALG2 loop state (B1 boundary of include/dotsocp.h).
#pragma once
#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "hostmem.h"
#include "kernels.h"

namespace dotsocp {

// Device allocations of the solver state go through guard.hip: with DOTSOCP_CANARY=1 in the environment every buffer
// gets a guard band of NaN-pattern words on both sides, checked by canary_check() (finish(), destroy()) -- an
// out-of-bounds WRITE of a tile kernel on a partial tile changes a guard word, an out-of-bounds READ pulls NaNs
// into the iterates.  Without the variable freed buffers are kept in a per-device cache and handed out again (guard.hip;
// DOTSOCP_DEVICE_CACHE=0: plain hipMalloc / hipFree).
int guarded_malloc(void **p, size_t bytes);
void guarded_free(void *p);
// the device block cache behind them (guard.hip): hipFree everything it holds; returns the bytes given back
long long device_cache_release();
// number of live buffers whose guard bands were overwritten (0 when the canaries are off); `report` receives a
// description of the first few.  Synchronises the current device.
int canary_check(std::string *report);
bool canary_enabled();
// DOTSOCP_STRESS_STREAMS=1: random stalls in front of the work of every slab stream (guard.hip)
bool stream_stress_enabled();
void stream_stress(hipStream_t st);

template <class T>
inline int dmalloc(T **p, i64 n) {
    *p = nullptr;
    if (n <= 0) n = 1;
    return guarded_malloc((void **)p, sizeof(T) * (size_t)n);
}

inline int dzalloc(double **p, i64 n, hipStream_t st) {
    DS_CHECK(dmalloc(p, n));
    DS_HIP(hipMemsetAsync(*p, 0, sizeof(double) * (size_t)(n > 0 ? n : 1), st));
    return 0;
}

inline void dfree(void *p) {
    if (p) guarded_free(p);
}

// Resources that live once per HIP device and are shared by the slabs placed on it.
struct DevRes {
    int dev = 0;
    DctPlan *py = nullptr, *px = nullptr, *pt = nullptr;
    double *cy = nullptr, *cx = nullptr, *ct = nullptr;   // DCT eigenvalue tables
};

// One time slab (common.h: Grid).  Single-GPU runs have exactly one; `nslabs` > 1 keeps several in
// one process -- every slab with its own device (dotsocp_create_multi: slab r on device (first + r) mod #devices;
// dotsocp_create: all on one device), its own pair of streams and peer copies between neighbours as
// "communication"; with an RCCL communicator attached the process holds the single slab `rank` of `world`.
#define DS_XEV 8
struct Slab {
    int index = 0;              // global slab number
    int dev = 0;                // HIP device this slab lives on
    hipStream_t st = nullptr;   // main stream of the slab (slab 0: Solver::stream)
    hipStream_t st_z = nullptr; // second stream: carries the slab's communication in time-slab mode (Solver::comm_z)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_halo = nullptr;
    hipEvent_t ev_cjoin = nullptr;        // synchronous communication on the second stream: "done" (Solver::comm_leave)
    hipEvent_t xev[DS_XEV] = {nullptr};   // ordering of cross-slab copies (Solver::xcopy), used round-robin
    hipEvent_t ev_tri = nullptr;          // "this slab's interface message is written" (Solver::tri_exchange, slabs of one process)
    hipEvent_t ev_msg = nullptr, ev_got = nullptr;   // batched neighbour exchanges: "my messages are written" / "I have pulled mine"
    int xev_next = 0;
    DevRes *res = nullptr;      // plans / tables of `dev`
    double *h_sums = nullptr;   // pinned host copy of this slab's KKT partial sums [S_COUNT]
    Grid g;
    double *phi = nullptr;      // NphiAlloc (owned nodes + halo layer)
    double *q = nullptr;        // NqAlloc
    double *alpha = nullptr;    // NqAlloc
    double *z = nullptr;        // 10 * Nz
    double *beta = nullptr;     // 10 * Nz
    double *c = nullptr;        // Nphi
    double *weight = nullptr;   // NqAlloc (weighted only)
    double *w0 = nullptr, *w1 = nullptr;   // Poisson work arrays (slab layout)
    // Poisson t-axis: this slab's pencil = columns [l0, l0+nl) of the ny*nx (y, x) columns, all nt nodes
    i64 l0 = 0, nl = 0;
    double *pencil = nullptr, *pencil2 = nullptr;   // nl * nt each (pencil2: dense / RCCL staging)
    double *stage = nullptr;                        // Nphi (RCCL pack buffer)
    // halos received from the LEFT neighbour (nullptr on the first slab)
    double *u0_prev = nullptr, *tail_bx = nullptr, *tail_by = nullptr;
    double *a0_prev = nullptr, *a0w_prev = nullptr, *btail_bx = nullptr, *btail_by = nullptr;
    // staging for what this slab sends to the RIGHT neighbour (nullptr on the last slab)
    double *send_plane = nullptr, *send_plane2 = nullptr, *send_bx = nullptr, *send_by = nullptr;
    KktWork kw{};
    // fused path (fused.hip): q^{k-1}, adjoint sums, ping-pong beta, tile-boundary side buffers
    double *q_old = nullptr, *q2 = nullptr, *beta2 = nullptr, *sx = nullptr, *sy = nullptr;
    double *q3 = nullptr, *p2 = nullptr, *sxp = nullptr, *syp = nullptr;    // PALM, one pass over beta per iteration (solver_palm.hip)
    // ... on time slabs: the second gather's share of the neighbour's first edge layer (sent to the right / received from the left)
    double *send_pbx = nullptr, *send_pby = nullptr, *ptail_bx = nullptr, *ptail_by = nullptr;
    double *alpha2 = nullptr;   // ping-pong partner of alpha (q-step that also forms the next rhs)
    // partitioned tridiagonal t-solve (tri.hip): messages to / from the owners of the modes, zero-mode work line
    double *tri_send = nullptr, *tri_recv = nullptr, *tri_bsend = nullptr, *tri_brecv = nullptr, *tri_zero = nullptr;
    double *carry = nullptr;    // 4 layer planes: hand-off between the chunk launches of a cone pass (FusedArgs::carry_in / _out)
    FusedGeom fg{};
    // acc-ADMM loop (solver_acc.hip): x^+ of the iteration (q^+ lives in q_old, beta^+ in beta2) and the
    // Halpern anchors / previous extrapolation points
    double *phi_p = nullptr, *alpha_p = nullptr, *z_p = nullptr;
    double *phi_a = nullptr, *q_a = nullptr, *alpha_a = nullptr, *z_a = nullptr, *beta_a = nullptr;
};

// every slab this process holds, with the slab's device made current first (member functions returning int)
#define FOR_SLABS(s) for (auto &s : slabs) if (int rc_use__ = use(s)) return rc_use__; else

bool if_adjust_sigma(double iter, double last_iter);   // IfAdjustSigma (solver_socp_inPALM.m:361-379)

enum Phase { PH_RHS = 0, PH_POISSON, PH_PROJ, PH_QSTEP, PH_BETA, PH_KKT, PH_FUSED_A, PH_FUSED_B, PH_MATERIALISE,
             PH_COMM, PH_INTERP, PH_ACC_CONE, PH_ACC_GATHER, PH_QSTEP0, PH_TRANSPOSE, PH_COUNT };

struct Solver {
    dotsocp_problem prob{};
    int device = 0;
    i64 ny = 0, nx = 0, nt = 0;     // internal dims (1-D problems: ny = nx1d, nx = 1)
    // streams of slab 0 (created by init() on `device`; the other in-process slabs create their own): the main stream
    // carries everything but the overlapped cone pass, and all RCCL communication
    hipStream_t stream = nullptr;
    hipStream_t stream_z = nullptr;    // cone pass when it overlaps the phi step
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_halo = nullptr;
    bool multi_device = false;         // dotsocp_create_multi: slab r on device (device + r) mod #devices
    int ndev_visible = 1;
    std::vector<DevRes *> devres;      // one per device in use
    int cur_dev = -1;                  // device made current by use() (-1: unknown)
    int use(const Slab &s);            // hipSetDevice(s.dev) unless it already is the current one
    int use_dev(int d);
    DevRes *res_for(int dev);          // plans + tables on `dev` (created on first request)
    // in-process slabs: dst (on `to`) <- src (on `from`), ordered after everything enqueued so far on both slabs'
    // main streams and before everything enqueued later on either (what one shared stream used to give for free)
    int xcopy(Slab &from, const double *src, Slab &to, double *dst, i64 count);
    int xcopy2d(Slab &from, const double *src, size_t spitch, Slab &to, double *dst, size_t dpitch, size_t width,
                size_t height);
    i64 column_pad() const;
    i64 row_pitch() const;             // row pitch of this context's device arrays (ny unless the single slab is pitched)
    int sync_all();                    // host waits for every stream of every slab
    // rows of `rowlen` doubles between a device array with rows `pitch` apart and a host array in the reference layout
    int copy_rows(double *dev, double *host, i64 rowlen, i64 pitch, i64 nrows, bool up, hipStream_t st);
    // Time slabs: every message between slabs travels on the slab's SECOND stream (RCCL calls, pull launches, peer copies),
    // kernels stay on the main stream.  A communication step is bracketed by comm_enter() / comm_leave(): the second
    // stream first waits for what the main stream holds (fork), and the main stream then waits for the messages (join) --
    // the synchronous form every caller gets by default.  Solver::step() instead runs the iteration's exchanges
    // asynchronously (comm_async): it forks, issues the exchange, marks an event on the second stream, enqueues kernels
    // that do not need the message on the main stream and lets the main stream wait for the mark only in front of the
    // first kernel that does -- the messages travel while kernels run, and no two kernels ever compete for the compute
    // units (measured: a cone pass on a second stream stretched the whole-CU DCT passes beside it from 60 to 400 us)
    bool comm_z = false;               // communication on the second streams (time-slab mode; DOTSOCP_OVERLAP=0: on the main stream)
    bool comm_async = false;           // inside an asynchronous exchange of step(): comm_enter / comm_leave do nothing
    int comm_depth = 0;
    hipStream_t cst(const Slab &s) const { return comm_z ? s.st_z : s.st; }
    int comm_enter();
    int comm_leave();
    int comm_fork();                               // second streams wait for the main streams
    int comm_mark(hipEvent_t Slab::*ev);           // record on the second streams
    int comm_wait(hipEvent_t Slab::*ev);           // main streams wait
    bool overlap = false;              // DOTSOCP_OVERLAP=0/1 overrides (default: on in time-slab mode)
    std::vector<Slab> slabs;        // the slabs held by THIS process
    // one host thread per slab issues that slab's launches while run() is active (defer.h); null: the caller's thread
    // issues everything (one slab, one process per GPU, DOTSOCP_HOST_THREADS=0)
    std::unique_ptr<DeferCtx> defer;
    int world = 1;                  // total number of slabs
    int rank = 0;                   // RCCL mode: this process's slab
    void *nccl = nullptr;           // ncclComm_t when a communicator is attached
    int open_groups = 0;            // ncclGroupStart calls not yet matched by ncclGroupEnd (comm.h: DS_NCCL_G)
    double *h_sums = nullptr;       // pinned host buffer [S_COUNT + 1] (RCCL mode: reduced sums + clock)
    double *d_red = nullptr;        // device buffer for the cross-rank reduction [S_COUNT + 1]

    // ---- loop state (mirrors solver_socp_inPALM.m:11-135) ----
    bool begun = false, finished = false, stopped = false;
    dotsocp_opts opts{};
    bool checkPrimDualFeas = true;
    double time_limit = 3600.0;
    double sigma = 1.0, sigmaScale = 1.0;
    double cScale = 1.0, dScale = 1.0, D = 1.0, E = 1.0;
    double norm_c = 0.0, norm_d = 0.0;
    double h = 1.0;
    int rescale = 0, use_feasOrg = 0;
    double maxFeas = 0.0, relGap = 0.0, tol_feasOrg = 0.0;
    double lastSigmaIt = 0.0;
    i64 it = 0;
    LoopCoef lc{};
    std::vector<double> hist_kkt, hist_time, hist_iter, hist_gap;   // kkt stored row-wise (7 per entry)
    std::chrono::steady_clock::time_point t_begin;
    double elapsed_prev = 0.0;
    double elapsed_agreed = 0.0;    // multi-process: max over ranks at the last KKT check

    // ---- profiling (HIP events on the launch stream) ----
    bool profiling = false;
    bool fused = true;       // DOTSOCP_FUSED=0 selects the unfused reference dataflow (z stored, 3 cone passes)
    bool deferred = false;   // fused path: beta still holds beta^{k-1}; z is not materialised
    bool z_valid = true;     // s.z holds the z of the last completed iteration
    bool z_prev_ok = false;  // s.beta2 / s.q_old still hold (beta^k, q^k): z can be regenerated (MODE_Z)
    int ensure_z();
    // fused path: a scaling of beta that the next pass over beta applies on load (saves a 20 Nz pass)
    int bpend = 0;           // pending operations on beta: 0, 1 or 2 (x * bmul / bdiv, then x * bmul2 / bdiv2)
    double bmul = 1.0, bdiv = 1.0, bmul2 = 1.0, bdiv2 = 1.0;
    int push_beta_op(double mul, double div);
    int zp_pend = 0;                 // the ops that were pending on the kept beta^k (beta2) when it was read
    double zp_mul = 1.0, zp_div = 1.0, zp_mul2 = 1.0, zp_div2 = 1.0;
    // KKT sums of the last check (Solver::kkt_block) and the sigma they were taken with: the rescale block of the
    // NEXT iteration finds its five norms there instead of making its own passes over the state
    double last_S[S_COUNT] = {0};
    double last_S_sigma = 1.0;
    i64 last_S_it = -1;
    int flush_beta();
    // same for alpha after a sigma update on the folded KKT path: the next q-step divides on load
    bool apend = false;
    double amul = 1.0, adiv = 1.0;
    int flush_alpha();
    int sigma_scale_folded(double factor);
    bool norm_cache = true;  // DOTSOCP_NORM_CACHE=0: the rescale block always makes its own passes for its norms
    bool kkt_fold = true;    // DOTSOCP_KKT_FOLD=0: KKT sums by the separate node / edge launches on every path
    KktCoef kkt_coef() const;
    void set_pending(FusedArgs &a) const {
        a.bpend = bpend; a.bmul = bmul; a.bdiv = bdiv; a.bmul2 = bmul2; a.bdiv2 = bdiv2;
    }
    struct Pending { hipEvent_t a, b; int phase; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    double phase_ms[PH_COUNT] = {0};
    i64 phase_launches[PH_COUNT] = {0};

    ~Solver();
    int init(const dotsocp_problem *p, int device, int nslabs, bool multi_dev = false);
    int attach_rccl(const unsigned char *id, int rank, int world);
    int upload(int field, const double *host);
    int upload_layers(int field, const double *host, i64 t0, i64 n);
    int download(int field, double *host);
    int begin(const dotsocp_opts *o);
    int begin_method(const dotsocp_opts *o, int method, const dotsocp_acc_opts *acc);
    int run(i64 n_iters, i64 *done);
    int finish(dotsocp_result *res);

    // internals
    int alloc_slabs(int first, int count);
    int ensure_alloc();
    void free_slabs();
    bool multi() const { return world > 1; }
    bool remote() const { return nccl != nullptr; }
    int step(bool *brk);
    int rescale_block();
    // hooks of the asynchronous schedule inside the partitioned t-solve: `fill` runs on the main streams while the first
    // interface exchange travels, `behind` right after the second one has been issued
    struct PhiHooks { std::function<int()> fill, behind; };
    int phase_phi(const PhiHooks *hooks = nullptr);
    // part 0: all chunks; 1: all but the last chunk; 2: the last chunk (the only one that reads the q halo)
    int phase_z(int part = 0);
    int phase_z_tails();
    int make_tails();        // time slabs: finalise the adjoint sums of the last owned cell for the right neighbour
    int send_tails();        // ... adjoint tails -> right
    int send_phi_head();     // ... first phi layer -> left
    int ship_tails();        // all three, one group
    // part 0: whole q-step; 1: all chunks but the last; 2: the last chunk (the one that reads the phi halo), then finish
    int phase_q(int part = 0, bool kkt = false);
    int phase_mult();
    int materialise();
    int kkt_sums(double *S, bool folded = false);
    int reduce_sums(double *S);
    int norms_light(double *S);
    int kkt_block(bool adjustSigmaYes, bool timed_out, bool *brk, bool folded = false);
    int scale_state(double a_mul, double a_div, double q_div, bool with_c);
    void update_coef();
    double elapsed() const;
    // HIP events on slab 0's main stream (on_z: its second stream); slab 0 stands for all (slabs run in lockstep)
    void prof_begin(int phase, bool on_z = false);
    void prof_end(int phase, bool on_z = false);
    int prof_flush();
    int poisson_all(const PhiHooks *hooks = nullptr);
    int transpose(bool forward);
    int exchange_q_halo(bool with_u0);
    int ensure_halo();       // run the q-halo exchange the last q-step left pending (halo_pending)
    int exchange_u0_tail();
    int make_u0_tail();
    int group_begin();
    int group_end();
    bool tri_tsolve = true;  // time-slab Poisson solve by partitioned tridiagonal systems (tri.hip); DOTSOCP_TSOLVE=dct:
                             // slab <-> pencil transposes around the t-axis DCT instead
    int tri_alloc();
    int tri_exchange(bool back);
    int poisson_t_tridiag(const PhiHooks *hooks);
    const bool qrhs = true;  // the fused dataflow's q-step also forms the next right-hand side (k_qstep_rhs)
    bool rhs_valid = false;  // w0 holds A'(w.*q - alpha) + c of the current iterate (left there by the q-step)
    // the q halo / u0 tail of the newest iterate have not been exchanged yet: step() issues the exchange behind the
    // fork so that the cone chunks that do not read the halo overlap it; every other reader calls ensure_halo()
    bool halo_pending = false;
    bool u0_fresh = false;   // u0_prev holds w.*q0 - alpha0 of the CURRENT iterate of the left neighbour
    bool u0_made = false;    // send_plane holds the u0 tail of the current iterate (written by the q-step itself)
    // every slab with a neighbour in direction `dir` (+1 right, -1 left) sends `count` doubles
    // from src(slab) to dst(neighbour)
    typedef std::function<double *(Slab &)> Sel;
    int shift(int dir, const Sel &src, const Sel &dst, i64 count);
    // slabs of one process: the copies of the shift() calls between group_begin() and group_end() (a lone shift() is a
    // group of one) are collected and pulled by ONE launch per receiving slab (flush_msgs)
    struct Msg { int from, to; const double *src; double *dst; i64 count; };
    std::vector<Msg> msgs;
    int msg_depth = 0;
    bool peer_ok = true;      // every pair of devices in use can address each other's memory (alloc_slabs)
    bool cross_device = false;   // some pair of this process's slabs lives on different devices (alloc_slabs)
    bool pull_default(const char *env_var) const;
    bool msg_batching() const;
    int flush_msgs();
    int shift_edge_halo(const Sel &base);      // first owned bx / by layers of base(s) -> halo layer of the left slab
    i64 field_len(int field) const;

    // ---- loop variants (include/dotsocp.h: DOTSOCP_METHOD_*) ----
    int method = DOTSOCP_METHOD_INPALM;
    // acc-ADMM (solver_socp_accADMM.m:12-34,157-163)
    i64 acc_restart = 100, acc_k = 0;
    double acc_rho = 2.0, acc_theta = 2.0;
    bool acc_halpern = true;
    bool acc_gather_valid = false;     // q2 / sx / sy hold F*B*(z + beta) of the current state
    bool acc_swapped = false;          // state and x^+ pointers are exchanged (during the KKT block / after a stop)
    bool acc_post = true;              // Halpern: after a KKT check z and beta take ONE pass (k_acc_cone modes 1 / 3) instead of
                                       // scalings, anchor copies, extrapolation passes and the gather pass (DOTSOCP_ACC_POST=0: off)
    bool acc_light = false;            // inside that iteration's KKT block: the sigma update leaves z, beta and their anchors alone
    double acc_factor = 1.0;           // factor of that block's sigma update (1: none)
    int acc_alloc();
    int acc_begin(const dotsocp_acc_opts *acc);
    int acc_step(bool *brk);
    int acc_rescale_block();
    int acc_set_anchors();
    void acc_swap_state();
    int acc_on_sigma_factor(double factor);
    AccCoef acc_coef() const;
    // driver steps on the device (solver_io.hip)
    int recover_outputs(const double *rho0, const double *rho1, double *rho, double *Ex, double *Ey, double *q0,
                        double *bx, double *by);
    int jump_from(Solver &coarse);
    // PALM (solver_palm.hip)
    bool palm_fast = true;     // one pass over beta per iteration (DOTSOCP_PALM_FAST=0: the two-pass dataflow)
    bool palm_p_valid = false; // p2 / sxp / syp hold the second gather of the last cone pass
    int palm_begin();
    int palm_step(bool *brk);
};

}  // namespace dotsocp
