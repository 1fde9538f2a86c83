/*
 * dotsocp.h -- C ABI of libdotsocp, the MI355X (gfx950) implementation of the
 * inPALM/ADMM SOCP iteration loop of chlhnu/DOT-SOCP.
 *
 * Two drop-in boundaries:
 *
 *   B2  operator level -- the five MEX operators the reference ships as binaries.
 *       Host pointers in MATLAB column-major layout; like the MEX files, the FIRST
 *       argument is overwritten in place.  A MEX gateway binds them one-to-one
 *       (dot-socp_amd/mex/).
 *
 *   B1  solver level -- [runHist, sigma] = solver_socp_inPALM(var, opts, model)
 *       (socp/dot2d/algorithms/solver_socp_inPALM.m:1, the dot1d twin, and
 *       socp/wdot2d/algorithms/solver_wsocp_inPALM.m:1).  The device owns the whole
 *       loop state between create() and destroy(); the caller uploads the fields of
 *       VarHandle / ModelHandle once, runs, and downloads the iterates.
 *
 * All functions return 0 on success and a negative DOTSOCP_E* code otherwise;
 * dotsocp_last_error() returns a thread-local description.  No exceptions cross the
 * ABI.  There is NO CPU fallback: without a HIP device every compute entry point
 * fails with DOTSOCP_ENODEVICE.
 *
 * Layout conventions (identical to the reference): grid ny x nx x nt, y fastest, then
 * x, then t;  Nphi = ny*nx*nt, Nz = ny*nx*(nt-1), Nbx = ny*(nx-1)*nt,
 * Nby = (ny-1)*nx*nt, Nq = Nz+Nbx+Nby;  q = [q0; bx; by];  z, beta = Nz x 10
 * column-major (1-D problems: grid nx x nt, q = [q0; bx], z = Nz x 6).
 */
#ifndef DOTSOCP_H
#define DOTSOCP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef long long dotsocp_i64;

enum {
    DOTSOCP_OK = 0,
    DOTSOCP_EINVAL = -1,     /* bad argument (sizes, NULL, unknown field)            */
    DOTSOCP_ENODEVICE = -2,  /* no HIP device / HIP runtime error at initialisation  */
    DOTSOCP_EHIP = -3,       /* a HIP call or kernel failed                          */
    DOTSOCP_ESTATE = -4,     /* call sequence violated (e.g. run() before begin())   */
    DOTSOCP_ECOMM = -5       /* RCCL / communicator failure                          */
};

const char *dotsocp_last_error(void);
const char *dotsocp_version(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int dotsocp_device_count(void);
/* Device buffers of destroyed contexts are kept (per device) and reused by the next context that asks for buffers of
 * their size: on this platform a hipMalloc that follows the release of tens of GB takes seconds.  This call returns
 * them to the driver (bytes released); DOTSOCP_DEVICE_CACHE=0 in the environment disables the cache altogether. */
dotsocp_i64 dotsocp_release_cache(void);

/* ===================================================================================
 * B2 -- operator level, HOST pointers (replaces the reference MEX binaries)
 * =================================================================================== */

/* mexProjSoc(out, in)  -- socp/{dot1d,dot2d,wdot2d}/utils/mexProjSoc.mexa64;
 * call sites socp/dot2d/algorithms/solver_socp_inPALM.m:199,240.
 * Row-wise projection of the M x K column-major matrix `in` onto {x1 >= ||x_2..K||}. */
int dotsocp_proj_soc(double *out, const double *in, dotsocp_i64 M, dotsocp_i64 K);

/* mexBFd(z, q, nt, nx, ny, scale, dF)  -- socp/dot2d/utils/mexBFd.mexa64;
 * call sites solver_socp_inPALM.m:133,187,212,242.   z <- B F q + d  (Nz x 10).
 * Slots whose edge lies outside the domain are not written (keep the caller's value). */
int dotsocp_bfd(double *z, const double *q, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny,
                double scale, double dF);

/* mexBFdConj(q, z, nt, nx, ny, scale)  -- socp/dot2d/utils/mexBFdConj.mexa64;
 * call sites solver_socp_inPALM.m:205,225; socp/dot2d/utils/jump_nextLevel.m:16.
 * q <- F* B* z  (Nq). */
int dotsocp_bfd_conj(double *q, const double *z, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny,
                     double scale);

/* mexBFd1d(z, q, nt, nx, scale, dF)  -- socp/dot1d/utils/mexBFd1d.mexa64;
 * call sites socp/dot1d/algorithms/solver_socp_inPALM.m:132,186,211,241.  z is Nz x 6. */
int dotsocp_bfd1d(double *z, const double *q, dotsocp_i64 nt, dotsocp_i64 nx, double scale, double dF);

/* mexBFdConj1d(q, z, nt, nx, scale)  -- socp/dot1d/utils/mexBFdConj1d.mexa64;
 * call sites socp/dot1d/algorithms/solver_socp_inPALM.m:204,224. */
int dotsocp_bfd_conj1d(double *q, const double *z, dotsocp_i64 nt, dotsocp_i64 nx, double scale);

/* res = oper_poisson3dim(kernelScale * initialize_FFTkernel(nt,nx,ny), rhs)
 * -- socp/dot2d/utils/oper_poisson3dim.m:4, initialize_FFTkernel.m:6-15,
 *    mirt_dctn.m / mirt_idctn.m (orthonormal DCT-II / DCT-III along every axis);
 *    call site solver_socp_inPALM.m:96,194.   For a 1-D problem pass ny = nx1d, nx = 1
 *    (socp/dot1d/utils/oper_poisson.m:4).  rhs and res hold ny*nx*nt doubles. */
int dotsocp_oper_poisson(double *res, const double *rhs, dotsocp_i64 ny, dotsocp_i64 nx,
                         dotsocp_i64 nt, double kernelScale);

/* a <- dctn(a) (inverse = 0) or idctn(a) (inverse = 1) of an ny x nx x nt array
 * -- socp/dot2d/utils/mirt_dctn.m:64-141, mirt_idctn.m:59-128. */
int dotsocp_dctn(double *a, dotsocp_i64 ny, dotsocp_i64 nx, dotsocp_i64 nt, int inverse);

/* Same operators on DEVICE pointers (no PCIe traffic), enqueued on `stream`
 * (a hipStream_t, NULL = default stream).  Used by the parity tests and by bench.py. */
int dotsocp_proj_soc_dev(double *d_out, const double *d_in, dotsocp_i64 M, dotsocp_i64 K, void *stream);
int dotsocp_bfd_dev(double *d_z, const double *d_q, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny,
                    double scale, double dF, void *stream);
int dotsocp_bfd_conj_dev(double *d_q, const double *d_z, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny,
                         double scale, void *stream);

/* ===================================================================================
 * B1 -- solver level (replaces solver_socp_inPALM.m / solver_wsocp_inPALM.m)
 * =================================================================================== */

typedef struct dotsocp_ctx dotsocp_ctx;

/* Discrete model + scaling state on entry: the scalar fields of VarHandle / ModelHandle
 * (socp/dot2d/utils/VarHandle.m:3-17, ModelHandle.m:3-16) after InitialScaling
 * (socp/dot2d/solver_dotsocp2d.m:304-365). */
typedef struct {
    int dim;               /* 2: grid ny x nx x nt; 1: grid nx x nt (ny ignored)            */
    int weighted;          /* 1: solver_wsocp_inPALM semantics, model.weight uploaded       */
    dotsocp_i64 ny, nx, nt;
    double D, E;           /* var.D, var.E                                                   */
    double cScale, dScale; /* var.cScale, var.dScale                                         */
    double normc, normd;   /* model.normc, model.normd                                       */
} dotsocp_problem;

/* opts struct read by the loop (solver_socp_inPALM.m:20-37,64-68) */
typedef struct {
    double tau;                 /* 1.9 inPALM, 1.0 ALG2 (solver_dotsocp2d.m:100-101,133-137) */
    double sigma;               /* initial penalty                                          */
    double tol;
    dotsocp_i64 maxit;
    int ifCheckStepByStep;
    int checkPrimDualFeas;      /* -1 = reference default (true; weighted: false)           */
    int scaling;                /* enables the in-loop rescale block (:64-77,138-190)        */
    double time_limit;          /* seconds; <= 0 means the reference default 3600           */
} dotsocp_opts;

/* Loop variants behind the same boundary: which reference solver file the loop follows. */
enum {
    DOTSOCP_METHOD_INPALM = 0,  /* solver_socp_inPALM.m / solver_wsocp_inPALM.m (ALG2: tau = 1)     */
    DOTSOCP_METHOD_PALM = 1,    /* solver_socp_PALM.m (2-D unweighted only)                          */
    DOTSOCP_METHOD_ACCADMM = 2  /* solver_socp_accADMM.m / solver_wsocp_accADMM.m (2-D only)         */
};

/* extra options of the accelerated ADMM (solver_socp_accADMM.m:12-34); <= 0 selects the reference default */
typedef struct {
    dotsocp_i64 restart;        /* opts.restart, default 100                                         */
    double rho;                 /* opts.rho, default 2                                               */
    double theta;               /* opts.theta, default 2 (== 2: Halpern iteration)                   */
} dotsocp_acc_opts;

/* Field selectors for upload / download */
enum {
    DOTSOCP_F_PHI = 0,   /* Nphi                                  var.phi        */
    DOTSOCP_F_Q = 1,     /* Nq                                    var.q          */
    DOTSOCP_F_ALPHA = 2, /* Nq                                    var.alpha      */
    DOTSOCP_F_Z = 3,     /* Nz x 10 (1-D: Nz x 6)                 var.z          */
    DOTSOCP_F_BETA = 4,  /* Nz x 10 (1-D: Nz x 6)                 var.beta       */
    DOTSOCP_F_C = 5,     /* Nphi                                  model.c        */
    DOTSOCP_F_WEIGHT = 6 /* Nq (weighted only)                    model.weight   */
};

/* Loop outputs (solver_socp_inPALM.m:329-357) */
typedef struct {
    double sigma;          /* returned `sigma` = sigma / sigmaScale (:357)                  */
    double sigma_internal; /* sigma used to un-scale alpha,beta on download (:335-336)      */
    double cScale, dScale; /* var.cScale / var.dScale after in-loop rescales (:344-345)     */
    double times[7];       /* Step_1_1_FFT, Step_1_2_ProjSOC, Step_2_Q_Step,
                              Step_3_Multiplier, KKT, Total_Time, Iters (:339-341)          */
    dotsocp_i64 iters;     /* iterations executed                                           */
    dotsocp_i64 hist_len;  /* runHist.len                                                   */
    int stopped;           /* 1 when the stop criterion (:287-290) fired                    */
    double time_extra;     /* acc-ADMM: 'Interp' (solver_socp_accADMM.m:438-439); PALM: 'Step_1_Q_Step' */
} dotsocp_result;

/* `device` = HIP device ordinal.  `nslabs` >= 1 splits the time axis into that many slabs that live in this one
 * process ON THAT ONE DEVICE (the multi-GPU algorithm -- halo exchange and the t-axis coupling of the Poisson
 * solve -- with the slabs sharing the device's pair of streams, one after the other, and pull launches as messages): a
 * rehearsal / diagnostic mode.  Multi-GPU runs use either
 *   - dotsocp_create_multi(): ONE process, slab r on device (first_device + r) mod #visible devices, neighbour layers
 *     and the interface values of the t-solve travel as peer copies (hipMemcpyPeerAsync over xGMI), per-device KKT
 *     partial sums are added up on the host.  This is what a single MATLAB process (solver_dotsocp2d.m:208 calls
 *     the loop synchronously from the interpreter thread) uses: opts.ngpu of the MEX gateway.  With fewer devices
 *     than slabs, slabs share devices (every slab keeps its OWN pair of streams, also on one device: that is what the
 *     tests of the cross-slab ordering run).  upload / download /
 *     recover_outputs take and return the GLOBAL fields, exactly as with one slab.  STATUS: every loop is verified with
 *     2 .. 8 slabs on concurrent streams of ONE device (the build's boxes have one GPU); slabs on DIFFERENT devices
 *     (hipDeviceEnablePeerAccess, hipMemcpyPeerAsync, cross-device stream waits) have never run on hardware.  Between
 *     different devices messages travel as event-ordered peer copies by default; the launches that pull them through
 *     peer pointers are opt-in there (DOTSOCP_MSG_BATCH=1, DOTSOCP_TRI_GATHER=1), and so is one issuing host thread per
 *     slab (DOTSOCP_HOST_THREADS=1); or
 *   - one process per GPU: dotsocp_create(prob, device, 1) + dotsocp_attach_rccl() (bench.py, torch.distributed). */
dotsocp_ctx *dotsocp_create(const dotsocp_problem *prob, int device, int nslabs);
dotsocp_ctx *dotsocp_create_multi(const dotsocp_problem *prob, int first_device, int ngpu);
void dotsocp_destroy(dotsocp_ctx *ctx);

/* One process per GPU: this process owns time slab `rank` of `world`.  `unique_id` is the
 * 128-byte ncclUniqueId produced by dotsocp_rccl_unique_id() on rank 0 and broadcast by
 * the host (torch.distributed / MPI / MATLAB parallel pool).  Must be called right after
 * create() (before any upload).  With a communicator attached, upload/download take and
 * return the LOCAL slab of each field (dotsocp_slab_range()). */
int dotsocp_rccl_unique_id(unsigned char id[128]);
int dotsocp_attach_rccl(dotsocp_ctx *ctx, const unsigned char id[128], int rank, int world);

/* Time-slab partition used by the multi-GPU mode (pure host arithmetic, no device):
 * nodes [*t0, *t1) of the nt time nodes belong to slab `rank`; its staggered cells are
 * [*t0, min(*t1, nt-1)). */
int dotsocp_slab_range(dotsocp_i64 nt, int world, int rank, dotsocp_i64 *t0, dotsocp_i64 *t1);

/* Number of doubles of the GLOBAL field `field` (DOTSOCP_F_*) of problem `prob` in the reference layout -- what
 * upload() reads and download() writes for a context without an RCCL communicator.  Pure host arithmetic (no
 * device); -1 for an unknown field or a grid with nt < 2.  The MEX gateways check their mxArrays against it
 * before any pointer reaches the library. */
dotsocp_i64 dotsocp_field_len(const dotsocp_problem *prob, int field);

int dotsocp_upload(dotsocp_ctx *ctx, int field, const double *host);
int dotsocp_download(dotsocp_ctx *ctx, int field, double *host);
/* Extension for drivers: time layers [t0, t0 + n) of a NODE field (DOTSOCP_F_PHI, DOTSOCP_F_C) from a host buffer that
 * holds just those n layers of ny*nx doubles (with an RCCL communicator: layers of this process's slab); the other layers
 * keep their contents (zeros after create).  model.c of socp/dot2d/utils/initialize.m:42-50 is zero except for its first
 * and last layer, so a driver hands over two layers instead of a grid-sized vector. */
int dotsocp_upload_layers(dotsocp_ctx *ctx, int field, const double *host, dotsocp_i64 t0, dotsocp_i64 n);

/* solver_socp_inPALM.m:11-135 (setup), :136-325 (loop; `n_iters` < 0 = until maxit /
 * stop), :329-357 (outputs).  run() may be called repeatedly; the trajectory is identical
 * to a single call.  After finish(), download PHI/Q/Z/ALPHA/BETA gives var.* of :332-336
 * (alpha and beta multiplied by sigma). */
int dotsocp_begin(dotsocp_ctx *ctx, const dotsocp_opts *opts);
/* Same as dotsocp_begin for another loop file of the reference: DOTSOCP_METHOD_*; `acc` may be NULL
 * (reference defaults) and is read for DOTSOCP_METHOD_ACCADMM only.  run / finish / downloads are
 * unchanged; result.times follows the inPALM order with the variant's extra column in time_extra.
 * All three loops run on time slabs (nslabs > 1 or dotsocp_attach_rccl). */
int dotsocp_begin_method(dotsocp_ctx *ctx, const dotsocp_opts *opts, int method, const dotsocp_acc_opts *acc);
int dotsocp_run(dotsocp_ctx *ctx, dotsocp_i64 n_iters, dotsocp_i64 *done);
int dotsocp_finish(dotsocp_ctx *ctx, dotsocp_result *res);

/* Driver outputs on the device: recoverOrgVar (socp/dot2d/solver_dotsocp2d.m:368-386) +
 * recover_RhoE (utils/recover_RhoE.m:14-25; wdot2d: alpha = weight .* alpha, :11) + recover_q
 * (utils/recover_q.m:12-22) from the device-resident alpha and q, so that only the outputs cross PCIe.  Call
 * after finish().  rho0, rho1: model.rho0 / rho1 (ny x nx, host).  Outputs (host, column-major, any may be NULL):
 * rho, Ex, Ey: ny x nx x nt;  q0, bx, by: ny x nx x (nt-1).  1-D problems: rho, Ex: nx x nt; q0, bx: nx x (nt-1);
 * Ey, by ignored.  Works on time slabs too: in-process slabs (nslabs / dotsocp_create_multi) fill the global arrays;
 * with an RCCL communicator attached every rank gets its own slab of them (node layers [t0, t1) of rho, Ex, Ey,
 * cell layers of q0, bx, by; rho0 / rho1 are read on the first / last rank only). */
int dotsocp_recover_outputs(dotsocp_ctx *ctx, const double *rho0, const double *rho1, double *rho, double *Ex,
                            double *Ey, double *q0, double *bx, double *by);

/* Multilevel transfer on the device: socp/dot2d/utils/jump_nextLevel.m:5-16 with
 * interpolate.m:20-84, between the finished context `coarse` and the context `fine` of the next level (created
 * from the fine level's InitialScaling scalars, c and weight uploaded, begin() not yet called; grid
 * 2 (n - 1) + 1 per dimension).  Fills phi, q, alpha, z, beta of `fine` exactly as recoverOrgVar ->
 * jump_nextLevel -> InitialScaling -> upload would.  Either context may be cut into in-process time slabs
 * (dotsocp_create / dotsocp_create_multi, any two slab counts and placements): a fine slab gathers the coarse
 * layers it interpolates from out of the coarse slabs that own them (peer copies).  Contexts with an RCCL
 * communicator attached are refused (DOTSOCP_EINVAL): their levels exchange state through the host. */
int dotsocp_jump_next_level(dotsocp_ctx *coarse, dotsocp_ctx *fine);

/* runHist.{kkt (len x 7, column-major), time, iter, pdGap} (:350-354); any pointer may be NULL */
int dotsocp_get_history(dotsocp_ctx *ctx, double *kkt, double *time, double *iter, double *pdGap);

/* Per-step device times (HIP events on the launch stream): feeds result.times[0..4] (the reference's tic/toc
 * columns Step_1_1_FFT .. KKT, solver_socp_inPALM.m:339-341) and dotsocp_kernel_time().  Off by default; may be
 * switched at any time outside run() -- before begin() to cover the whole solve (the MEX gateways do that), or
 * between two run() calls to cover only what follows (bench.py switches it on after its warm-up iterations).
 * dotsocp_kernel_time: average device time in ms of the named kernel family over the profiled launches since
 * begin(), and their count; names: "rhs", "poisson", "cone_proj", "qstep", "beta", "kkt", "cone_fused_a",
 * "cone_fused_b", "materialise", "comm", "interp", "acc_cone", "acc_gather", "qstep_first", "transpose". */
int dotsocp_set_profiling(dotsocp_ctx *ctx, int on);
int dotsocp_kernel_time(dotsocp_ctx *ctx, const char *name, double *avg_ms, dotsocp_i64 *launches);

/* Guard bands (debugging aid): with DOTSOCP_CANARY=1 in the environment every device buffer of
 * the solver state is allocated between two bands of NaN-pattern words.  dotsocp_finish() fails with DOTSOCP_EHIP if a
 * band was overwritten (message: which buffer, how many words, where), dotsocp_destroy() reports to stderr, and this
 * call checks all live buffers of the process at any time: returns the number of damaged buffers (0 = clean, also
 * when the canaries are off) and sets dotsocp_last_error() accordingly.  An out-of-bounds READ shows up as NaNs. */
int dotsocp_canary_check(void);

/* Blocks until all work enqueued by this context has completed. */
int dotsocp_synchronize(dotsocp_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* DOTSOCP_H */
