/* Stateful-free solver-level gateway behind solver_socp_inPALM.m / solver_wsocp_inPALM.m:
 *
 *   out = dotsocp_inpalm_mex(S, opts)
 *
 * S    : struct with the fields of VarHandle / ModelHandle the loop reads
 *        (socp/dot2d/utils/VarHandle.m:3-17, ModelHandle.m:3-16):
 *        phi, q, alpha, z, beta, c, [weight], nx, [ny], nt, D, E, cScale, dScale, normc, normd
 * opts : the options struct of solver_socp_inPALM.m:20-37,64-68 (tau, sigma, maxit, tol,
 *        ifCheckStepByStep required; checkPrimDualFeas, time_limit, scaling optional; device, ngpu
 *        optional extensions: first device ordinal and number of GPUs, one time slab per GPU).  opts.method (extension, set by the .m wrappers): 'inPALM' (default),
 *        'PALM' (solver_socp_PALM.m) or 'accADMM' (solver_socp_accADMM.m / solver_wsocp_accADMM.m,
 *        which also reads restart, rho, theta, :12-28, and no tau)
 * out  : struct phi, q, z, alpha (= sigma*alpha), beta (= sigma*beta), sigma, cScale, dScale,
 *        times (1x7), time_extra, kkt (len x 7), time, iter, pdGap (len x 1)      (:329-357)
 *
 * The .m wrapper copies `out` back into the handle objects, so demo_dot*.m run unmodified. */
#include <string.h>

#include "mex_common.h"

#define ID "dotsocp:inPALM"

static const mxArray *need(const mxArray *s, const char *f) {
    const mxArray *a = mxGetField(s, 0, f);
    if (!a) mexErrMsgIdAndTxt(ID, "missing field '%s'", f);
    return a;
}

static double opt(const mxArray *s, const char *f, double dflt) {
    const mxArray *a = mxGetField(s, 0, f);
    return (a && !mxIsEmpty(a)) ? mxGetScalar(a) : dflt;
}

/* a full real double array of exactly dotsocp_field_len(p, field) elements, or a MATLAB error (raised before any
 * pointer reaches the library: upload() reads that many doubles from it) */
static const double *sized(const mxArray *a, const dotsocp_problem *p, int field, const char *name) {
    const double *pr = ds_real(a, ID, name);
    const dotsocp_i64 want = dotsocp_field_len(p, field);
    if (want < 0) mexErrMsgIdAndTxt(ID ":size", "grid %lld x %lld x %lld is not a valid problem", p->ny, p->nx, p->nt);
    if ((dotsocp_i64)mxGetNumberOfElements(a) != want)
        mexErrMsgIdAndTxt(ID ":size", "field '%s' has %lld elements, the %lld x %lld x %lld grid needs %lld", name,
                          (long long)mxGetNumberOfElements(a), p->ny, p->nx, p->nt, want);
    return pr;
}

/* The library keeps the device buffers of a finished call for the next call on a grid of that size (on this platform a
 * hipMalloc that follows the release of tens of GB takes seconds); `clear mex` / MATLAB exit hands them back. */
static void release_cache_at_exit(void) { (void)dotsocp_release_cache(); }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    static int at_exit_set = 0;
    if (!at_exit_set) { mexAtExit(release_cache_at_exit); at_exit_set = 1; }
    if (nrhs != 2 || !mxIsStruct(prhs[0]) || !mxIsStruct(prhs[1]))
        mexErrMsgIdAndTxt(ID, "usage: out = dotsocp_inpalm_mex(S, opts)");
    if (nlhs > 1) mexErrMsgIdAndTxt(ID, "one output");
    const mxArray *S = prhs[0], *O = prhs[1];
    dotsocp_problem p;
    memset(&p, 0, sizeof p);
    const mxArray *nyf = mxGetField(S, 0, "ny");
    p.dim = (nyf && !mxIsEmpty(nyf)) ? 2 : 1;
    p.ny = p.dim == 2 ? (dotsocp_i64)mxGetScalar(nyf) : 1;
    p.nx = (dotsocp_i64)mxGetScalar(need(S, "nx"));
    p.nt = (dotsocp_i64)mxGetScalar(need(S, "nt"));
    const mxArray *wf = mxGetField(S, 0, "weight");
    p.weighted = (wf && !mxIsEmpty(wf)) ? 1 : 0;
    p.D = mxGetScalar(need(S, "D"));
    p.E = mxGetScalar(need(S, "E"));
    p.cScale = mxGetScalar(need(S, "cScale"));
    p.dScale = mxGetScalar(need(S, "dScale"));
    p.normc = mxGetScalar(need(S, "normc"));
    p.normd = opt(S, "normd", 0.0);
    dotsocp_opts o;
    memset(&o, 0, sizeof o);
    int method = DOTSOCP_METHOD_INPALM;
    const mxArray *mf = mxGetField(O, 0, "method");
    if (mf && !mxIsEmpty(mf)) {
        char name[16];
        if (mxGetString(mf, name, sizeof name) != 0) mexErrMsgIdAndTxt(ID, "opts.method must be a short char array");
        if (strcmp(name, "PALM") == 0) method = DOTSOCP_METHOD_PALM;
        else if (strcmp(name, "accADMM") == 0) method = DOTSOCP_METHOD_ACCADMM;
        else if (strcmp(name, "inPALM") != 0) mexErrMsgIdAndTxt(ID, "unknown opts.method '%s'", name);
    }
    dotsocp_acc_opts acc;
    acc.restart = (dotsocp_i64)opt(O, "restart", 0);      /* 0: reference defaults (solver_socp_accADMM.m:12-28) */
    acc.rho = opt(O, "rho", 0);
    acc.theta = opt(O, "theta", 0);
    o.tau = (method == DOTSOCP_METHOD_ACCADMM) ? 1.0 : mxGetScalar(need(O, "tau"));
    o.sigma = mxGetScalar(need(O, "sigma"));
    o.maxit = (dotsocp_i64)mxGetScalar(need(O, "maxit"));
    o.tol = mxGetScalar(need(O, "tol"));
    o.ifCheckStepByStep = mxGetScalar(need(O, "ifCheckStepByStep")) != 0;
    o.checkPrimDualFeas = mxGetField(O, 0, "checkPrimDualFeas") ? (opt(O, "checkPrimDualFeas", 1) != 0) : -1;
    o.scaling = opt(O, "scaling", 0) != 0;
    o.time_limit = opt(O, "time_limit", 3600);
    const int device = (int)opt(O, "device", 0);
    /* opts.ngpu > 1: the time axis is cut into that many slabs, slab r on device (device + r) mod #devices of THIS
     * process (dotsocp_create_multi: streams per slab, peer copies between neighbours) -- MATLAB stays one process */
    int ngpu = (int)opt(O, "ngpu", 1);

    /* every array is checked against the grid BEFORE the context exists (nothing to release on these errors) */
    static const char *names[] = {"phi", "q", "alpha", "z", "beta", "c"};
    static const int fields[] = {DOTSOCP_F_PHI, DOTSOCP_F_Q, DOTSOCP_F_ALPHA, DOTSOCP_F_Z, DOTSOCP_F_BETA, DOTSOCP_F_C};
    const double *in[7];
    for (int i = 0; i < 6; ++i) in[i] = sized(need(S, names[i]), &p, fields[i], names[i]);
    in[6] = p.weighted ? sized(wf, &p, DOTSOCP_F_WEIGHT, "weight") : NULL;
    if (ngpu < 1) mexErrMsgIdAndTxt(ID, "opts.ngpu must be >= 1");
    /* the multilevel driver hands the same opts to every level (solver_dotsocp2d.m:208): a coarse level with fewer than
     * 2 * ngpu time nodes runs on fewer slabs (at least two time nodes per slab), like dotsocp_level_mex does */
    if (ngpu > p.nt / 2) ngpu = (int)(p.nt / 2);
    if (ngpu < 1) ngpu = 1;

    dotsocp_ctx *ctx = dotsocp_create_multi(&p, device, ngpu);
    if (!ctx) mexErrMsgIdAndTxt(ID, "%s", dotsocp_last_error());
    /* outputs are sized from the grid, not from the inputs' dims (download() writes field_len doubles) */
    const size_t K = p.dim == 2 ? 10 : 6;
    const size_t Nphi = (size_t)dotsocp_field_len(&p, DOTSOCP_F_PHI), Nq = (size_t)dotsocp_field_len(&p, DOTSOCP_F_Q);
    const size_t Nz = (size_t)dotsocp_field_len(&p, DOTSOCP_F_Z) / K;
    static const char *outf[] = {"phi", "q", "z", "alpha", "beta", "sigma", "cScale", "dScale", "times",
                                 "kkt", "time", "iter", "pdGap", "time_extra"};
    static const int ofield[] = {DOTSOCP_F_PHI, DOTSOCP_F_Q, DOTSOCP_F_Z, DOTSOCP_F_ALPHA, DOTSOCP_F_BETA};
    const size_t om[] = {Nphi, Nq, Nz, Nq, Nz}, on[] = {1, 1, K, 1, K};
    mxArray *out = mxCreateStructMatrix(1, 1, 14, outf);
    dotsocp_result res;
    memset(&res, 0, sizeof res);
    /* var.time carries the reference's tic/toc columns (solver_socp_inPALM.m:339-341): per-step device times on */
    int rc = dotsocp_set_profiling(ctx, 1);
    /* the inPALM / ALG2 loop overwrites z before its first use (solver_socp_inPALM.m:199; the rescale block, the only other
     * reader, needs it >= 10): with at least one iteration to run, S.z -- 10 of the 27 N doubles of the state -- stays home */
    const int z_unread = method == DOTSOCP_METHOD_INPALM && o.maxit >= 1;
    for (int i = 0; i < 6 && rc == 0; ++i)
        if (!(z_unread && fields[i] == DOTSOCP_F_Z)) rc = dotsocp_upload(ctx, fields[i], in[i]);
    if (rc == 0 && p.weighted) rc = dotsocp_upload(ctx, DOTSOCP_F_WEIGHT, in[6]);
    if (rc == 0) rc = dotsocp_begin_method(ctx, &o, method, &acc);
    if (rc == 0) rc = dotsocp_run(ctx, -1, NULL);
    if (rc == 0) rc = dotsocp_finish(ctx, &res);
    for (int i = 0; i < 5 && rc == 0; ++i) {
        mxArray *a = mxCreateDoubleMatrix(om[i], on[i], mxREAL);
        mxSetField(out, 0, outf[i], a);
        rc = dotsocp_download(ctx, ofield[i], mxGetPr(a));
    }
    mxArray *kkt = NULL, *t = NULL, *it = NULL, *gap = NULL;
    if (rc == 0) {
        const size_t n = (size_t)res.hist_len;
        kkt = mxCreateDoubleMatrix(n, 7, mxREAL); t = mxCreateDoubleMatrix(n, 1, mxREAL);
        it = mxCreateDoubleMatrix(n, 1, mxREAL); gap = mxCreateDoubleMatrix(n, 1, mxREAL);
        rc = dotsocp_get_history(ctx, mxGetPr(kkt), mxGetPr(t), mxGetPr(it), mxGetPr(gap));
    }
    if (rc != 0) {
        /* device state is freed before the error is raised (mexErrMsgIdAndTxt does not return; the mxArrays made
         * so far belong to MATLAB's memory manager and are reclaimed by it) */
        char msg[1024];
        strncpy(msg, dotsocp_last_error(), sizeof msg - 1);
        msg[sizeof msg - 1] = 0;
        dotsocp_destroy(ctx);
        mexErrMsgIdAndTxt(ID, "%s", msg);
    }
    dotsocp_destroy(ctx);
    mxSetField(out, 0, "time_extra", mxCreateDoubleScalar(res.time_extra));
    mxSetField(out, 0, "sigma", mxCreateDoubleScalar(res.sigma));
    mxSetField(out, 0, "cScale", mxCreateDoubleScalar(res.cScale));
    mxSetField(out, 0, "dScale", mxCreateDoubleScalar(res.dScale));
    mxArray *tm = mxCreateDoubleMatrix(1, 7, mxREAL);
    memcpy(mxGetPr(tm), res.times, sizeof res.times);
    mxSetField(out, 0, "times", tm);
    mxSetField(out, 0, "kkt", kkt);
    mxSetField(out, 0, "time", t);
    mxSetField(out, 0, "iter", it);
    mxSetField(out, 0, "pdGap", gap);
    plhs[0] = out;
}
