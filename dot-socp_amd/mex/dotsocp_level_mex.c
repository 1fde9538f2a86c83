/* Persistent-handle gateway for the multilevel drivers (solver_dotsocp2d.m:154-250 and twins): one device
 * context per level, kept alive between calls, so that the state never leaves the GPU between levels
 * (jump_nextLevel.m:5-16 and recover_RhoE.m / recover_q.m run on the device).
 *
 *   h   = dotsocp_level_mex('create', S, opts)           S, opts as for dotsocp_inpalm_mex; phi, q, alpha, z, beta of S
 *                                                         are optional (absent: zeros -- the cold start of initialize.m)
 *   h   = dotsocp_level_mex('create', S, opts, hCoarse)   state produced from the finished level hCoarse by
 *                                                         dotsocp_jump_next_level; S carries c [, weight] and the scalars
 *   out = dotsocp_level_mex('solve', h)                   runs the loop; out: sigma, cScale, dScale, times, time_extra,
 *                                                         kkt, time, iter, pdGap                (:329-357)
 *   st  = dotsocp_level_mex('fields', h)                  phi, q, z, alpha (= sigma*alpha), beta (= sigma*beta)
 *   o   = dotsocp_level_mex('outputs', h, rho0, rho1)     rho, Ex, Ey, q0, bx, by (1-D: rho, Ex, q0, bx)   (:262-281)
 *         dotsocp_level_mex('destroy', h)
 *
 * Handles are small positive integers (doubles) into a table of this MEX file; 'destroy' releases the device memory
 * of one level, and clearing the MEX file (`clear mex`, MATLAB exit) releases every level still open (mexAtExit). */
#include <string.h>

#include "mex_common.h"

#define ID "dotsocp:level"
#define MAXH 64

typedef struct {
    dotsocp_ctx *ctx;
    dotsocp_problem p;
    dotsocp_opts o;
    dotsocp_acc_opts acc;
    int method;
    int solved;
} level_t;

static level_t g_levels[MAXH];
static int g_atexit = 0;

static void release_all(void) {           /* mexAtExit: `clear mex` must not strand full-grid HBM allocations */
    for (int i = 0; i < MAXH; ++i)
        if (g_levels[i].ctx) { dotsocp_destroy(g_levels[i].ctx); g_levels[i].ctx = NULL; }
    (void)dotsocp_release_cache();        /* ... nor the buffers the library keeps for the next context of their size */
}

/* a full real double array of exactly dotsocp_field_len(p, field) elements, or a MATLAB error -- checked before the
 * level's context exists, so there is nothing to release */
static const double *sized(const mxArray *a, const dotsocp_problem *p, int field, const char *name) {
    const double *pr = ds_real(a, ID, name);
    const dotsocp_i64 want = dotsocp_field_len(p, field);
    if (want < 0) mexErrMsgIdAndTxt(ID ":size", "grid %lld x %lld x %lld is not a valid problem", p->ny, p->nx, p->nt);
    if ((dotsocp_i64)mxGetNumberOfElements(a) != want)
        mexErrMsgIdAndTxt(ID ":size", "field '%s' has %lld elements, the %lld x %lld x %lld grid needs %lld", name,
                          (long long)mxGetNumberOfElements(a), p->ny, p->nx, p->nt, want);
    return pr;
}

static const mxArray *need(const mxArray *s, const char *f) {
    const mxArray *a = mxGetField(s, 0, f);
    if (!a) mexErrMsgIdAndTxt(ID, "missing field '%s'", f);
    return a;
}

static double opt(const mxArray *s, const char *f, double dflt) {
    const mxArray *a = mxGetField(s, 0, f);
    return (a && !mxIsEmpty(a)) ? mxGetScalar(a) : dflt;
}

static level_t *level(const mxArray *h) {
    const int i = (int)ds_scalar(h, ID, "handle");
    if (i < 1 || i > MAXH || !g_levels[i - 1].ctx) mexErrMsgIdAndTxt(ID, "invalid level handle %d", i);
    return &g_levels[i - 1];
}

static void fail(level_t *L) {            /* release the level, then raise the library's message */
    char msg[1024];
    strncpy(msg, dotsocp_last_error(), sizeof msg - 1);
    msg[sizeof msg - 1] = 0;
    if (L && L->ctx) { dotsocp_destroy(L->ctx); L->ctx = NULL; }
    mexErrMsgIdAndTxt(ID, "%s", msg);
}

static void cmd_create(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 3 || nrhs > 4 || !mxIsStruct(prhs[1]) || !mxIsStruct(prhs[2]) || nlhs > 1)
        mexErrMsgIdAndTxt(ID, "usage: h = dotsocp_level_mex('create', S, opts[, hCoarse])");
    const mxArray *S = prhs[1], *O = prhs[2];
    if (!g_atexit) { mexAtExit(release_all); g_atexit = 1; }
    int slot = -1;
    for (int i = 0; i < MAXH && slot < 0; ++i)
        if (!g_levels[i].ctx) slot = i;
    if (slot < 0) mexErrMsgIdAndTxt(ID, "too many open levels (destroy some)");
    level_t *L = &g_levels[slot];
    memset(L, 0, sizeof *L);
    const mxArray *nyf = mxGetField(S, 0, "ny");
    L->p.dim = (nyf && !mxIsEmpty(nyf)) ? 2 : 1;
    L->p.ny = L->p.dim == 2 ? (dotsocp_i64)mxGetScalar(nyf) : 1;
    L->p.nx = (dotsocp_i64)mxGetScalar(need(S, "nx"));
    L->p.nt = (dotsocp_i64)mxGetScalar(need(S, "nt"));
    const mxArray *wf = mxGetField(S, 0, "weight");
    L->p.weighted = (wf && !mxIsEmpty(wf)) ? 1 : 0;
    L->p.D = mxGetScalar(need(S, "D"));
    L->p.E = mxGetScalar(need(S, "E"));
    L->p.cScale = mxGetScalar(need(S, "cScale"));
    L->p.dScale = mxGetScalar(need(S, "dScale"));
    L->p.normc = mxGetScalar(need(S, "normc"));
    L->p.normd = opt(S, "normd", 0.0);
    L->method = DOTSOCP_METHOD_INPALM;
    const mxArray *mf = mxGetField(O, 0, "method");
    if (mf && !mxIsEmpty(mf)) {
        char name[16];
        if (mxGetString(mf, name, sizeof name) != 0) mexErrMsgIdAndTxt(ID, "opts.method must be a short char array");
        if (strcmp(name, "PALM") == 0) L->method = DOTSOCP_METHOD_PALM;
        else if (strcmp(name, "accADMM") == 0) L->method = DOTSOCP_METHOD_ACCADMM;
        else if (strcmp(name, "inPALM") != 0) mexErrMsgIdAndTxt(ID, "unknown opts.method '%s'", name);
    }
    L->acc.restart = (dotsocp_i64)opt(O, "restart", 0);
    L->acc.rho = opt(O, "rho", 0);
    L->acc.theta = opt(O, "theta", 0);
    L->o.tau = (L->method == DOTSOCP_METHOD_ACCADMM) ? 1.0 : mxGetScalar(need(O, "tau"));
    L->o.sigma = mxGetScalar(need(O, "sigma"));
    L->o.maxit = (dotsocp_i64)mxGetScalar(need(O, "maxit"));
    L->o.tol = mxGetScalar(need(O, "tol"));
    L->o.ifCheckStepByStep = mxGetScalar(need(O, "ifCheckStepByStep")) != 0;
    L->o.checkPrimDualFeas = mxGetField(O, 0, "checkPrimDualFeas") ? (opt(O, "checkPrimDualFeas", 1) != 0) : -1;
    L->o.scaling = opt(O, "scaling", 0) != 0;
    L->o.time_limit = opt(O, "time_limit", 3600);
    level_t *coarse = (nrhs == 4) ? level(prhs[3]) : NULL;
    /* all arrays are checked against the grid before the context exists */
    static const char *names[] = {"phi", "q", "alpha", "z", "beta"};
    static const int fields[] = {DOTSOCP_F_PHI, DOTSOCP_F_Q, DOTSOCP_F_ALPHA, DOTSOCP_F_Z, DOTSOCP_F_BETA};
    const double *st[5] = {NULL, NULL, NULL, NULL, NULL};
    const double *cvec = sized(need(S, "c"), &L->p, DOTSOCP_F_C, "c");
    const double *wvec = L->p.weighted ? sized(wf, &L->p, DOTSOCP_F_WEIGHT, "weight") : NULL;
    if (!coarse)
        for (int i = 0; i < 5; ++i) {
            const mxArray *a = mxGetField(S, 0, names[i]);
            if (a && !mxIsEmpty(a)) st[i] = sized(a, &L->p, fields[i], names[i]);
        }
    /* opts.ngpu: time slabs on that many devices of this process (at least two time nodes per slab) */
    int ngpu = (int)opt(O, "ngpu", 1);
    if (ngpu > L->p.nt / 2) ngpu = (int)(L->p.nt / 2);
    L->ctx = (ngpu > 1) ? dotsocp_create_multi(&L->p, (int)opt(O, "device", 0), ngpu)
                        : dotsocp_create(&L->p, (int)opt(O, "device", 0), 1);
    if (!L->ctx) mexErrMsgIdAndTxt(ID, "%s", dotsocp_last_error());
    if (dotsocp_set_profiling(L->ctx, 1) != 0) fail(L);        /* var.time columns (solver_socp_inPALM.m:339-341) */
    if (dotsocp_upload(L->ctx, DOTSOCP_F_C, cvec) != 0) fail(L);
    if (wvec && dotsocp_upload(L->ctx, DOTSOCP_F_WEIGHT, wvec) != 0) fail(L);
    if (coarse) {
        if (dotsocp_jump_next_level(coarse->ctx, L->ctx) != 0) fail(L);
    } else {
        for (int i = 0; i < 5; ++i)
            if (st[i] && dotsocp_upload(L->ctx, fields[i], st[i]) != 0) fail(L);
    }
    plhs[0] = mxCreateDoubleScalar((double)(slot + 1));
}

static void cmd_solve(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs != 2 || nlhs > 1) mexErrMsgIdAndTxt(ID, "usage: out = dotsocp_level_mex('solve', h)");
    level_t *L = level(prhs[1]);
    if (L->solved) mexErrMsgIdAndTxt(ID, "level already solved");
    dotsocp_result res;
    if (dotsocp_begin_method(L->ctx, &L->o, L->method, &L->acc) != 0) fail(L);
    if (dotsocp_run(L->ctx, -1, NULL) != 0) fail(L);
    if (dotsocp_finish(L->ctx, &res) != 0) fail(L);
    L->solved = 1;
    static const char *outf[] = {"sigma", "cScale", "dScale", "times", "time_extra", "kkt", "time", "iter", "pdGap"};
    mxArray *out = mxCreateStructMatrix(1, 1, 9, outf);
    mxSetField(out, 0, "sigma", mxCreateDoubleScalar(res.sigma));
    mxSetField(out, 0, "cScale", mxCreateDoubleScalar(res.cScale));
    mxSetField(out, 0, "dScale", mxCreateDoubleScalar(res.dScale));
    mxSetField(out, 0, "time_extra", mxCreateDoubleScalar(res.time_extra));
    mxArray *tm = mxCreateDoubleMatrix(1, 7, mxREAL);
    memcpy(mxGetPr(tm), res.times, sizeof res.times);
    mxSetField(out, 0, "times", tm);
    const size_t n = (size_t)res.hist_len;
    mxArray *kkt = mxCreateDoubleMatrix(n, 7, mxREAL), *t = mxCreateDoubleMatrix(n, 1, mxREAL);
    mxArray *it = mxCreateDoubleMatrix(n, 1, mxREAL), *gap = mxCreateDoubleMatrix(n, 1, mxREAL);
    if (dotsocp_get_history(L->ctx, mxGetPr(kkt), mxGetPr(t), mxGetPr(it), mxGetPr(gap)) != 0) fail(L);
    mxSetField(out, 0, "kkt", kkt);
    mxSetField(out, 0, "time", t);
    mxSetField(out, 0, "iter", it);
    mxSetField(out, 0, "pdGap", gap);
    plhs[0] = out;
}

static void cmd_fields(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs != 2 || nlhs > 1) mexErrMsgIdAndTxt(ID, "usage: st = dotsocp_level_mex('fields', h)");
    level_t *L = level(prhs[1]);
    const dotsocp_problem *p = &L->p;
    const size_t plane = (size_t)(p->ny * p->nx), nt = (size_t)p->nt;
    const size_t Nz = plane * (nt - 1);
    const size_t Nq = Nz + (size_t)(p->ny * (p->nx - 1)) * nt + (size_t)((p->ny - 1) * p->nx) * nt;
    const size_t K = p->dim == 2 ? 10 : 6;
    static const char *outf[] = {"phi", "q", "z", "alpha", "beta"};
    static const int fields[] = {DOTSOCP_F_PHI, DOTSOCP_F_Q, DOTSOCP_F_Z, DOTSOCP_F_ALPHA, DOTSOCP_F_BETA};
    const size_t m[] = {plane * nt, Nq, Nz, Nq, Nz}, n[] = {1, 1, K, 1, K};
    mxArray *out = mxCreateStructMatrix(1, 1, 5, outf);
    for (int i = 0; i < 5; ++i) {
        mxArray *a = mxCreateDoubleMatrix(m[i], n[i], mxREAL);
        if (dotsocp_download(L->ctx, fields[i], mxGetPr(a)) != 0) fail(L);
        mxSetField(out, 0, outf[i], a);
    }
    plhs[0] = out;
}

static void cmd_outputs(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs != 4 || nlhs > 1) mexErrMsgIdAndTxt(ID, "usage: o = dotsocp_level_mex('outputs', h, rho0, rho1)");
    level_t *L = level(prhs[1]);
    const dotsocp_problem *p = &L->p;
    const size_t plane = (size_t)(p->ny * p->nx), nt = (size_t)p->nt;
    if (mxGetNumberOfElements(prhs[2]) != plane || mxGetNumberOfElements(prhs[3]) != plane)
        mexErrMsgIdAndTxt(ID, "rho0 and rho1 must have ny*nx elements");
    const double *r0 = ds_real(prhs[2], ID, "rho0"), *r1 = ds_real(prhs[3], ID, "rho1");
    if (p->dim == 2) {
        static const char *outf[] = {"rho", "Ex", "Ey", "q0", "bx", "by"};
        mxArray *out = mxCreateStructMatrix(1, 1, 6, outf);
        mxArray *a[6];
        for (int i = 0; i < 6; ++i) {
            a[i] = mxCreateDoubleMatrix(plane, i < 3 ? nt : nt - 1, mxREAL);     /* (ny*nx) x nt: reshape on the MATLAB side */
            mxSetField(out, 0, outf[i], a[i]);
        }
        if (dotsocp_recover_outputs(L->ctx, r0, r1, mxGetPr(a[0]), mxGetPr(a[1]), mxGetPr(a[2]), mxGetPr(a[3]),
                                    mxGetPr(a[4]), mxGetPr(a[5])) != 0) fail(L);
        plhs[0] = out;
    } else {
        static const char *outf[] = {"rho", "Ex", "q0", "bx"};
        mxArray *out = mxCreateStructMatrix(1, 1, 4, outf);
        mxArray *a[4];
        for (int i = 0; i < 4; ++i) {
            a[i] = mxCreateDoubleMatrix(plane, i < 2 ? nt : nt - 1, mxREAL);
            mxSetField(out, 0, outf[i], a[i]);
        }
        if (dotsocp_recover_outputs(L->ctx, r0, r1, mxGetPr(a[0]), mxGetPr(a[1]), NULL, mxGetPr(a[2]), mxGetPr(a[3]),
                                    NULL) != 0) fail(L);
        plhs[0] = out;
    }
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[16];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd) != 0)
        mexErrMsgIdAndTxt(ID, "first argument: 'create' | 'solve' | 'fields' | 'outputs' | 'destroy'");
    if (strcmp(cmd, "create") == 0) cmd_create(nlhs, plhs, nrhs, prhs);
    else if (strcmp(cmd, "solve") == 0) cmd_solve(nlhs, plhs, nrhs, prhs);
    else if (strcmp(cmd, "fields") == 0) cmd_fields(nlhs, plhs, nrhs, prhs);
    else if (strcmp(cmd, "outputs") == 0) cmd_outputs(nlhs, plhs, nrhs, prhs);
    else if (strcmp(cmd, "destroy") == 0) {
        if (nrhs != 2) mexErrMsgIdAndTxt(ID, "usage: dotsocp_level_mex('destroy', h)");
        level_t *L = level(prhs[1]);
        dotsocp_destroy(L->ctx);
        L->ctx = NULL;
    } else {
        mexErrMsgIdAndTxt(ID, "unknown command '%s'", cmd);
    }
}
