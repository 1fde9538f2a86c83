/* Stateful-free solver-level gateway behind solver_socp_inPALM.m / solver_wsocp_inPALM.m:
 *
 *   out = dotsocp_inpalm_mex(S, opts)
 *
 * S    : struct with the fields of VarHandle / ModelHandle the loop reads
 *        (socp/dot2d/utils/VarHandle.m:3-17, ModelHandle.m:3-16):
 *        phi, q, alpha, z, beta, c, [weight], nx, [ny], nt, D, E, cScale, dScale, normc, normd
 * opts : the options struct of solver_socp_inPALM.m:20-37,64-68 (tau, sigma, maxit, tol,
 *        ifCheckStepByStep required; checkPrimDualFeas, time_limit, scaling optional; device, ngpu
 *        optional extensions).  opts.method (extension, set by the .m wrappers): 'inPALM' (default),
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

static mxArray *take(dotsocp_ctx *ctx, int field, const mxArray *like) {
    mxArray *o = mxCreateDoubleMatrix(mxGetM(like), mxGetN(like), mxREAL);
    DS_MEX_CHECK(dotsocp_download(ctx, field, mxGetPr(o)), ID);
    return o;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
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
    const int nslabs = (int)opt(O, "ngpu", 1);       /* > 1: time slabs in this process (one device) */

    dotsocp_ctx *ctx = dotsocp_create(&p, device, nslabs);
    if (!ctx) mexErrMsgIdAndTxt(ID, "%s", dotsocp_last_error());
    static const char *names[] = {"phi", "q", "alpha", "z", "beta", "c"};
    static const int fields[] = {DOTSOCP_F_PHI, DOTSOCP_F_Q, DOTSOCP_F_ALPHA, DOTSOCP_F_Z, DOTSOCP_F_BETA, DOTSOCP_F_C};
    int rc = 0;
    for (int i = 0; i < 6 && rc == 0; ++i) rc = dotsocp_upload(ctx, fields[i], ds_real(need(S, names[i]), ID, names[i]));
    if (rc == 0 && p.weighted) rc = dotsocp_upload(ctx, DOTSOCP_F_WEIGHT, ds_real(wf, ID, "weight"));
    dotsocp_result res;
    if (rc == 0) rc = dotsocp_begin_method(ctx, &o, method, &acc);
    if (rc == 0) rc = dotsocp_run(ctx, -1, NULL);
    if (rc == 0) rc = dotsocp_finish(ctx, &res);
    if (rc != 0) {
        dotsocp_destroy(ctx);                       /* device state is freed before the error is raised */
        mexErrMsgIdAndTxt(ID, "%s", dotsocp_last_error());
    }
    static const char *outf[] = {"phi", "q", "z", "alpha", "beta", "sigma", "cScale", "dScale", "times",
                                 "kkt", "time", "iter", "pdGap", "time_extra"};
    mxArray *out = mxCreateStructMatrix(1, 1, 14, outf);
    mxSetField(out, 0, "time_extra", mxCreateDoubleScalar(res.time_extra));
    mxSetField(out, 0, "phi", take(ctx, DOTSOCP_F_PHI, need(S, "phi")));
    mxSetField(out, 0, "q", take(ctx, DOTSOCP_F_Q, need(S, "q")));
    mxSetField(out, 0, "z", take(ctx, DOTSOCP_F_Z, need(S, "z")));
    mxSetField(out, 0, "alpha", take(ctx, DOTSOCP_F_ALPHA, need(S, "alpha")));
    mxSetField(out, 0, "beta", take(ctx, DOTSOCP_F_BETA, need(S, "beta")));
    mxSetField(out, 0, "sigma", mxCreateDoubleScalar(res.sigma));
    mxSetField(out, 0, "cScale", mxCreateDoubleScalar(res.cScale));
    mxSetField(out, 0, "dScale", mxCreateDoubleScalar(res.dScale));
    mxArray *tm = mxCreateDoubleMatrix(1, 7, mxREAL);
    memcpy(mxGetPr(tm), res.times, sizeof res.times);
    mxSetField(out, 0, "times", tm);
    const size_t n = (size_t)res.hist_len;
    mxArray *kkt = mxCreateDoubleMatrix(n, 7, mxREAL), *t = mxCreateDoubleMatrix(n, 1, mxREAL);
    mxArray *it = mxCreateDoubleMatrix(n, 1, mxREAL), *gap = mxCreateDoubleMatrix(n, 1, mxREAL);
    DS_MEX_CHECK(dotsocp_get_history(ctx, mxGetPr(kkt), mxGetPr(t), mxGetPr(it), mxGetPr(gap)), ID);
    mxSetField(out, 0, "kkt", kkt);
    mxSetField(out, 0, "time", t);
    mxSetField(out, 0, "iter", it);
    mxSetField(out, 0, "pdGap", gap);
    dotsocp_destroy(ctx);
    plhs[0] = out;
}
