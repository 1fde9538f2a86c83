/* TEST INFRASTRUCTURE -- runs the CPU-side C of this repository under AddressSanitizer + UBSan (SURVEY.md section 5:
 * "ASAN/UBSAN build of the CPU side"; GPU ASAN is not available on the pool):
 *   - oracle/mex_kernels.c (the C part of the oracle): all five operators on small grids incl. degenerate ones,
 *     every output element read back, adjoint identity checked;
 *   - the seven MEX gateways of dot-socp_amd/mex/ through tests/fake_mx: every argument-error path (wrong counts,
 *     non-scalars, size mismatches, missing / short fields, unknown method, stale handles, exit handler).  On a box
 *     without a GPU the calls that reach the library come back as "<id>:device" / "no HIP device" errors -- also a
 *     path worth running under the sanitizers; nothing here needs a device.
 * tests/test_sanitizers.py builds this file with -fsanitize=address,undefined together with the sources above
 * (each gateway with -DmexFunction=mexFunction_<name>) and runs it; any sanitizer report or failed check makes the
 * process exit non-zero. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

typedef long long i64;
void oracle_proj_soc(double *out, const double *in, i64 M, i64 K, double *tmp);
void oracle_bfd(double *z, const double *q, i64 nt, i64 nx, i64 ny, double s, double dF);
void oracle_bfd_conj(double *q, const double *w, i64 nt, i64 nx, i64 ny, double s);
void oracle_bfd1d(double *z, const double *q, i64 nt, i64 nx, double s, double dF);
void oracle_bfd_conj1d(double *q, const double *w, i64 nt, i64 nx, double s);

/* fake_mx harness */
mxArray *fmx_wrap_double(size_t m, size_t n, double *data);
mxArray *fmx_string(const char *s);
mxArray *fmx_struct(int nfields, const char **names);
void fmx_free(mxArray *a);
typedef void (*mexfun_t)(int, mxArray **, int, const mxArray **);
int fmx_call(mexfun_t fn, int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs);
const char *fmx_error_id(void);
const char *fmx_error_msg(void);
void fmx_clear_mex(void);

#define GATE(name) void mexFunction_##name(int, mxArray **, int, const mxArray **)
GATE(mexProjSoc); GATE(mexBFd); GATE(mexBFdConj); GATE(mexBFd1d); GATE(mexBFdConj1d); GATE(dotsocp_inpalm_mex);
GATE(dotsocp_level_mex);

static int g_fail = 0, g_checks = 0;
#define CHECK(cond, ...)                                                         \
    do {                                                                         \
        ++g_checks;                                                              \
        if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } \
    } while (0)

static unsigned long long g_seed = 88172645463325252ull;
static double rnd(void) {       /* xorshift, uniform in (-1, 1) */
    g_seed ^= g_seed << 13; g_seed ^= g_seed >> 7; g_seed ^= g_seed << 17;
    return (double)(g_seed >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}
static double *vec(size_t n, int random) {
    double *p = (double *)malloc((n ? n : 1) * sizeof(double));      /* exact size: ASAN sees any overrun */
    for (size_t i = 0; i < n; ++i) p[i] = random ? rnd() : 0.0;
    return p;
}
static double dot(const double *a, const double *b, size_t n) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static void oracle_ops(void) {
    static const i64 dims[][3] = {{4, 6, 5}, {2, 1, 1}, {2, 2, 1}, {2, 1, 2}, {3, 7, 2}, {9, 3, 8}};   /* nt, nx, ny */
    for (size_t c = 0; c < sizeof dims / sizeof dims[0]; ++c) {
        const i64 nt = dims[c][0], nx = dims[c][1], ny = dims[c][2];
        const size_t Nz = (size_t)(ny * nx * (nt - 1));
        const size_t Nq = Nz + (size_t)(ny * (nx - 1) * nt) + (size_t)((ny - 1) * nx * nt);
        double *q = vec(Nq, 1), *w = vec(10 * Nz, 1), *z = vec(10 * Nz, 0), *qa = vec(Nq, 0);
        double *out = vec(10 * Nz, 0), *tmp = vec(3 * Nz, 0);
        oracle_bfd(z, q, nt, nx, ny, 0.83, 0.0);
        oracle_bfd_conj(qa, w, nt, nx, ny, 0.83);
        const double lhs = dot(z, w, 10 * Nz), rhs = dot(q, qa, Nq);
        CHECK(fabs(lhs - rhs) <= 1e-12 * (1 + fabs(lhs)), "2-D adjoint identity %lldx%lldx%lld: %g vs %g", ny, nx, nt, lhs, rhs);
        w[0] = w[Nz] = 0.0;                                   /* an all-zero row: NaN row like the reference */
        for (int j = 2; j < 10; ++j) w[(size_t)j * Nz] = 0.0;
        oracle_proj_soc(out, w, (i64)Nz, 10, tmp);
        int in_cone = 1;
        for (size_t i = 1; i < Nz; ++i) {
            double n2 = 0;
            for (int j = 1; j < 10; ++j) n2 += out[j * Nz + i] * out[j * Nz + i];
            if (!(out[i] >= sqrt(n2) * (1 - 1e-14))) in_cone = 0;
        }
        CHECK(in_cone, "projection leaves the cone");
        free(q); free(w); free(z); free(qa); free(out); free(tmp);
        /* 1-D twins on (nt, nx = ny*nx or at least 1) */
        const i64 n1 = nx * ny;
        const size_t Nz1 = (size_t)(n1 * (nt - 1)), Nq1 = Nz1 + (size_t)((n1 - 1) * nt);
        double *q1 = vec(Nq1, 1), *w1 = vec(6 * Nz1, 1), *z1 = vec(6 * Nz1, 0), *qa1 = vec(Nq1, 0);
        oracle_bfd1d(z1, q1, nt, n1, 1.2, 0.0);
        oracle_bfd_conj1d(qa1, w1, nt, n1, 1.2);
        const double l1 = dot(z1, w1, 6 * Nz1), r1 = dot(q1, qa1, Nq1);
        CHECK(fabs(l1 - r1) <= 1e-12 * (1 + fabs(l1)), "1-D adjoint identity %lldx%lld", n1, nt);
        double *o1 = vec(6 * Nz1, 0), *t1 = vec(3 * Nz1, 0);
        oracle_proj_soc(o1, w1, (i64)Nz1, 6, t1);
        free(q1); free(w1); free(z1); free(qa1); free(o1); free(t1);
    }
}

/* ---- gateway calls ---- */
#define MAXA 8
typedef struct { mxArray *a[MAXA]; int n; } args_t;
static void push(args_t *A, mxArray *m) { A->a[A->n++] = m; }
static mxArray *scal(double v) { return mxCreateDoubleScalar(v); }
static int call(mexfun_t fn, args_t *A, int nlhs, const char *want_id) {
    mxArray *plhs[2] = {NULL, NULL};
    const int rc = fmx_call(fn, nlhs, plhs, A->n, (const mxArray **)A->a);
    int ok = 1;
    if (want_id) ok = rc == 1 && strcmp(fmx_error_id(), want_id) == 0;
    else ok = rc == 0;
    if (!ok) fprintf(stderr, "  got rc=%d id='%s' msg='%s' (wanted '%s')\n", rc, fmx_error_id(), fmx_error_msg(), want_id ? want_id : "success");
    for (int i = 0; i < A->n; ++i) fmx_free(A->a[i]);
    for (int i = 0; i < 2; ++i) if (plhs[i]) fmx_free(plhs[i]);
    A->n = 0;
    return ok;
}

static void set(mxArray *s, const char *name, mxArray *v) { mxSetField(s, 0, name, v); }

static mxArray *state_struct(i64 ny, i64 nx, i64 nt, const char *short_field, int with_weight) {
    static const char *names[] = {"phi", "q", "alpha", "z", "beta", "c", "weight", "nx", "ny", "nt", "D", "E", "cScale", "dScale",
                                  "normc", "normd"};
    mxArray *S = fmx_struct(16, names);
    const size_t Nphi = (size_t)(ny * nx * nt), Nz = (size_t)(ny * nx * (nt - 1));
    const size_t Nq = Nz + (size_t)(ny * (nx - 1) * nt) + (size_t)((ny - 1) * nx * nt);
    const char *f[] = {"phi", "q", "alpha", "z", "beta", "c", "weight"};
    const size_t len[] = {Nphi, Nq, Nq, 10 * Nz, 10 * Nz, Nphi, Nq};
    for (int i = 0; i < 7; ++i) {
        if (i == 6 && !with_weight) continue;
        size_t n = len[i];
        if (short_field && strcmp(short_field, f[i]) == 0) n -= 1;
        set(S, f[i], mxCreateDoubleMatrix(n, 1, mxREAL));
    }
    set(S, "nx", scal((double)nx)); set(S, "ny", scal((double)ny)); set(S, "nt", scal((double)nt));
    set(S, "D", scal(1.0)); set(S, "E", scal(0.7)); set(S, "cScale", scal(1.0)); set(S, "dScale", scal(1.0));
    set(S, "normc", scal(1.0)); set(S, "normd", scal(1.0));
    return S;
}

static mxArray *opts_struct(const char *drop, const char *method) {
    static const char *names[] = {"tau", "sigma", "maxit", "tol", "ifCheckStepByStep", "scaling", "method", "ngpu"};
    mxArray *O = fmx_struct(8, names);
    const char *f[] = {"tau", "sigma", "maxit", "tol", "ifCheckStepByStep", "scaling"};
    const double v[] = {1.9, 1.0, 5, 0.0, 0.0, 1.0};
    for (int i = 0; i < 6; ++i)
        if (!drop || strcmp(drop, f[i]) != 0) set(O, f[i], scal(v[i]));
    if (method) set(O, "method", fmx_string(method));
    return O;
}

static void gateways(void) {
    args_t A = {{0}, 0};
    const i64 nt = 4, nx = 6, ny = 5;
    const size_t Nz = (size_t)(ny * nx * (nt - 1)), Nq = Nz + (size_t)(ny * (nx - 1) * nt) + (size_t)((ny - 1) * nx * nt);
    /* operator gateways: argument errors */
    push(&A, mxCreateDoubleMatrix(Nz, 10, mxREAL));
    CHECK(call(mexFunction_mexProjSoc, &A, 0, "mexProjSoc:invalidNumInputs"), "mexProjSoc with one argument");
    push(&A, mxCreateDoubleMatrix(Nz, 10, mxREAL)); push(&A, mxCreateDoubleMatrix(Nz + 1, 10, mxREAL));
    CHECK(call(mexFunction_mexProjSoc, &A, 0, "mexProjSoc:invalidInput"), "mexProjSoc size mismatch");
    push(&A, mxCreateDoubleMatrix(Nz, 10, mxREAL)); push(&A, mxCreateDoubleMatrix(Nz, 10, mxREAL));
    CHECK(call(mexFunction_mexProjSoc, &A, 1, "mexProjSoc:invalidNumOutputs"), "mexProjSoc with an output");
    push(&A, mxCreateDoubleMatrix(Nz, 10, mxREAL)); push(&A, mxCreateDoubleMatrix(Nq, 1, mxREAL));
    push(&A, scal((double)nt));
    CHECK(call(mexFunction_mexBFd, &A, 0, "mexBFd:invalidNumInputs"), "mexBFd with three arguments");
    push(&A, mxCreateDoubleMatrix(Nz, 10, mxREAL)); push(&A, mxCreateDoubleMatrix(Nq - 1, 1, mxREAL));
    push(&A, scal((double)nt)); push(&A, scal((double)nx)); push(&A, scal((double)ny));
    CHECK(call(mexFunction_mexBFd, &A, 0, "mexBFd:invalidInput"), "mexBFd short q");
    push(&A, mxCreateDoubleMatrix(Nq, 1, mxREAL)); push(&A, mxCreateDoubleMatrix(Nz, 9, mxREAL));
    push(&A, scal((double)nt)); push(&A, scal((double)nx)); push(&A, scal((double)ny));
    CHECK(call(mexFunction_mexBFdConj, &A, 0, "mexBFdConj:invalidInput"), "mexBFdConj narrow z");
    const size_t Nz1 = (size_t)(nx * (nt - 1)), Nq1 = Nz1 + (size_t)((nx - 1) * nt);
    push(&A, mxCreateDoubleMatrix(Nz1, 6, mxREAL)); push(&A, mxCreateDoubleMatrix(Nq1, 1, mxREAL)); push(&A, scal((double)nt));
    CHECK(call(mexFunction_mexBFd1d, &A, 0, "mexBFd:invalidNumInputs"), "mexBFd1d with three arguments");
    push(&A, mxCreateDoubleMatrix(Nz1, 6, mxREAL)); push(&A, mxCreateDoubleMatrix(Nq1, 1, mxREAL)); push(&A, scal((double)nt));
    push(&A, scal((double)nx));
    CHECK(call(mexFunction_mexBFd1d, &A, 1, "mexBFd:invalidNumOutputs"), "mexBFd1d with an output");
    push(&A, mxCreateDoubleMatrix(Nz1, 6, mxREAL)); push(&A, mxCreateDoubleMatrix(Nq1, 1, mxREAL)); push(&A, scal((double)nt));
    push(&A, scal((double)nx)); push(&A, mxCreateDoubleMatrix(2, 1, mxREAL));
    CHECK(call(mexFunction_mexBFd1d, &A, 0, "mexBFd:invalidInput"), "mexBFd1d non-scalar scale");
    push(&A, mxCreateDoubleMatrix(Nz1 + 1, 6, mxREAL)); push(&A, mxCreateDoubleMatrix(Nq1, 1, mxREAL)); push(&A, scal((double)nt));
    push(&A, scal((double)nx));
    CHECK(call(mexFunction_mexBFd1d, &A, 0, "mexBFd:invalidInput"), "mexBFd1d long z");
    push(&A, mxCreateDoubleMatrix(Nq1, 1, mxREAL)); push(&A, mxCreateDoubleMatrix(Nz1, 6, mxREAL)); push(&A, scal((double)nt));
    CHECK(call(mexFunction_mexBFdConj1d, &A, 0, "mexBFd:invalidNumInputs"), "mexBFdConj1d with three arguments");
    /* solver gateway: usage, missing field, unknown method, short arrays (all before any context exists) */
    push(&A, scal(1.0)); push(&A, scal(2.0));
    CHECK(call(mexFunction_dotsocp_inpalm_mex, &A, 1, "dotsocp:inPALM"), "inpalm gateway with non-structs");
    push(&A, state_struct(ny, nx, nt, NULL, 0)); push(&A, opts_struct("sigma", NULL));
    CHECK(call(mexFunction_dotsocp_inpalm_mex, &A, 1, "dotsocp:inPALM"), "inpalm gateway without opts.sigma");
    push(&A, state_struct(ny, nx, nt, NULL, 0)); push(&A, opts_struct(NULL, "fastest"));
    CHECK(call(mexFunction_dotsocp_inpalm_mex, &A, 1, "dotsocp:inPALM"), "inpalm gateway with an unknown method");
    static const char *fields[] = {"phi", "q", "alpha", "z", "beta", "c", "weight"};
    for (int i = 0; i < 7; ++i) {
        push(&A, state_struct(ny, nx, nt, fields[i], 1)); push(&A, opts_struct(NULL, "inPALM"));
        CHECK(call(mexFunction_dotsocp_inpalm_mex, &A, 1, "dotsocp:inPALM:size"), "inpalm gateway with a short %s", fields[i]);
    }
    push(&A, state_struct(ny, nx, 1, NULL, 0)); push(&A, opts_struct(NULL, NULL));
    CHECK(call(mexFunction_dotsocp_inpalm_mex, &A, 1, "dotsocp:inPALM:size"), "inpalm gateway with nt = 1");
    /* level gateway */
    push(&A, scal(3.0));
    CHECK(call(mexFunction_dotsocp_level_mex, &A, 1, "dotsocp:level"), "level gateway without a command");
    push(&A, fmx_string("explode"));
    CHECK(call(mexFunction_dotsocp_level_mex, &A, 1, "dotsocp:level"), "level gateway with an unknown command");
    push(&A, fmx_string("solve")); push(&A, scal(7.0));
    CHECK(call(mexFunction_dotsocp_level_mex, &A, 1, "dotsocp:level"), "level gateway with a stale handle");
    push(&A, fmx_string("create")); push(&A, state_struct(ny, nx, nt, "c", 0)); push(&A, opts_struct(NULL, NULL));
    CHECK(call(mexFunction_dotsocp_level_mex, &A, 1, "dotsocp:level:size"), "level gateway with a short c");
    push(&A, fmx_string("create")); push(&A, state_struct(ny, nx, nt, "beta", 0)); push(&A, opts_struct(NULL, NULL));
    CHECK(call(mexFunction_dotsocp_level_mex, &A, 1, "dotsocp:level:size"), "level gateway with a short beta");
    fmx_clear_mex();
}

int main(void) {
    oracle_ops();
    gateways();
    printf("sanitizer driver: %d checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
