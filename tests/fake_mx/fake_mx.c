/* TEST INFRASTRUCTURE -- a stand-in for MATLAB's libmx / libmex, written from the public C Matrix API
 * documentation, just large enough to EXECUTE the MEX gateways of dot-socp_amd/mex/ on a box without MATLAB:
 * real double matrices, 1 x 1 structs, char row vectors, and mexErrMsgIdAndTxt as a non-local exit back to
 * the harness (fmx_call).  Nothing of MATLAB or of the reference is loaded.  tests/test_gpu_mex_gateways.py
 * builds it, builds the gateways against it and calls their mexFunction through ctypes. */
#define _POSIX_C_SOURCE 200809L
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;
typedef struct mxArray_tag {
    int kind;                 /* 0 double, 1 struct, 2 char */
    size_t m, n;
    double *pr;
    int owns;
    int nfields;
    char **names;
    struct mxArray_tag **vals;
    char *str;
} mxArray;

static jmp_buf g_jmp;
static int g_active = 0;
static char g_err_id[128], g_err_msg[1024];

/* ---- the part of the C Matrix API the gateways use ---- */
double *mxGetPr(const mxArray *a) { return a ? a->pr : NULL; }
double mxGetScalar(const mxArray *a) { return (a && a->pr && a->m * a->n > 0) ? a->pr[0] : 0.0; }
size_t mxGetM(const mxArray *a) { return a->m; }
size_t mxGetN(const mxArray *a) { return a->n; }
size_t mxGetNumberOfElements(const mxArray *a) { return a->m * a->n; }
int mxIsDouble(const mxArray *a) { return a && a->kind == 0; }
int mxIsComplex(const mxArray *a) { (void)a; return 0; }
int mxIsSparse(const mxArray *a) { (void)a; return 0; }
int mxIsStruct(const mxArray *a) { return a && a->kind == 1; }
int mxIsChar(const mxArray *a) { return a && a->kind == 2; }
int mxIsEmpty(const mxArray *a) { return !a || a->m * a->n == 0; }

mxArray *mxGetField(const mxArray *a, size_t index, const char *name) {
    if (!a || a->kind != 1 || index != 0) return NULL;
    for (int i = 0; i < a->nfields; ++i)
        if (strcmp(a->names[i], name) == 0) return a->vals[i];
    return NULL;
}

mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity flag) {
    (void)flag;
    mxArray *a = (mxArray *)calloc(1, sizeof *a);
    a->m = m; a->n = n; a->owns = 1;
    a->pr = (double *)calloc((m * n) > 0 ? m * n : 1, sizeof(double));
    return a;
}

mxArray *mxCreateDoubleScalar(double v) {
    mxArray *a = mxCreateDoubleMatrix(1, 1, mxREAL);
    a->pr[0] = v;
    return a;
}

mxArray *mxCreateStructMatrix(size_t m, size_t n, int nfields, const char **names) {
    mxArray *a = (mxArray *)calloc(1, sizeof *a);
    a->kind = 1; a->m = m; a->n = n; a->nfields = nfields;
    a->names = (char **)calloc(nfields ? nfields : 1, sizeof(char *));
    a->vals = (mxArray **)calloc(nfields ? nfields : 1, sizeof(mxArray *));
    for (int i = 0; i < nfields; ++i) a->names[i] = strdup(names[i]);
    return a;
}

void mxSetField(mxArray *a, size_t index, const char *name, mxArray *value) {
    if (!a || a->kind != 1 || index != 0) return;
    for (int i = 0; i < a->nfields; ++i)
        if (strcmp(a->names[i], name) == 0) { a->vals[i] = value; return; }
}

int mxGetString(const mxArray *a, char *buf, size_t buflen) {
    if (!a || a->kind != 2 || !a->str || strlen(a->str) + 1 > buflen) return 1;
    strcpy(buf, a->str);
    return 0;
}

void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err_msg, sizeof g_err_msg, fmt, ap);
    va_end(ap);
    snprintf(g_err_id, sizeof g_err_id, "%s", id ? id : "");
    if (g_active) longjmp(g_jmp, 1);
    fprintf(stderr, "mexErrMsgIdAndTxt outside fmx_call: %s: %s\n", g_err_id, g_err_msg);
    abort();
}

/* exit handlers run when the harness "clears" the MEX file (fmx_clear_mex) */
static void (*g_exit_fn[8])(void);
static int g_exit_n = 0;

int mexAtExit(void (*fn)(void)) {
    if (g_exit_n < 8) g_exit_fn[g_exit_n++] = fn;
    return 0;
}

void fmx_clear_mex(void) {
    for (int i = 0; i < g_exit_n; ++i) g_exit_fn[i]();
    g_exit_n = 0;
}

/* ---- harness side ---- */
mxArray *fmx_wrap_double(size_t m, size_t n, double *data) {       /* aliases the caller's memory (in-place gateways) */
    mxArray *a = (mxArray *)calloc(1, sizeof *a);
    a->m = m; a->n = n; a->pr = data;
    return a;
}

mxArray *fmx_string(const char *s) {
    mxArray *a = (mxArray *)calloc(1, sizeof *a);
    a->kind = 2; a->m = 1; a->n = strlen(s); a->str = strdup(s);
    return a;
}

mxArray *fmx_struct(int nfields, const char **names) { return mxCreateStructMatrix(1, 1, nfields, names); }

void fmx_free(mxArray *a) {
    if (!a) return;
    if (a->kind == 1) {
        for (int i = 0; i < a->nfields; ++i) { fmx_free(a->vals[i]); free(a->names[i]); }
        free(a->names); free(a->vals);
    }
    if (a->owns) free(a->pr);
    free(a->str);
    free(a);
}

typedef void (*mexfun_t)(int, mxArray **, int, const mxArray **);

/* 0: returned normally; 1: the gateway raised mexErrMsgIdAndTxt (fmx_error_id / fmx_error_msg) */
int fmx_call(mexfun_t fn, int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {
    g_err_id[0] = g_err_msg[0] = 0;
    g_active = 1;
    int rc = 0;
    if (setjmp(g_jmp) == 0) fn(nlhs, plhs, nrhs, prhs);
    else rc = 1;
    g_active = 0;
    return rc;
}

const char *fmx_error_id(void) { return g_err_id; }
const char *fmx_error_msg(void) { return g_err_msg; }
