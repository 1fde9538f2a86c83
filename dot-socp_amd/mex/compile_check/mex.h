/* COMPILE-CHECK STUB -- NOT MATLAB'S HEADER.
 * MATLAB is not available in this environment.  This file declares, from the public MATLAB C
 * Matrix / MEX API documentation, only the handful of functions the gateways in this directory
 * use, so that `make -C dot-socp_amd/mex check` can type-check them with gcc.  A real build uses
 * MATLAB's own <mex.h> through the `mex` command; nothing here is linked
 * or executed. */
#ifndef DOTSOCP_MEX_STUB_H
#define DOTSOCP_MEX_STUB_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;
#ifdef __cplusplus
extern "C" {
#endif
double *mxGetPr(const mxArray *pa);
double mxGetScalar(const mxArray *pa);
size_t mxGetM(const mxArray *pa);
size_t mxGetN(const mxArray *pa);
size_t mxGetNumberOfElements(const mxArray *pa);
int mxIsDouble(const mxArray *pa);
int mxIsComplex(const mxArray *pa);
int mxIsSparse(const mxArray *pa);
int mxIsStruct(const mxArray *pa);
int mxIsChar(const mxArray *pa);
int mxIsEmpty(const mxArray *pa);
mxArray *mxGetField(const mxArray *pa, size_t index, const char *fieldname);
mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double value);
mxArray *mxCreateStructMatrix(size_t m, size_t n, int nfields, const char **fieldnames);
void mxSetField(mxArray *pa, size_t index, const char *fieldname, mxArray *value);
int mxGetString(const mxArray *pa, char *buf, size_t buflen);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
int mexAtExit(void (*fn)(void));
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
