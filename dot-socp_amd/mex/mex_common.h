/* Shared helpers of the MEX gateways (real, non-interleaved API like the reference binaries). */
#ifndef DOTSOCP_MEX_COMMON_H
#define DOTSOCP_MEX_COMMON_H
#include "mex.h"
#include "dotsocp.h"

#define DS_MEX_CHECK(call, id)                                        \
    do {                                                              \
        int rc__ = (call);                                            \
        if (rc__ != 0) mexErrMsgIdAndTxt(id, "%s", dotsocp_last_error()); \
    } while (0)

static inline double *ds_real(const mxArray *a, const char *id, const char *what) {
    if (!mxIsDouble(a) || mxIsComplex(a) || mxIsSparse(a))
        mexErrMsgIdAndTxt(id, "%s must be a full real double array", what);
    return mxGetPr(a);
}

static inline double ds_scalar(const mxArray *a, const char *id, const char *what) {
    if (!mxIsDouble(a) || mxGetNumberOfElements(a) != 1)
        mexErrMsgIdAndTxt(id, "%s must be a scalar", what);
    return mxGetScalar(a);
}
#endif
