/* mexBFd(z, q, nt, nx, ny[, scale = 1[, dF = 1]]) -- drop-in for socp/dot2d/utils/mexBFd.mex* :
 * z <- B F q + d, in place in prhs[0] (call sites solver_socp_inPALM.m:133,187,212,242).
 * Scalars arrive as doubles and are truncated to integers like the original (cvttsd2si). */
#include "mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    (void)plhs;
    if (nrhs < 5 || nrhs > 7) mexErrMsgIdAndTxt("mexBFd:invalidNumInputs", "usage: mexBFd(z, q, nt, nx, ny[, scale[, dF]])");
    if (nlhs > 0) mexErrMsgIdAndTxt("mexBFd:invalidNumOutputs", "mexBFd writes into its first argument");
    const dotsocp_i64 nt = (dotsocp_i64)ds_scalar(prhs[2], "mexBFd:invalidInput", "nt");
    const dotsocp_i64 nx = (dotsocp_i64)ds_scalar(prhs[3], "mexBFd:invalidInput", "nx");
    const dotsocp_i64 ny = (dotsocp_i64)ds_scalar(prhs[4], "mexBFd:invalidInput", "ny");
    const double scale = nrhs > 5 ? ds_scalar(prhs[5], "mexBFd:invalidInput", "scale") : 1.0;
    const double dF = nrhs > 6 ? ds_scalar(prhs[6], "mexBFd:invalidInput", "dF") : 1.0;
    const dotsocp_i64 Nz = ny * nx * (nt - 1), Nq = Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt;
    if ((dotsocp_i64)mxGetNumberOfElements(prhs[0]) != 10 * Nz || (dotsocp_i64)mxGetNumberOfElements(prhs[1]) != Nq)
        mexErrMsgIdAndTxt("mexBFd:invalidInput", "z must be Nz x 10 and q of length Nq");
    DS_MEX_CHECK(dotsocp_bfd(ds_real(prhs[0], "mexBFd:invalidInput", "z"), ds_real(prhs[1], "mexBFd:invalidInput", "q"),
                             nt, nx, ny, scale, dF), "mexBFd:device");
}
