/* mexBFd1d(z, q, nt, nx[, scale[, dF]]) -- drop-in for socp/dot1d/utils/mexBFd1d.mex* (z is Nz x 6);
 * error identifiers as in the original: mexBFd:invalidNumInputs / invalidNumOutputs / invalidInput. */
#include "mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    (void)plhs;
    if (nrhs < 4 || nrhs > 6) mexErrMsgIdAndTxt("mexBFd:invalidNumInputs", "usage: mexBFd1d(z, q, nt, nx[, scale[, dF]])");
    if (nlhs > 0) mexErrMsgIdAndTxt("mexBFd:invalidNumOutputs", "mexBFd1d writes into its first argument");
    const dotsocp_i64 nt = (dotsocp_i64)ds_scalar(prhs[2], "mexBFd:invalidInput", "nt");
    const dotsocp_i64 nx = (dotsocp_i64)ds_scalar(prhs[3], "mexBFd:invalidInput", "nx");
    const double scale = nrhs > 4 ? ds_scalar(prhs[4], "mexBFd:invalidInput", "scale") : 1.0;
    const double dF = nrhs > 5 ? ds_scalar(prhs[5], "mexBFd:invalidInput", "dF") : 1.0;
    const dotsocp_i64 Nz = nx * (nt - 1), Nq = Nz + (nx - 1) * nt;
    if ((dotsocp_i64)mxGetNumberOfElements(prhs[0]) != 6 * Nz || (dotsocp_i64)mxGetNumberOfElements(prhs[1]) != Nq)
        mexErrMsgIdAndTxt("mexBFd:invalidInput", "z must be Nz x 6 and q of length Nq");
    DS_MEX_CHECK(dotsocp_bfd1d(ds_real(prhs[0], "mexBFd:invalidInput", "z"), ds_real(prhs[1], "mexBFd:invalidInput", "q"),
                               nt, nx, scale, dF), "mexBFd:device");
}
