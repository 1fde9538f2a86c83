/* mexBFdConj1d(q, z, nt, nx[, scale]) -- drop-in for socp/dot1d/utils/mexBFdConj1d.mex*. */
#include "mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    (void)plhs;
    if (nrhs < 4 || nrhs > 5) mexErrMsgIdAndTxt("mexBFd:invalidNumInputs", "usage: mexBFdConj1d(q, z, nt, nx[, scale])");
    if (nlhs > 0) mexErrMsgIdAndTxt("mexBFd:invalidNumOutputs", "mexBFdConj1d writes into its first argument");
    const dotsocp_i64 nt = (dotsocp_i64)ds_scalar(prhs[2], "mexBFd:invalidInput", "nt");
    const dotsocp_i64 nx = (dotsocp_i64)ds_scalar(prhs[3], "mexBFd:invalidInput", "nx");
    const double scale = nrhs > 4 ? ds_scalar(prhs[4], "mexBFd:invalidInput", "scale") : 1.0;
    const dotsocp_i64 Nz = nx * (nt - 1), Nq = Nz + (nx - 1) * nt;
    if ((dotsocp_i64)mxGetNumberOfElements(prhs[1]) != 6 * Nz || (dotsocp_i64)mxGetNumberOfElements(prhs[0]) != Nq)
        mexErrMsgIdAndTxt("mexBFd:invalidInput", "z must be Nz x 6 and q of length Nq");
    DS_MEX_CHECK(dotsocp_bfd_conj1d(ds_real(prhs[0], "mexBFd:invalidInput", "q"), ds_real(prhs[1], "mexBFd:invalidInput", "z"),
                                    nt, nx, scale), "mexBFd:device");
}
