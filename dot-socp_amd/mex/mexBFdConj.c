/* mexBFdConj(q, z, nt, nx, ny[, scale = 1]) -- drop-in for socp/dot2d/utils/mexBFdConj.mex* :
 * q <- F* B* z, in place in prhs[0] (call sites solver_socp_inPALM.m:205,225; jump_nextLevel.m:16). */
#include "mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    (void)plhs;
    if (nrhs < 5 || nrhs > 6) mexErrMsgIdAndTxt("mexBFdConj:invalidNumInputs", "usage: mexBFdConj(q, z, nt, nx, ny[, scale])");
    if (nlhs > 0) mexErrMsgIdAndTxt("mexBFdConj:invalidNumOutputs", "mexBFdConj writes into its first argument");
    const dotsocp_i64 nt = (dotsocp_i64)ds_scalar(prhs[2], "mexBFdConj:invalidInput", "nt");
    const dotsocp_i64 nx = (dotsocp_i64)ds_scalar(prhs[3], "mexBFdConj:invalidInput", "nx");
    const dotsocp_i64 ny = (dotsocp_i64)ds_scalar(prhs[4], "mexBFdConj:invalidInput", "ny");
    const double scale = nrhs > 5 ? ds_scalar(prhs[5], "mexBFdConj:invalidInput", "scale") : 1.0;
    const dotsocp_i64 Nz = ny * nx * (nt - 1), Nq = Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt;
    if ((dotsocp_i64)mxGetNumberOfElements(prhs[1]) != 10 * Nz || (dotsocp_i64)mxGetNumberOfElements(prhs[0]) != Nq)
        mexErrMsgIdAndTxt("mexBFdConj:invalidInput", "z must be Nz x 10 and q of length Nq");
    DS_MEX_CHECK(dotsocp_bfd_conj(ds_real(prhs[0], "mexBFdConj:invalidInput", "q"),
                                  ds_real(prhs[1], "mexBFdConj:invalidInput", "z"), nt, nx, ny, scale), "mexBFdConj:device");
}
