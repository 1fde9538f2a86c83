/* mexProjSoc(out, in) -- drop-in for socp/{dot1d,dot2d,wdot2d}/utils/mexProjSoc.mex* :
 * row-wise projection of the M x K matrix `in` onto the second-order cone, written IN PLACE into
 * prhs[0] (call sites socp/dot2d/algorithms/solver_socp_inPALM.m:199,240). */
#include "mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    (void)plhs;
    if (nrhs != 2) mexErrMsgIdAndTxt("mexProjSoc:invalidNumInputs", "usage: mexProjSoc(out, in)");
    if (nlhs > 0) mexErrMsgIdAndTxt("mexProjSoc:invalidNumOutputs", "mexProjSoc writes into its first argument");
    const size_t M = mxGetM(prhs[1]), K = mxGetN(prhs[1]);
    if (mxGetM(prhs[0]) != M || mxGetN(prhs[0]) != K)
        mexErrMsgIdAndTxt("mexProjSoc:invalidInput", "out and in must have the same size");
    DS_MEX_CHECK(dotsocp_proj_soc(ds_real(prhs[0], "mexProjSoc:invalidInput", "out"),
                                  ds_real(prhs[1], "mexProjSoc:invalidInput", "in"),
                                  (dotsocp_i64)M, (dotsocp_i64)K), "mexProjSoc:device");
}
