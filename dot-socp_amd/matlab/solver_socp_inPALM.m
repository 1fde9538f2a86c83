function [runHist, sigma] = solver_socp_inPALM(var, opts, model)
%% Drop-in replacement of socp/dot2d/algorithms/solver_socp_inPALM.m (and the dot1d twin):
% the whole inPALM / ALG2 loop runs on the MI355X inside libdotsocp; this wrapper only
% marshals the handle objects across the MEX gateway dotsocp_inpalm_mex.
% Put this directory in front of socp/<variant>/algorithms on the MATLAB path.
    [runHist, sigma] = dotsocp_run_inpalm(var, opts, model, false);
end
