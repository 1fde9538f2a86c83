function [runHist, sigma] = solver_wsocp_accADMM(var, opts, model)
%% Drop-in replacement of socp/wdot2d/algorithms/solver_wsocp_accADMM.m (model.weight required).
    [runHist, sigma] = dotsocp_run_inpalm(var, opts, model, true, 'accADMM');
end
