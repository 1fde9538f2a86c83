function [runHist, sigma] = solver_socp_accADMM(var, opts, model)
%% Drop-in replacement of socp/dot2d/algorithms/solver_socp_accADMM.m: the accelerated ADMM loop
% (Halpern iteration for opts.theta == 2, restart / rho / theta as in :12-34) runs on the MI355X
% inside libdotsocp (csrc/solver_acc.hip).
    [runHist, sigma] = dotsocp_run_inpalm(var, opts, model, false, 'accADMM');
end
