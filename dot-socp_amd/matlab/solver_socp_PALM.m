function [runHist, sigma] = solver_socp_PALM(var, opts, model)
%% Drop-in replacement of socp/dot2d/algorithms/solver_socp_PALM.m: the proximal ALM loop runs on
% the MI355X inside libdotsocp (csrc/solver_palm.hip).
    [runHist, sigma] = dotsocp_run_inpalm(var, opts, model, false, 'PALM');
end
