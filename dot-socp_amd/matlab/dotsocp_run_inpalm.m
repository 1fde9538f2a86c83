function [runHist, sigma] = dotsocp_run_inpalm(var, opts, model, weighted)
%% Shared body of the two wrappers: handle objects -> struct -> MEX -> handle objects.
    S = struct('phi', var.phi, 'q', var.q, 'alpha', var.alpha, 'z', var.z, 'beta', var.beta, ...
               'c', model.c, 'nx', model.nx, 'nt', model.nt, 'D', var.D, 'E', var.E, ...
               'cScale', var.cScale, 'dScale', var.dScale, 'normc', model.normc);
    if isprop(model, 'ny') && ~isempty(model.ny), S.ny = model.ny; end
    if isprop(model, 'normd') && ~isempty(model.normd), S.normd = model.normd; end
    if weighted, S.weight = model.weight; end

    out = dotsocp_inpalm_mex(S, opts);

    var.name  = 'Inexact Proximal ALM';
    var.phi   = out.phi;   var.q    = out.q;   var.z = out.z;
    var.alpha = out.alpha; var.beta = out.beta;          % already multiplied by sigma
    names = {'Step_1_1_FFT', 'Step_1_2_ProjSOC', 'Step_2_Q_Step', 'Step_3_Multiplier', 'KKT', 'Total_Time', 'Iters'};
    var.time   = array2table(out.times, 'VariableNames', names);
    var.cScale = out.cScale;
    var.dScale = out.dScale;
    runHist = struct('kkt', out.kkt, 'time', out.time, 'iter', out.iter, 'pdGap', out.pdGap, 'len', numel(out.iter));
    sigma = out.sigma;
end
