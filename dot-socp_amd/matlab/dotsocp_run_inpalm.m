function [runHist, sigma] = dotsocp_run_inpalm(var, opts, model, weighted, method)
%% Shared body of the wrappers: handle objects -> struct -> MEX -> handle objects.
% method: 'inPALM' (default; ALG2 through opts.tau), 'PALM' or 'accADMM'.
    if nargin < 5, method = 'inPALM'; end
    opts.method = method;
    S = struct('phi', var.phi, 'q', var.q, 'alpha', var.alpha, 'z', var.z, 'beta', var.beta, ...
               'c', model.c, 'nx', model.nx, 'nt', model.nt, 'D', var.D, 'E', var.E, ...
               'cScale', var.cScale, 'dScale', var.dScale, 'normc', model.normc);
    if isprop(model, 'ny') && ~isempty(model.ny), S.ny = model.ny; end
    if isprop(model, 'normd') && ~isempty(model.normd), S.normd = model.normd; end
    if weighted, S.weight = model.weight; end

    out = dotsocp_inpalm_mex(S, opts);

    t = out.times;      % Step_1_1_FFT, Step_1_2_ProjSOC, Step_2_Q_Step, Step_3_Multiplier, KKT, Total_Time, Iters
    switch method
        case 'PALM'      % solver_socp_PALM.m:341,351-352
            var.name = 'Proximal ALM';
            names = {'Step_1_Q_Step', 'Step_2_1_FFT', 'Step_2_2_ProjSOC', 'Step_3_Q_Step', 'Step_4_Multiplier', 'KKT', 'Total_Time', 'Iters'};
            times = [out.time_extra, t];
        case 'accADMM'   % solver_socp_accADMM.m:428,438-439; solver_wsocp_accADMM.m:443-444
            var.name = 'Accelerated ADMM';
            if weighted
                names = {'Step_1_Q_Step', 'Step_2_Multiplier', 'Step_3_1_FFT', 'Step_3_2_ProjSOC', 'Step_4_Interp', 'KKT', 'Total_Time', 'Iters'};
                times = [t(3), t(4), t(1), t(2), out.time_extra, t(5), t(6), t(7)];
            else
                names = {'Step_1_Q_Step', 'Step_2_Multiplier', 'Step_3_1_FFT', 'Step_3_2_ProjSOC', 'KKT', 'Interp', 'Total_Time', 'Iters'};
                times = [t(3), t(4), t(1), t(2), t(5), out.time_extra, t(6), t(7)];
            end
        otherwise        % solver_socp_inPALM.m:329,339-341
            var.name = 'Inexact Proximal ALM';
            names = {'Step_1_1_FFT', 'Step_1_2_ProjSOC', 'Step_2_Q_Step', 'Step_3_Multiplier', 'KKT', 'Total_Time', 'Iters'};
            times = t;
    end
    var.phi   = out.phi;   var.q    = out.q;   var.z = out.z;
    var.alpha = out.alpha; var.beta = out.beta;          % already multiplied by sigma
    var.time   = array2table(times, 'VariableNames', names);
    var.cScale = out.cScale;
    var.dScale = out.dScale;
    runHist = struct('kkt', out.kkt, 'time', out.time, 'iter', out.iter, 'pdGap', out.pdGap, 'len', numel(out.iter));
    sigma = out.sigma;
end
