// Host side of the device-resident accelerated ADMM loop: socp/dot2d/algorithms/solver_socp_accADMM.m
// (weighted: socp/wdot2d/algorithms/solver_wsocp_accADMM.m), line numbers below refer to the former.
//
// The KKT block (:251-366) is the inPALM one (Solver::kkt_block) evaluated at the raw ADMM outputs
// x^+ = (phi^+, z^+, q^+, alpha^+, beta^+) of the iteration; only the sigma update differs (it also
// scales the previous iterates and restarts the extrapolation, :346-358).
//
// Device dataflow.  At the top of an iteration the "Old" copies of the reference equal the current
// state, so only the state x, the outputs x^+ and the anchors x0 (Halpern) or previous extrapolation
// points (theta != 2) are kept.  Per iteration:
//   q-step      q^+, alpha^+, rhs of the phi-step <- phi, alpha, q2 = F*B*(z + beta)   k_qstep_rhs<.,1|2>  (:227-237,243)
//               without a KKT check (Halpern): extrapolation of q and alpha in the same pass
//   cone pass   beta^+, z^+   <- z, beta, q^+                               k_acc_cone           (:236-249)
//               without a KKT check (Halpern): extrapolation of z, beta and the NEXT iteration's q2 in the same pass
//   phi-step    phi^+         <- rhs                                        DCT Poisson          (:241-244)
//   [KKT block at x^+]
//   extrapolation of phi (and of q, alpha, z, beta when not folded)         k_acc_interp         (:369-423)
#include <algorithm>
#include <cmath>
#include <cstring>

#include "solver.h"

namespace dotsocp {

int Solver::acc_alloc() {
    FOR_SLABS(s) {
        const Grid &g = s.g;
        if (!s.q2) {                              // DOTSOCP_FUSED=0 contexts come without the tile buffers
            fused_geometry(g, s.fg);
            DS_CHECK(dzalloc(&s.q_old, g.NqAlloc, s.st));
            DS_CHECK(dzalloc(&s.q2, g.NqAlloc, s.st));
            DS_CHECK(dzalloc(&s.beta2, 10 * g.Nc, s.st));       // pads of rows and columns stay zero (common.h)
            DS_CHECK(dzalloc(&s.sx, s.fg.sx_len, s.st));
            DS_CHECK(dzalloc(&s.sy, s.fg.sy_len, s.st));
            DS_CHECK(dzalloc(&s.alpha2, g.NqAlloc, s.st));
        }
        if (s.phi_p) continue;
        DS_CHECK(dzalloc(&s.phi_p, g.NphiAlloc, s.st));
        DS_CHECK(dzalloc(&s.alpha_p, g.NqAlloc, s.st));
        DS_CHECK(dzalloc(&s.z_p, 10 * g.Nc, s.st));
        DS_CHECK(dzalloc(&s.phi_a, g.NphiAlloc, s.st));
        DS_CHECK(dzalloc(&s.q_a, g.NqAlloc, s.st));
        DS_CHECK(dzalloc(&s.alpha_a, g.NqAlloc, s.st));
        DS_CHECK(dzalloc(&s.z_a, 10 * g.Nc, s.st));
        DS_CHECK(dzalloc(&s.beta_a, 10 * g.Nc, s.st));
    }
    return 0;
}

// anchors <- current state (CopyVar, :162,221,356,387)
int Solver::acc_set_anchors() {
    FOR_SLABS(s) {
        const Grid &g = s.g;
        auto cp = [&](double *dst, const double *src, i64 n) {
            return ds_memcpy_async(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s.st);
        };
        DS_HIP(cp(s.phi_a, s.phi, g.NphiAlloc));
        DS_HIP(cp(s.q_a, s.q, g.NqAlloc));
        DS_HIP(cp(s.alpha_a, s.alpha, g.NqAlloc));
        DS_HIP(cp(s.z_a, s.z, 10 * g.Nc));
        DS_HIP(cp(s.beta_a, s.beta, 10 * g.Nc));
    }
    return 0;
}

int Solver::acc_begin(const dotsocp_acc_opts *acc) {
    acc_restart = (acc && acc->restart > 0) ? acc->restart : 100;      // :12-28
    acc_rho = (acc && acc->rho > 0) ? acc->rho : 2.0;
    acc_theta = (acc && acc->theta > 0) ? acc->theta : 2.0;
    acc_halpern = (acc_theta == 2.0);                                   // :30-34
    acc_k = 0;                                                          // :158
    acc_gather_valid = false;
    acc_swapped = false;
    if (const char *e = getenv("DOTSOCP_ACC_POST")) acc_post = (atoi(e) != 0);
    if (acc_halpern) DS_CHECK(acc_set_anchors());                       // :161-163 (after alpha, beta, c /= sigma)
    return 0;
}

// state <-> x^+ : the KKT block, the outputs after a stop and the sigma update address x^+ as "the iterates"
void Solver::acc_swap_state() {
    for (auto &s : slabs) {
        std::swap(s.phi, s.phi_p);
        std::swap(s.q, s.q_old);
        std::swap(s.alpha, s.alpha_p);
        std::swap(s.z, s.z_p);
        std::swap(s.beta, s.beta2);
    }
    acc_swapped = !acc_swapped;
}

// :346-358, called from kkt_block() with the pointers swapped: scale_state() has divided alpha^+, beta^+, c;
// here the previous iterates follow and the extrapolation restarts from x^+
int Solver::acc_on_sigma_factor(double factor) {
    acc_factor = factor;
    acc_k = 0;
    // acc_light: the previous iterates, the anchors and the extrapolation take one pass per array after the block
    // (k_acc_restart, k_acc_cone<ACC_RESTART>), which divides on the way
    if (acc_light) return 0;
    FOR_SLABS(s) {
        DS_CHECK(launch_scale(s.alpha_p, s.g.NqAlloc, 1.0, factor, s.st));
        DS_CHECK(launch_scale(s.beta2, 10 * s.g.Nc, 1.0, factor, s.st));
    }
    if (acc_halpern) DS_CHECK(acc_set_anchors());
    return 0;
}

// :167-225
int Solver::acc_rescale_block() {
    bool scaleYes = false;
    double normPhis = 0, normAlps = 0;
    auto norms = [&](double &nPhis, double &nAlps) -> int {
        double S[S_COUNT + 1];
        if (multi()) {
            // the sums are taken at the extrapolated state, whose phi / q halos nobody maintains (the q-step folds the
            // extrapolation of q over owned entries only, phi is extrapolated over owned nodes)
            prof_begin(PH_COMM);
            DS_CHECK(shift(-1, [](Slab &s) { return s.phi; }, [](Slab &s) { return s.phi + s.g.plane * s.g.ntl; }, slabs[0].g.plane));
            prof_end(PH_COMM);
            DS_CHECK(exchange_q_halo(false));
        }
        DS_CHECK(kkt_sums(S));
        const double sh = sqrt(h);
        const double normPhi = sh * sqrt(S[S_PHI2]), normQ = sh * sqrt(S[S_Q2]), normZ = sh * sqrt(S[S_Z2]);
        const double normAlpha = sigma * (sh * sqrt(S[S_ALPHA2])), normBeta = sigma * (sh * sqrt(S[S_BETA2]));
        nPhis = std::max(std::max(normPhi, normQ), normZ);
        nAlps = std::max(normAlpha, normBeta);
        return 0;
    };
    if (rescale >= 3 && (it % 200) == 0) {                               // checkRescaleIters = 200 (:96)
        DS_CHECK(norms(normPhis, normAlps));
        const double ratio = std::max(normAlps, normPhis) / std::min(normAlps, normPhis);
        if (ratio > 1.2) scaleYes = true;
    }
    const bool first = (rescale == 1) && (maxFeas < 2e-2) && (it >= 10) && (relGap < 5e-2);
    const bool second = (rescale == 2) && (maxFeas < 5e-3) && (it >= 50) && (relGap < 1e-2);
    if (!(first || second || scaleYes)) return 0;
    if (!scaleYes) DS_CHECK(norms(normPhis, normAlps));
    const double dScale2 = normPhis, cScale2 = normAlps;
    sigma = sigma * (cScale2 / dScale2);
    norm_c = norm_c / cScale2;
    if (!prob.weighted) norm_d = norm_d / dScale2;                        // solver_wsocp_accADMM.m has no norm_d
    DS_CHECK(scale_state(dScale2, cScale2 * cScale2, dScale2, true));    // c, alpha, beta, q, z
    FOR_SLABS(s) DS_CHECK(launch_scale(s.phi, s.g.NphiAlloc, 1.0, dScale2, s.st));   // :207
    dScale = dScale2 * dScale;
    cScale = cScale2 * cScale;
    sigmaScale = sigmaScale * (cScale2 / dScale2);
    update_coef();
    acc_k = 0;                                                            // :217-222
    if (acc_halpern) DS_CHECK(acc_set_anchors());
    acc_gather_valid = false;
    rescale += 1;
    return 0;
}

AccCoef Solver::acc_coef() const {
    AccCoef k{};
    const double kk = (double)acc_k;
    k.rho = acc_rho;
    k.om_rho = 1.0 - acc_rho;
    if (acc_halpern) {
        k.c1 = 1.0 / (kk + 2.0);                                          // :373-374
        k.c2 = (kk + 1.0) / (kk + 2.0);
    } else {
        k.c1 = acc_theta / (2.0 * (kk + acc_theta));                      // :397,404-405
        k.c2 = kk / (kk + acc_theta);
        k.om_c1 = 1.0 - k.c1;
        k.c1c2 = k.c1 + k.c2;
    }
    return k;
}

int Solver::acc_step(bool *brk) {
    *brk = false;
    it += 1;
    DS_CHECK(acc_rescale_block());
    const bool adjustSigmaYes = if_adjust_sigma((double)it, lastSigmaIt);                   // :253
    // one slab per process: time-outs are detected at KKT checks only, from the agreed clock (see Solver::kkt_block)
    const bool timed_out = remote() ? false : (elapsed() > time_limit);
    // the time-limit term of :254 is evaluated before the q-step here (the device queue is asynchronous)
    const bool kkt_due = opts.ifCheckStepByStep || adjustSigmaYes || it == opts.maxit || timed_out;
    const AccCoef kc = acc_coef();
    const bool fold = acc_halpern && !kkt_due;      // extrapolation of z, beta inside the cone pass
    // an iteration with a KKT check: x^+ is stored for the block; afterwards z and beta are extrapolated by a second cone pass
    // that recomputes x^+ from the (untouched) state, applies the block's sigma factor and emits the next gather
    const bool post = acc_halpern && kkt_due && acc_post;
    // one slab: the KKT sums that need z^+ and beta^+ are taken by the cone pass itself, which then runs AFTER the phi-step
    // (it does not depend on phi^+, the sums do): nothing of the block reads z^+ or beta^+ from memory again
    const bool kfold = kkt_due && kkt_fold && !multi();

    // ---- step q (:227-232) ----
    if (!acc_gather_valid) {
        prof_begin(PH_ACC_GATHER);
        FOR_SLABS(s) {
            AccArgs a{};
            a.z_in = s.z; a.beta_in = s.beta;
            a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy;
            DS_CHECK(launch_acc_cone(2, s.g, lc, s.fg, a, s.st));
        }
        prof_end(PH_ACC_GATHER);
    }
    // time slabs: adjoint tails of (z + beta) -> right, head of the extrapolated phi -> left
    DS_CHECK(ship_tails());
    // q^+ (raw, for the cone pass and the KKT block) always goes to q_old and the right-hand side of the phi-step
    // to w0; without a KKT check the Halpern step of q (in place) and alpha (ping-pong) is part of the same pass
    prof_begin(PH_QSTEP);
    FOR_SLABS(s) {
        if (fold) {
            DS_CHECK(launch_qstep_rhs_acc(2, s.g, lc, s.fg, s.phi, s.q2, s.sx, s.sy, s.weight, s.c, s.q_old, s.alpha,
                                          s.alpha2, s.w0, s.q, s.q_a, s.alpha_a, kc, s.st, s.tail_bx, s.tail_by,
                                          s.g.last ? nullptr : s.send_plane));
            std::swap(s.alpha, s.alpha2);
        } else {
            DS_CHECK(launch_qstep_rhs_acc(1, s.g, lc, s.fg, s.phi, s.q2, s.sx, s.sy, s.weight, s.c, s.q_old, s.alpha,
                                          s.alpha_p, s.w0, nullptr, nullptr, nullptr, kc, s.st, s.tail_bx, s.tail_by,
                                          s.g.last ? nullptr : s.send_plane));
        }
    }
    prof_end(PH_QSTEP);
    if (multi()) {
        // raw q^+ halo -> left (cone pass of the last cell layer, KKT block), raw u0 tail -> right (first rhs layer)
        prof_begin(PH_COMM);
        DS_CHECK(group_begin());
        DS_CHECK(shift_edge_halo([](Slab &s) { return s.q_old; }));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_plane; }, [](Slab &s) { return s.u0_prev; }, slabs[0].g.plane));
        DS_CHECK(group_end());
        FOR_SLABS(s)
            if (!s.g.first) DS_CHECK(launch_rhs_fixup(s.g, lc, s.u0_prev, s.w0, s.st));
        prof_end(PH_COMM);
    }

    // ---- multipliers + step z (:234-239,246-249); the cone pass does not need phi^+ ----
    auto cone_pass = [&]() -> int {
    prof_begin(kfold ? PH_FUSED_A : PH_ACC_CONE);      // the KKT flavour is timed apart (bench.py prices acc_cone as mode 1)
    if (kfold) {
        FOR_SLABS(s) DS_HIP(ds_memset_async(s.kw.partials, 0, sizeof(double) * s.kw.maxBlocks * S_COUNT, s.st));
    }
    FOR_SLABS(s) {
        AccArgs a{};
        a.q = s.q_old;
        a.z_in = s.z; a.beta_in = s.beta;
        a.z_out = s.z_p; a.beta_out = s.beta2;
        if (fold) {
            a.z0 = s.z_a; a.beta0 = s.beta_a;
            a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy;
            a.c1 = kc.c1; a.c2 = kc.c2; a.om_rho = kc.om_rho; a.rho = kc.rho;
        }
        if (kfold) {
            a.kk = kkt_coef();
            a.alpha_p = s.alpha_p; a.weight = s.weight;
            a.nostore = post ? 1 : 0;       // x^+ is recomputed by the pass after the block; stored only for a stop (below)
            a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy;          // scratch: this iteration's q-step has consumed the gather
            DS_CHECK(launch_acc_cone_kkt(s.g, lc, s.fg, a, s.kw, s.st));
        } else {
            DS_CHECK(launch_acc_cone(fold ? 1 : 0, s.g, lc, s.fg, a, s.st));
        }
    }
    prof_end(kfold ? PH_FUSED_A : PH_ACC_CONE);
    return 0;
    };
    if (!kfold) DS_CHECK(cone_pass());

    // ---- step phi (:241-244) ----
    prof_begin(PH_POISSON);
    for (auto &s : slabs) std::swap(s.phi, s.phi_p);      // poisson_all() writes s.phi
    int rc = poisson_all();
    for (auto &s : slabs) std::swap(s.phi, s.phi_p);
    DS_CHECK(rc);
    prof_end(PH_POISSON);
    if (kfold) DS_CHECK(cone_pass());

    // ---- KKT (:251-367) at x^+ ----
    if (kkt_due) {
        if (multi()) {     // A phi^+ of a slab's last cell layer reads the right neighbour's first phi^+ layer
            prof_begin(PH_COMM);
            DS_CHECK(shift(-1, [](Slab &s) { return s.phi_p; }, [](Slab &s) { return s.phi_p + s.g.plane * s.g.ntl; }, slabs[0].g.plane));
            prof_end(PH_COMM);
        }
        acc_swap_state();
        acc_light = post;
        acc_factor = 1.0;
        rc = kkt_block(adjustSigmaYes, timed_out, brk, kfold);
        acc_light = false;
        DS_CHECK(rc);
        if (*brk) {                                        // :322-325: the outputs are x^+ (pointers stay swapped)
            if (kfold && post) {                           // ... whose z^+, beta^+ the folded cone pass did not store
                acc_swap_state();
                FOR_SLABS(s) {
                    AccArgs a{};
                    a.q = s.q_old;
                    a.z_in = s.z; a.beta_in = s.beta;
                    a.z_out = s.z_p; a.beta_out = s.beta2;
                    DS_CHECK(launch_acc_cone(0, s.g, lc, s.fg, a, s.st));
                }
                acc_swap_state();
            }
            return 0;
        }
        acc_swap_state();
    }

    // ---- interpolation (:369-423) ----
    prof_begin(PH_INTERP);
    const AccCoef k2 = acc_coef();                         // k may have been reset by the sigma update
    const int mode = acc_halpern ? 0 : (acc_k == 0 ? 1 : 2);
    const int write_aux = (!acc_halpern && acc_k + 1 < acc_restart) ? 1 : 0;      // :417-421
    FOR_SLABS(s) {
        const Grid &g = s.g;
        const bool restart = post && acc_factor != 1.0;
        if (restart) {
            DS_CHECK(launch_acc_restart(s.phi, s.phi_p, s.phi_a, g.Nphi, k2, 1.0, s.st));
            DS_CHECK(launch_acc_restart(s.q, s.q_old, s.q_a, g.NqAlloc, k2, 1.0, s.st));
            DS_CHECK(launch_acc_restart(s.alpha, s.alpha_p, s.alpha_a, g.NqAlloc, k2, acc_factor, s.st));
        } else {
            DS_CHECK(launch_acc_interp(s.phi, s.phi_p, s.phi_a, g.Nphi, k2, mode, write_aux, s.st));
        }
        if (!fold && !restart) {
            DS_CHECK(launch_acc_interp(s.q, s.q_old, s.q_a, g.NqAlloc, k2, mode, write_aux, s.st));
            DS_CHECK(launch_acc_interp(s.alpha, s.alpha_p, s.alpha_a, g.NqAlloc, k2, mode, write_aux, s.st));
        }
        if (post) {
            AccArgs a{};
            a.q = s.q_old;                                 // raw q^+ (the sigma update does not touch q)
            a.z_in = s.z; a.beta_in = s.beta;
            a.z_out = s.z_p; a.beta_out = s.beta2;         // over the stored x^+, which is not needed any more
            a.z0 = s.z_a; a.beta0 = s.beta_a;
            a.z0_out = s.z_a; a.beta0_out = s.beta_a;
            a.bdiv = acc_factor;
            a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy;
            a.c1 = k2.c1; a.c2 = k2.c2; a.om_rho = k2.om_rho; a.rho = k2.rho;
            DS_CHECK(launch_acc_cone(acc_factor != 1.0 ? 3 : 1, s.g, lc, s.fg, a, s.st));
        }
        if (fold || post) {
            std::swap(s.z, s.z_p);                         // the cone pass wrote the new state there
            std::swap(s.beta, s.beta2);
        } else {
            DS_CHECK(launch_acc_interp(s.z, s.z_p, s.z_a, 10 * g.Nc, k2, mode, write_aux, s.st));
            DS_CHECK(launch_acc_interp(s.beta, s.beta2, s.beta_a, 10 * g.Nc, k2, mode, write_aux, s.st));
        }
    }
    acc_gather_valid = fold || post;
    prof_end(PH_INTERP);
    acc_k += 1;                                             // :381,413
    if (acc_k >= acc_restart) {                             // :385-388,417-418
        acc_k = 0;
        if (acc_halpern) DS_CHECK(acc_set_anchors());
    }
    return 0;
}

}  // namespace dotsocp
