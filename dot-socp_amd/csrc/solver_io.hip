// Device-side driver steps around the loop: the multilevel transfer
// jump_nextLevel (+ recoverOrgVar of the coarse level, InitialScaling of the fine one) and the outputs
// recover_RhoE / recover_q.  Kernels: transfer.hip.  The outputs work on time slabs; the level transfer is one-slab.
#include <cstring>

#include "solver.h"

namespace dotsocp {

// solver_dotsocp2d.m:262-281 on the device; call after finish().  Any output pointer may be NULL.  Time slabs: every
// slab produces its own layers (the density at its first node needs the left neighbour's last cell: one ny x nx layer
// to the right); in-process slabs fill the global host arrays, an RCCL rank its own slab of them.
int Solver::recover_outputs(const double *rho0, const double *rho1, double *rho, double *Ex, double *Ey, double *q0,
                            double *bx, double *by) {
    if (!finished) { set_error("recover_outputs() needs finish()"); return DOTSOCP_ESTATE; }
    DS_ARG(rho == nullptr || (rho0 != nullptr && rho1 != nullptr), "rho needs rho0 and rho1");
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    const i64 plane = ny * nx;
    const double cD = cScale * D, dD = dScale / D;         // recoverOrgVar (solver_dotsocp2d.m:368-386)
    if (rho && multi()) {
        FOR_SLABS(s)
            if (!s.g.last) DS_CHECK(launch_out_tail(s.g, s.alpha, s.weight, sigma, cD, s.send_plane, s.st));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_plane; }, [](Slab &s) { return s.a0_prev; }, plane));
    }
    double *outs[6] = {rho, Ex, Ey, q0, bx, by};
    for (int which = 0; which < 6; ++which) {
        if (!outs[which]) continue;
        FOR_SLABS(s) {
            const Grid &g = s.g;
            double *d_r0 = s.w1, *d_r1 = s.w1 + plane;     // w1 holds at least two layers (nt >= 2 per slab)
            if (which == 0) {
                if (g.first) DS_HIP(hipMemcpyAsync(d_r0, rho0, sizeof(double) * plane, hipMemcpyHostToDevice, s.st));
                if (g.last) DS_HIP(hipMemcpyAsync(d_r1, rho1, sizeof(double) * plane, hipMemcpyHostToDevice, s.st));
            }
            const i64 layers = (which >= 3) ? g.ncl : g.ntl;
            DS_CHECK(launch_outputs(g, s.q, s.alpha, s.weight, d_r0, d_r1, s.a0_prev, sigma, cD, dD, which, s.w0, s.st));
            double *h = outs[which] + (remote() ? 0 : plane * g.t0);
            if (layers > 0)
                DS_HIP(hipMemcpyAsync(h, s.w0, sizeof(double) * (size_t)(plane * layers), hipMemcpyDeviceToHost, s.st));
        }
        DS_CHECK(sync_all());                               // w0 is reused by the next output
    }
    return 0;
}

// jump_nextLevel.m:5-16: this = fine level (created, c / weight uploaded, begin() not yet called),
// `coarse` = the finished coarse level on the same device.
int Solver::jump_from(Solver &coarse) {
    if (begun) { set_error("jump_next_level() must precede begin() of the fine level"); return DOTSOCP_ESTATE; }
    if (!coarse.finished) { set_error("jump_next_level() needs finish() of the coarse level"); return DOTSOCP_ESTATE; }
    DS_ARG(!multi() && !remote() && !coarse.multi() && !coarse.remote(), "multilevel transfer runs on one slab");
    DS_ARG(device == coarse.device, "both levels must live on the same device");
    DS_ARG(prob.dim == coarse.prob.dim && prob.weighted == coarse.prob.weighted, "level kinds differ");
    DS_ARG(ny == 2 * (coarse.ny - 1) + 1 && nt == 2 * (coarse.nt - 1) + 1 &&
               (nx == 2 * (coarse.nx - 1) + 1 || (nx == 1 && coarse.nx == 1)),
           "fine grid must be 2 (n - 1) + 1 of the coarse grid in every dimension");
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    DS_CHECK(ensure_alloc());
    Slab &f = slabs[0];
    Slab &c = coarse.slabs[0];
    // everything of the coarse level has to be complete before this level's stream reads it
    DS_CHECK(coarse.sync_all());
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    cScale = prob.cScale; dScale = prob.dScale; D = prob.D; E = prob.E;
    update_coef();
    // phi: dScale_c * phi_c (recoverOrgVar) -> interpolate -> (1/dScale_f) * (InitialScaling)
    DS_CHECK(launch_prolong_phi(f.g, c.g, c.phi, f.phi, coarse.dScale, 1.0 / dScale, stream));
    // beta: (cScale_c E_c) * (sigma_c * beta_c) -> interpolate -> (1/cScale_f/E_f) * ; the unscaled -betaR goes to z
    DS_CHECK(launch_prolong_beta(f.g, c.g, c.beta, f.beta, f.z, coarse.sigma, coarse.cScale * coarse.E,
                                 1.0 / cScale / E, stream));
    // alpha = F*B*(-betaR) ./ w, times 1/cScale_f/D_f
    DS_CHECK(launch_bfd_conj(f.g, f.alpha, f.z, 1.0, stream));
    DS_CHECK(launch_scale_div(f.alpha, f.weight, f.g.NqAlloc, 1.0 / cScale / D, stream));
    // q = (D_f/dScale_f) * grad(phiR) ./ w : the D_f / h factors are those of the loop's own stencil
    DS_CHECK(launch_grad(f.g, lc, f.phi, f.q, stream));
    if (f.weight) DS_CHECK(launch_scale_div(f.q, f.weight, f.g.NqAlloc, 1.0, stream));
    DS_HIP(hipMemsetAsync(f.z, 0, sizeof(double) * 10 * f.g.Nz, stream));       // var.z of initialize.m
    DS_HIP(hipStreamSynchronize(stream));
    return 0;
}

}  // namespace dotsocp
