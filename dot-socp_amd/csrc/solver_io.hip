// Device-side driver steps around the loop: the multilevel transfer
// jump_nextLevel (+ recoverOrgVar of the coarse level, InitialScaling of the fine one) and the outputs
// recover_RhoE / recover_q.  Kernels: transfer.hip.  Both work on in-process time slabs (dotsocp_create_multi).
#include <algorithm>
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
    const i64 plane = slabs[0].g.plane;                    // device layers (pitched rows, common.h)
    const i64 hplane = ny * nx;                            // host layers (reference layout)
    const i64 py = slabs[0].g.py;
    const double cD = cScale * D, dD = dScale / D;         // recoverOrgVar (solver_dotsocp2d.m:368-386)
    if (rho && multi()) {
        FOR_SLABS(s)
            if (!s.g.last) DS_CHECK(launch_out_tail(s.g, s.alpha, s.weight, sigma, cD, s.send_plane, s.st));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_plane; }, [](Slab &s) { return s.a0_prev; }, plane));
    }
    double *outs[6] = {rho, Ex, Ey, q0, bx, by};
    for (int which = 0; which < 6; ++which) {
        if (!outs[which]) continue;
        {   // the host array of this output (this process's layers of it): make its pages exist before the copies
            i64 layers = 0;
            for (auto &s : slabs) layers += (which >= 3) ? s.g.ncl : s.g.ntl;
            host_first_touch(outs[which], sizeof(double) * (size_t)(hplane * layers));
        }
        FOR_SLABS(s) {
            const Grid &g = s.g;
            double *d_r0 = s.w1, *d_r1 = s.w1 + plane;     // w1 holds at least two layers (nt >= 2 per slab)
            if (which == 0) {
                if (g.first) DS_CHECK(copy_rows(d_r0, const_cast<double *>(rho0), ny, py, nx, true, s.st));
                if (g.last) DS_CHECK(copy_rows(d_r1, const_cast<double *>(rho1), ny, py, nx, true, s.st));
            }
            const i64 layers = (which >= 3) ? g.ncl : g.ntl;
            DS_CHECK(launch_outputs(g, s.q, s.alpha, s.weight, d_r0, d_r1, s.a0_prev, sigma, cD, dD, which, s.w0, s.st));
            double *h = outs[which] + (remote() ? 0 : hplane * g.t0);
            if (layers > 0) DS_CHECK(copy_rows(s.w0, h, ny, py, nx * layers, false, s.st));
        }
        DS_CHECK(sync_all());                               // w0 is reused by the next output
    }
    return 0;
}

// jump_nextLevel.m:5-16: this = fine level (created, c / weight uploaded, begin() not yet called),
// `coarse` = the finished coarse level.  Both may be cut into in-process time slabs (any numbers of slabs, any
// placement on devices): a fine slab interpolates from the coarse layers it needs -- nodes floor(t/2) and, for odd
// t, floor(t/2) + 1; cells floor(t/2) -- which are first gathered from the coarse slabs that own them into the fine
// slab's work arrays (peer copies between devices); F*B*(-betaR) and grad(phiR) then need one layer from the
// neighbouring fine slab each, as in the loop.  One slab per process (RCCL): not available, transfer through the host.
int Solver::jump_from(Solver &coarse) {
    if (begun) { set_error("jump_next_level() must precede begin() of the fine level"); return DOTSOCP_ESTATE; }
    if (!coarse.finished) { set_error("jump_next_level() needs finish() of the coarse level"); return DOTSOCP_ESTATE; }
    DS_ARG(!remote() && !coarse.remote(), "multilevel transfer between one-slab-per-process contexts goes through the host");
    DS_ARG(prob.dim == coarse.prob.dim && prob.weighted == coarse.prob.weighted, "level kinds differ");
    DS_ARG(ny == 2 * (coarse.ny - 1) + 1 && nt == 2 * (coarse.nt - 1) + 1 &&
               (nx == 2 * (coarse.nx - 1) + 1 || (nx == 1 && coarse.nx == 1)),
           "fine grid must be 2 (n - 1) + 1 of the coarse grid in every dimension");
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    DS_CHECK(ensure_alloc());
    // everything of the coarse level has to be complete before this level's streams read it
    coarse.cur_dev = -1;
    DS_CHECK(coarse.sync_all());
    cur_dev = -1;
    cScale = prob.cScale; dScale = prob.dScale; D = prob.D; E = prob.E;
    update_coef();
    const i64 planec = coarse.slabs[0].g.plane;           // coarse layers as they are stored (pitched rows, common.h)
    // dst (on fine slab f) <- src (on coarse slab c); the coarse level is idle, so ordering on f's stream suffices
    auto pull = [&](Slab &f, double *dst, const Slab &c, const double *src, i64 count) -> int {
        if (count <= 0) return 0;
        const size_t bytes = sizeof(double) * (size_t)count;
        if (f.dev == c.dev) DS_HIP(ds_memcpy_async(dst, src, bytes, hipMemcpyDeviceToDevice, f.st));
        else DS_HIP(ds_memcpy_peer_async(dst, f.dev, src, c.dev, bytes, f.st));
        return 0;
    };
    const bool direct = !multi() && !coarse.multi() && slabs[0].dev == coarse.slabs[0].dev;
    FOR_SLABS(f) {
        const Grid &gf = f.g;
        const Grid &gc0 = coarse.slabs[0].g;                  // ny, nx of the coarse grid
        if (direct) {
            Slab &c = coarse.slabs[0];
            // phi: dScale_c * phi_c (recoverOrgVar) -> interpolate -> (1/dScale_f) * (InitialScaling)
            DS_CHECK(launch_prolong_phi(gf, c.g, c.phi, f.phi, coarse.dScale, 1.0 / dScale, f.st));
            // beta: (cScale_c E_c) * (sigma_c * beta_c) -> interpolate -> (1/cScale_f/E_f) * ; the unscaled -betaR goes to z
            DS_CHECK(launch_prolong_beta(gf, c.g, c.beta, f.beta, f.z, coarse.sigma, coarse.cScale * coarse.E,
                                         1.0 / cScale / E, f.st));
            continue;
        }
        // coarse node layers [nA, nB] -> w0 ; coarse cell layers [cA, cB] of all ten columns -> beta2 (fused buffers) or w1
        const i64 tlast = gf.t0 + gf.ntl - 1;
        const i64 nA = gf.t0 >> 1, nB = (tlast >> 1) + (tlast & 1);
        for (auto &c : coarse.slabs) {
            const i64 a = std::max<i64>(nA, c.g.t0), b = std::min<i64>(nB, c.g.t0 + c.g.ntl - 1);
            if (a <= b) DS_CHECK(pull(f, f.w0 + planec * (a - nA), c, c.phi + planec * (a - c.g.t0), planec * (b - a + 1)));
        }
        DS_CHECK(launch_prolong_phi(gf, gc0, f.w0, f.phi, coarse.dScale, 1.0 / dScale, f.st, nA));
        if (gf.ncl > 0) {
            const i64 cA = gf.t0 >> 1, cB = (gf.t0 + gf.ncl - 1) >> 1, nl = cB - cA + 1;
            double *tmp = f.beta2;
            if (!tmp) { set_error("multilevel transfer on time slabs needs the fused dataflow"); return DOTSOCP_EINVAL; }
            for (auto &c : coarse.slabs) {
                const i64 a = std::max<i64>(cA, c.g.t0), b = std::min<i64>(cB, c.g.t0 + c.g.ncl - 1);
                if (a > b) continue;
                for (int j = 0; j < 10; ++j)
                    DS_CHECK(pull(f, tmp + (i64)j * nl * planec + planec * (a - cA), c,
                                  c.beta + (i64)j * c.g.Nc + planec * (a - c.g.t0), planec * (b - a + 1)));
            }
            DS_CHECK(launch_prolong_beta(gf, gc0, tmp, f.beta, f.z, coarse.sigma, coarse.cScale * coarse.E, 1.0 / cScale / E,
                                         f.st, cA, nl * planec));
        }
    }
    // x <- sc * x ./ w over the owned entries of a q-layout array (the halo layers of a slab are exchanged at begin())
    auto scale_owned = [&](Slab &f, double *x, double sc, bool always) -> int {
        const Grid &g = f.g;
        if (!always && !f.weight) return 0;
        const double *w = f.weight;
        DS_CHECK(launch_scale_div(x, w, g.Nz, sc, f.st));
        DS_CHECK(launch_scale_div(x + g.offBx, w ? w + g.offBx : nullptr, g.bxLayer * g.ntl, sc, f.st));
        DS_CHECK(launch_scale_div(x + g.offBy, w ? w + g.offBy : nullptr, g.byLayer * g.ntl, sc, f.st));
        return 0;
    };
    // alpha = F*B*(-betaR) ./ w, times 1/cScale_f/D_f; a slab's first edge layer needs the last cell layer of its left neighbour
    if (multi()) {
        FOR_SLABS(f)
            if (!f.g.last)
                DS_CHECK(launch_kkt_tail(f.g, f.alpha, f.z, nullptr, f.send_plane, f.send_plane2, f.send_bx, f.send_by, f.st));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_bx; }, [](Slab &s) { return s.btail_bx; }, slabs[0].g.bxLayer));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_by; }, [](Slab &s) { return s.btail_by; }, slabs[0].g.byLayer));
    }
    FOR_SLABS(f) {
        DS_CHECK(launch_bfd_conj(f.g, f.alpha, f.z, 1.0, f.st, f.btail_bx, f.btail_by));
        DS_CHECK(scale_owned(f, f.alpha, 1.0 / cScale / D, true));
    }
    // q = (D_f/dScale_f) * grad(phiR) ./ w : the D_f / h factors are those of the loop's own stencil; the forward time
    // difference of a slab's last cell layer reads the first phi layer of its right neighbour
    if (multi())
        DS_CHECK(shift(-1, [](Slab &s) { return s.phi; }, [](Slab &s) { return s.phi + s.g.plane * s.g.ntl; }, slabs[0].g.plane));
    FOR_SLABS(f) {
        DS_CHECK(launch_grad(f.g, lc, f.phi, f.q, f.st));
        DS_CHECK(scale_owned(f, f.q, 1.0, false));
        DS_HIP(ds_memset_async(f.z, 0, sizeof(double) * 10 * f.g.Nc, f.st));       // var.z of initialize.m
    }
    DS_CHECK(sync_all());
    return 0;
}

}  // namespace dotsocp
