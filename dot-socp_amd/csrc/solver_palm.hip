// Host side of the device-resident proximal ALM loop: socp/dot2d/algorithms/solver_socp_PALM.m (2-D,
// unweighted).  PALM = inPALM with (a) z0 = BF (A phi0) + d (:136-138), (b) an extra q-step in front of the
// phi-step (:196-200) whose result q~ feeds the phi-step and the projection (:204,209-210), and (c) a
// rescale block that scales tmp_q = A phi instead of q (:181-191).  Rescale and KKT blocks are the inPALM
// ones (Solver::rescale_block / kkt_block).
//
// Device dataflow (fused, z never stored), iteration k, entering with q^k in s.q, q~^{k-1} in s.q_old and
// the multiplier step of iteration k-1 still pending (beta^{k-1} in memory):
//   pass 1  k_cone_fused<4>: z^k = Pi(BF q~^{k-1} + d - beta^{k-1}) recomputed, beta^k stored,
//           q2 = F*B*(z^k + beta^k)                                                        (:198, :224-227 of k-1)
//   q~^k    = (A phi^k + alpha^k + q2) .* diagQInv            -> s.q_old                   (:199)
//   phi^{k+1} from q~^k, alpha^k                                                           (:204)
//   pass 2  k_cone_fused<0>: z^{k+1} = Pi(BF q~^k + d - beta^k), q2 = F*B*(z^{k+1} + beta^k) (:209-210,216)
//   q^{k+1}, alpha^{k+1}                                      -> s.q                       (:215-217,221,225)
//   beta^{k+1}: deferred to pass 1 of the next iteration (or to the KKT block)              (:222-226)
//
// palm_fast: ONE pass over beta per iteration.  F*B*BF is diagonal and F*B*d = 0, so the gather of pass 1,
//   F*B*(z^k + beta^k) = F*B*((1 + tau) z^k + beta^{k-1}) - tau F*B*(BF q^k + d),
// needs no pass over beta: the cone pass of iteration k-1 emits p2 = F*B*((1 + tau) z^k + beta^{k-1}) as a second gather
// beside the q2 of its own q-step (k_cone_fused modes 5 / 6), and the first q-step of iteration k subtracts the
// element-wise term (k_qstep_rhs VAR 3 with qk).  The multiplier step then waits for the cone pass of iteration k, which
// recomputes z^k from q~^{k-1} (kept: q~ alternates between q_old and q3), updates beta with q^k and projects at q~^k:
//   pass    k_cone_fused<5>: z^k = Pi(BF q~^{k-1} + d - beta^{k-1}), beta^k = beta^{k-1} + tau (z^k - BF q^k - d) stored,
//           z^{k+1} = Pi(BF q~^k + d - beta^k), q2 = F*B*(z^{k+1} + beta^k), p2 = F*B*((1 + tau) z^{k+1} + beta^k)
// Right after a KKT or rescale block (multiplier step executed, z regenerated on demand) the iteration runs as above with
// mode 6 (= mode 0 + p2) as its pass 2.
// Time slabs (round 4): the same dataflow.  The second gather's share of the right neighbour's first edge layer travels
// with the first gather's (one more pair of tail layers in the group behind the cone pass), q~^k lives in q3 when the pass
// is mode 5 -- its halo and the u0 tail are taken from there --, and the element-wise term needs no neighbour.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "solver.h"

namespace dotsocp {

int Solver::palm_begin() {
    if (!slabs[0].q2) {
        set_error("PALM needs the fused dataflow (unset DOTSOCP_FUSED=0)");
        return DOTSOCP_EINVAL;
    }
    // :136-138  tmp_q = A phi; z = BF tmp_q + d (boundary slots keep their uploaded values, like mexBFd)
    if (multi()) {     // the forward time difference of a slab's last cell layer reads the right neighbour's first phi layer
        prof_begin(PH_COMM);
        DS_CHECK(shift(-1, [](Slab &s) { return s.phi; }, [](Slab &s) { return s.phi + s.g.plane * s.g.ntl; }, slabs[0].g.plane));
        prof_end(PH_COMM);
    }
    FOR_SLABS(s) DS_CHECK(launch_grad(s.g, lc, s.phi, s.q2, s.st));
    if (multi()) {     // ... and BF of that layer the neighbour's first bx / by layers of tmp_q
        prof_begin(PH_COMM);
        DS_CHECK(group_begin());
        DS_CHECK(shift_edge_halo([](Slab &s) { return s.q2; }));
        DS_CHECK(group_end());
        prof_end(PH_COMM);
    }
    FOR_SLABS(s) DS_CHECK(launch_bfd(s.g, s.z, s.q2, lc.s, lc.dF, s.st));
    deferred = false;
    z_valid = true;
    if (const char *e = getenv("DOTSOCP_PALM_FAST")) palm_fast = (atoi(e) != 0);
    palm_p_valid = false;
    if (palm_fast) {
        FOR_SLABS(s) {
            if (s.q3) continue;
            DS_CHECK(dzalloc(&s.q3, s.g.NqAlloc, s.st));
            DS_CHECK(dzalloc(&s.p2, s.g.NqAlloc, s.st));
            DS_CHECK(dzalloc(&s.sxp, s.fg.sx_len, s.st));
            DS_CHECK(dzalloc(&s.syp, s.fg.sy_len, s.st));
            if (multi() && !s.g.first) {
                DS_CHECK(dzalloc(&s.ptail_bx, s.g.bxLayer, s.st));
                DS_CHECK(dzalloc(&s.ptail_by, s.g.byLayer, s.st));
            }
            if (multi() && !s.g.last) {
                DS_CHECK(dzalloc(&s.send_pbx, s.g.bxLayer, s.st));
                DS_CHECK(dzalloc(&s.send_pby, s.g.byLayer, s.st));
            }
        }
    }
    return 0;
}

int Solver::palm_step(bool *brk) {
    *brk = false;
    it += 1;
    DS_CHECK(rescale_block());            // :142-194 (scales phi in place of tmp_q for this method)
    // ---- first q-step :196-200 ----
    prof_begin(PH_QSTEP0);
    const bool fast = palm_fast;
    // the gather comes from the last cone pass's second output; the multiplier step stays pending until this iteration's pass
    const bool three = fast && deferred && palm_p_valid;
    if (three) {
        FOR_SLABS(s)
            DS_CHECK(launch_qstep_palm_first(s.g, lc, s.fg, s.phi, s.p2, s.sxp, s.syp, s.c, s.q3, s.alpha, s.w0, s.st, s.ptail_bx,
                                             s.ptail_by, s.q));      // (time slabs: the tails came with the last cone pass's)
    } else if (deferred) {
        FOR_SLABS(s) {
            FusedArgs a{};
            a.q_old = s.q_old; a.q = s.q;
            a.beta_in = s.beta; a.beta_out = s.beta2;
            a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy;
            set_pending(a);
            DS_CHECK(launch_cone_fused(4, s.g, lc, s.fg, a, s.st));
            std::swap(s.beta, s.beta2);
        }
        bpend = 0;
        deferred = false;
    } else {
        DS_CHECK(ensure_z());             // first iteration, or right after a KKT / rescale block
        DS_CHECK(flush_beta());
        FOR_SLABS(s) {
            AccArgs a{};
            a.z_in = s.z; a.beta_in = s.beta;
            a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy;
            DS_CHECK(launch_acc_cone(2, s.g, lc, s.fg, a, s.st));
        }
    }
    if (!three) {
        DS_CHECK(phase_z_tails());        // time slabs: adjoint tails -> right (the phi head travelled in the last iteration)
        FOR_SLABS(s)
            DS_CHECK(launch_qstep_palm_first(s.g, lc, s.fg, s.phi, s.q2, s.sx, s.sy, s.c, s.q_old, s.alpha, s.w0, s.st,
                                             s.tail_bx, s.tail_by));
    }
    prof_end(PH_QSTEP0);
    if (multi()) {
        // q~ halo -> left (projection of the last cell layer), u0 = q~0 - alpha0 of the last cell -> right (first rhs layer)
        prof_begin(PH_COMM);
        FOR_SLABS(s)
            if (!s.g.last) DS_CHECK(launch_u0_tail(s.g, three ? s.q3 : s.q_old, s.alpha, nullptr, s.send_plane, s.st));
        DS_CHECK(group_begin());
        if (three) DS_CHECK(shift_edge_halo([](Slab &s) { return s.q3; }));
        else DS_CHECK(shift_edge_halo([](Slab &s) { return s.q_old; }));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_plane; }, [](Slab &s) { return s.u0_prev; }, slabs[0].g.plane));
        DS_CHECK(group_end());
        FOR_SLABS(s)
            if (!s.g.first) DS_CHECK(launch_rhs_fixup(s.g, lc, s.u0_prev, s.w0, s.st));
        prof_end(PH_COMM);
    }
    // ---- step phi :202-205 (its right-hand side was formed by the q-step above) ----
    prof_begin(PH_POISSON);
    DS_CHECK(poisson_all());
    prof_end(PH_POISSON);
    // ---- step z :207-211 (+ the adjoint sums of :216) ----
    prof_begin(three ? PH_FUSED_B : PH_FUSED_A);
    FOR_SLABS(s) {
        FusedArgs a{};
        a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy;
        a.p2 = s.p2; a.sxp = s.sxp; a.syp = s.syp;
        if (three) {
            a.q_old = s.q_old;            // q~^{k-1}: z^k again
            a.q = s.q;                    // q^k: the multiplier step
            a.q3 = s.q3;                  // q~^k: the projection
            a.beta_in = s.beta; a.beta_out = s.beta2;
            set_pending(a);
            DS_CHECK(launch_cone_fused(5, s.g, lc, s.fg, a, s.st));
            std::swap(s.beta, s.beta2);
            std::swap(s.q_old, s.q3);     // q_old = q~^k again, as the KKT / rescale blocks and the next pass expect
        } else {
            a.q = s.q_old;
            a.beta_in = s.beta;
            DS_CHECK(launch_cone_fused(fast ? 6 : 0, s.g, lc, s.fg, a, s.st));
        }
    }
    if (three) bpend = 0;
    palm_p_valid = fast;
    prof_end(three ? PH_FUSED_B : PH_FUSED_A);
    z_valid = false;
    z_prev_ok = false;
    if (multi()) {
        // time slabs: phi^{k+1} head -> left, adjoint tails of both gathers -> right, one group
        DS_CHECK(make_tails());
        if (fast) {
            FOR_SLABS(s)
                if (!s.g.last) DS_CHECK(launch_tail_finalize(s.g, lc, s.fg, s.p2, s.sxp, s.syp, s.send_pbx, s.send_pby, s.st));
        }
        DS_CHECK(group_begin());
        DS_CHECK(send_phi_head());
        DS_CHECK(send_tails());
        if (fast) {
            DS_CHECK(shift(+1, [](Slab &s) { return s.send_pbx; }, [](Slab &s) { return s.ptail_bx; }, slabs[0].g.bxLayer));
            DS_CHECK(shift(+1, [](Slab &s) { return s.send_pby; }, [](Slab &s) { return s.ptail_by; }, slabs[0].g.byLayer));
        }
        DS_CHECK(group_end());
    }
    // ---- second q-step + alpha :213-218,221,225 ----
    // Whether the iteration ends with a KKT check (:231-232) is known here (the time limit is the one trigger that is
    // not: such a check takes the unfolded block).  If it does, the q-step runs in the flavour that accumulates its share
    // of the KKT sums -- everything made of phi^{k+1}, q^{k+1}, alpha^{k+1}, A phi and c -- as in the inPALM loop
    // (Solver::phase_q; the right-hand side it also writes is not used: PALM's first q-step forms its own), and the
    // cell pass of the block then carries the pending multiplier step, the cell sums and F*B*beta (Solver::kkt_sums).
    const bool adjustSigmaYes = if_adjust_sigma((double)it, lastSigmaIt);                 // :231
    const bool kkt_due = opts.ifCheckStepByStep || adjustSigmaYes || it == opts.maxit;
    const bool fold = kkt_due && kkt_fold && qrhs;
    prof_begin(PH_QSTEP);
    if (fold) {
        const KktCoef k = kkt_coef();
        FOR_SLABS(s) {
            DS_HIP(ds_memset_async(s.kw.partials, 0, sizeof(double) * s.kw.maxBlocks * S_COUNT, s.st));
            QStepExtra ex{};
            ex.apend = 0; ex.amul = 1.0; ex.adiv = 1.0;
            ex.partials = kkt_qstep_partials(s.g, s.kw);
            ex.resid = s.w1;                       // free between the Poisson solves
            ex.kappa = k.kappa; ex.dsD = k.dsD;
            DS_CHECK(launch_qstep_rhs(s.g, lc, s.fg, s.phi, s.q2, s.sx, s.sy, nullptr, s.tail_bx, s.tail_by, s.c, s.q, s.alpha,
                                      s.alpha2, s.w0, s.st, 0, -1, 1, &ex));
            std::swap(s.alpha, s.alpha2);
        }
    } else {
        FOR_SLABS(s)
            DS_CHECK(launch_qstep_fused(s.g, lc, s.fg, s.phi, s.q2, s.sx, s.sy, nullptr, s.tail_bx, s.tail_by, s.q, s.alpha,
                                        s.st));
    }
    prof_end(PH_QSTEP);
    DS_CHECK(exchange_q_halo(false));     // time slabs: q^{k+1} halo (multiplier step, KKT block)
    deferred = true;                       // beta^{k+1}: :222-226, executed by the next pass over beta
    const bool timed_out = elapsed() > time_limit;
    if (kkt_due || timed_out)                                                             // :232
        DS_CHECK(kkt_block(adjustSigmaYes, timed_out, brk, fold));
    return 0;
}

}  // namespace dotsocp
