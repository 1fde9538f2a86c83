/*
 * oracle/mex_kernels.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single-threaded like the originals) of the five MEX
 * operators on the inPALM hot path of chlhnu/DOT-SOCP.  The reference ships these
 * only as prebuilt binaries (socp/{dot1d,dot2d,wdot2d}/utils/mex*.mexa64 -- no C++
 * source in the tree); prebuilt reference binaries are never loaded or executed by
 * this repository, so the semantics restated here are the ones documented in
 * SURVEY.md section 8a (decoded from the disassembly) together with the reference's
 * own MATLAB call sites and the closed-form operators that must be consistent with
 * them:
 *   - call sites: socp/dot2d/algorithms/solver_socp_inPALM.m:133,187,199,205,212,225,240,242
 *   - diag(I + s^2 F*B*BF) must equal socp/dot2d/utils/oper_q.m:13-26 (dot1d/utils/oper_q.m:8-15)
 *   - cone geometry must reproduce socp/dot2d/utils/compute_kkt_dot_complement.m:3
 *
 * PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors for this
 * path and neither MATLAB nor the prebuilt MEX binaries can be
 * run here, so this oracle is pinned only by algebraic invariants (adjointness,
 * idempotence, diagonal identity) -- see tests/test_oracle_invariants.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 *
 * Layout (MATLAB column-major): grid ny x nx x nt, y fastest, then x, then t.
 *   q = [ q0 (ny,nx,nt-1) ; bx (ny,nx-1,nt) ; by (ny-1,nx,nt) ]
 *   z = Nz x 10 column-major (10 contiguous planes of Nz = ny*nx*(nt-1) cells)
 */
#include <math.h>
#include <stddef.h>

typedef long long i64;

/* mexProjSoc(out, in): row-wise projection of an M x K column-major matrix onto
 * the second-order cone {x1 >= ||x_{2..K}||}  (SURVEY.md 8a row a1;
 * call sites solver_socp_inPALM.m:199,240).  Three passes like the original:
 * row norms, coefficients, scale. `tmp` is caller-provided scratch of 2*M doubles. */
void oracle_proj_soc(double *out, const double *in, i64 M, i64 K, double *tmp)
{
    double *nrm = tmp, *coef = tmp + M;
    for (i64 i = 0; i < M; ++i) nrm[i] = 0.0;
    for (i64 j = 1; j < K; ++j) {
        const double *col = in + j * M;
        for (i64 i = 0; i < M; ++i) nrm[i] += col[i] * col[i];
    }
    for (i64 i = 0; i < M; ++i) {
        double n = sqrt(nrm[i]);
        double c = (in[i] / n + 1.0) * 0.5;
        if (c > 1.0) c = 1.0;          /* NaN falls through both tests */
        if (c < 0.0) c = 0.0;
        nrm[i] = n;
        coef[i] = c;
    }
    for (i64 j = 1; j < K; ++j) {
        const double *ci = in + j * M;
        double *co = out + j * M;
        for (i64 i = 0; i < M; ++i) co[i] = coef[i] * ci[i];
    }
    for (i64 i = 0; i < M; ++i) {
        double c = coef[i];
        out[i] = (c >= 1.0) ? in[i] : c * nrm[i];
    }
}

/* mexBFd(z, q, nt, nx, ny, s, dF):  z <- B F q + d   (SURVEY.md 8a row a2;
 * call sites solver_socp_inPALM.m:133,187,212,242).  Slots whose edge lies outside
 * the domain are NOT written (they keep whatever the caller's z holds: zeros()). */
void oracle_bfd(double *z, const double *q, i64 nt, i64 nx, i64 ny, double s, double dF)
{
    const i64 Nz = ny * nx * (nt - 1);
    const i64 offBx = Nz, offBy = Nz + ny * (nx - 1) * nt;
    const double sf = s / sqrt(2.0);
    for (i64 t = 0; t < nt - 1; ++t)
        for (i64 x = 0; x < nx; ++x)
            for (i64 y = 0; y < ny; ++y) {
                i64 i = y + ny * (x + nx * t);
                double q0 = q[i];
                z[i] = dF - s * q0;
                z[9 * Nz + i] = dF + s * q0;
                for (int dt = 0; dt < 2; ++dt) {
                    i64 tt = t + dt;
                    if (x >= 1)      z[(1 + 2 * dt) * Nz + i] = sf * q[offBx + y + ny * ((x - 1) + (nx - 1) * tt)];
                    if (x <= nx - 2) z[(2 + 2 * dt) * Nz + i] = sf * q[offBx + y + ny * (x + (nx - 1) * tt)];
                    if (y >= 1)      z[(5 + 2 * dt) * Nz + i] = sf * q[offBy + (y - 1) + (ny - 1) * (x + nx * tt)];
                    if (y <= ny - 2) z[(6 + 2 * dt) * Nz + i] = sf * q[offBy + y + (ny - 1) * (x + nx * tt)];
                }
            }
}

/* mexBFdConj(q, w, nt, nx, ny, s):  q <- F* B* w, the exact adjoint of oracle_bfd's
 * linear part (call sites solver_socp_inPALM.m:205,225,
 * socp/dot2d/utils/jump_nextLevel.m:16). */
void oracle_bfd_conj(double *q, const double *w, i64 nt, i64 nx, i64 ny, double s)
{
    const i64 Nz = ny * nx * (nt - 1);
    const i64 offBx = Nz, offBy = Nz + ny * (nx - 1) * nt;
    const double sf = s / sqrt(2.0);
    for (i64 i = 0; i < Nz; ++i) q[i] = s * (w[9 * Nz + i] - w[i]);
    for (i64 t = 0; t < nt; ++t)
        for (i64 xe = 0; xe < nx - 1; ++xe)
            for (i64 y = 0; y < ny; ++y) {
                double acc = 0.0;
                if (t <= nt - 2) {
                    acc += w[1 * Nz + y + ny * ((xe + 1) + nx * t)];
                    acc += w[2 * Nz + y + ny * (xe + nx * t)];
                }
                if (t >= 1) {
                    acc += w[3 * Nz + y + ny * ((xe + 1) + nx * (t - 1))];
                    acc += w[4 * Nz + y + ny * (xe + nx * (t - 1))];
                }
                q[offBx + y + ny * (xe + (nx - 1) * t)] = sf * acc;
            }
    for (i64 t = 0; t < nt; ++t)
        for (i64 x = 0; x < nx; ++x)
            for (i64 ye = 0; ye < ny - 1; ++ye) {
                double acc = 0.0;
                if (t <= nt - 2) {
                    acc += w[5 * Nz + (ye + 1) + ny * (x + nx * t)];
                    acc += w[6 * Nz + ye + ny * (x + nx * t)];
                }
                if (t >= 1) {
                    acc += w[7 * Nz + (ye + 1) + ny * (x + nx * (t - 1))];
                    acc += w[8 * Nz + ye + ny * (x + nx * (t - 1))];
                }
                q[offBy + ye + (ny - 1) * (x + nx * t)] = sf * acc;
            }
}

/* mexBFd1d(z, q, nt, nx, s, dF): 1-D version, z is Nz x 6 with Nz = nx*(nt-1),
 * q = [ q0 (nx,nt-1) ; bx (nx-1,nt) ]  (SURVEY.md 8a row a2, 1-D column list;
 * call sites socp/dot1d/algorithms/solver_socp_inPALM.m:132,186,211,241). */
void oracle_bfd1d(double *z, const double *q, i64 nt, i64 nx, double s, double dF)
{
    const i64 Nz = nx * (nt - 1), offBx = Nz;
    const double sf = s / sqrt(2.0);
    for (i64 t = 0; t < nt - 1; ++t)
        for (i64 x = 0; x < nx; ++x) {
            i64 i = x + nx * t;
            z[i] = dF - s * q[i];
            z[5 * Nz + i] = dF + s * q[i];
            for (int dt = 0; dt < 2; ++dt) {
                i64 tt = t + dt;
                if (x >= 1)      z[(1 + 2 * dt) * Nz + i] = sf * q[offBx + (x - 1) + (nx - 1) * tt];
                if (x <= nx - 2) z[(2 + 2 * dt) * Nz + i] = sf * q[offBx + x + (nx - 1) * tt];
            }
        }
}

/* mexBFdConj1d(q, w, nt, nx, s): adjoint of the 1-D operator
 * (call sites socp/dot1d/algorithms/solver_socp_inPALM.m:204,224). */
void oracle_bfd_conj1d(double *q, const double *w, i64 nt, i64 nx, double s)
{
    const i64 Nz = nx * (nt - 1), offBx = Nz;
    const double sf = s / sqrt(2.0);
    for (i64 i = 0; i < Nz; ++i) q[i] = s * (w[5 * Nz + i] - w[i]);
    for (i64 t = 0; t < nt; ++t)
        for (i64 xe = 0; xe < nx - 1; ++xe) {
            double acc = 0.0;
            if (t <= nt - 2) {
                acc += w[1 * Nz + (xe + 1) + nx * t];
                acc += w[2 * Nz + xe + nx * t];
            }
            if (t >= 1) {
                acc += w[3 * Nz + (xe + 1) + nx * (t - 1)];
                acc += w[4 * Nz + xe + nx * (t - 1)];
            }
            q[offBx + xe + (nx - 1) * t] = sf * acc;
        }
}
