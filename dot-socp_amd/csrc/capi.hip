// extern "C" surface of libdotsocp (include/dotsocp.h).
#include <cstring>
#include <new>

#include "comm.h"
#include "solver.h"

using namespace dotsocp;

namespace dotsocp {
extern thread_local std::string g_last_error;
int dotsocp_slab_range_impl(i64 nt, int world, int rank, i64 *t0, i64 *t1);
}

struct dotsocp_ctx {
    Solver s;
};

static int require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available (libdotsocp has no CPU fallback)");
        return DOTSOCP_ENODEVICE;
    }
    return 0;
}

// RAII device buffer for the host-pointer operator entry points (through the device block cache of guard.hip: a MATLAB
// loop over mexBFd / mexProjSoc gets the same buffers back call after call instead of a malloc / free pair per call)
struct DevBuf {
    double *p = nullptr;
    ~DevBuf() { dfree(p); }
    int alloc(i64 n) { return dmalloc(&p, n); }
};

extern "C" {

const char *dotsocp_last_error(void) { return g_last_error.c_str(); }
const char *dotsocp_version(void) { return "dotsocp-mi355x 0.1 (gfx950)"; }

int dotsocp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------ B2, device pointers
int dotsocp_proj_soc_dev(double *d_out, const double *d_in, dotsocp_i64 M, dotsocp_i64 K, void *stream) {
    DS_ARG(M >= 0 && K >= 2, "mexProjSoc needs an M x K matrix with K >= 2");
    DS_ARG(d_out && d_in, "NULL pointer");
    return launch_proj_soc(d_out, d_in, M, K, (hipStream_t)stream);
}

static int make_grid(Grid &g, i64 nt, i64 nx, i64 ny) {
    DS_ARG(nt >= 2 && nx >= 1 && ny >= 1, "need nt >= 2, nx >= 1, ny >= 1");
    g.set(ny, nx, nt, 0, nt);
    return 0;
}

int dotsocp_bfd_dev(double *d_z, const double *d_q, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny, double scale,
                    double dF, void *stream) {
    Grid g;
    DS_CHECK(make_grid(g, nt, nx, ny));
    DS_ARG(d_z && d_q, "NULL pointer");
    return launch_bfd(g, d_z, d_q, scale, dF, (hipStream_t)stream);
}

int dotsocp_bfd_conj_dev(double *d_q, const double *d_z, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny,
                         double scale, void *stream) {
    Grid g;
    DS_CHECK(make_grid(g, nt, nx, ny));
    DS_ARG(d_z && d_q, "NULL pointer");
    return launch_bfd_conj(g, d_q, d_z, scale, (hipStream_t)stream);
}

// ------------------------------------------------------------------ B2, host pointers
int dotsocp_proj_soc(double *out, const double *in, dotsocp_i64 M, dotsocp_i64 K) {
    DS_ARG(M >= 0 && K >= 2, "mexProjSoc needs an M x K matrix with K >= 2");
    if (M == 0) return 0;                      // empty matrix (mxGetPr may be NULL): nothing to do, like the reference
    DS_ARG(out && in, "NULL pointer");
    DS_CHECK(require_device());
    DevBuf a, b;
    DS_CHECK(a.alloc(M * K));
    DS_CHECK(b.alloc(M * K));
    DS_HIP(hipMemcpy(a.p, in, sizeof(double) * M * K, hipMemcpyHostToDevice));
    DS_CHECK(launch_proj_soc(b.p, a.p, M, K, nullptr));
    DS_HIP(hipDeviceSynchronize());
    DS_HIP(hipMemcpy(out, b.p, sizeof(double) * M * K, hipMemcpyDeviceToHost));
    return 0;
}

int dotsocp_bfd(double *z, const double *q, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny, double scale,
                double dF) {
    Grid g;
    DS_CHECK(make_grid(g, nt, nx, ny));
    DS_ARG(z && q, "NULL pointer");
    DS_CHECK(require_device());
    DevBuf dz, dq;
    DS_CHECK(dz.alloc(10 * g.Nz));
    DS_CHECK(dq.alloc(g.NqAlloc));
    // in-place semantics: slots that mexBFd does not write keep the caller's values
    DS_HIP(hipMemcpy(dz.p, z, sizeof(double) * 10 * g.Nz, hipMemcpyHostToDevice));
    DS_HIP(hipMemcpy(dq.p, q, sizeof(double) * g.NqAlloc, hipMemcpyHostToDevice));
    DS_CHECK(launch_bfd(g, dz.p, dq.p, scale, dF, nullptr));
    DS_HIP(hipDeviceSynchronize());
    DS_HIP(hipMemcpy(z, dz.p, sizeof(double) * 10 * g.Nz, hipMemcpyDeviceToHost));
    return 0;
}

int dotsocp_bfd_conj(double *q, const double *z, dotsocp_i64 nt, dotsocp_i64 nx, dotsocp_i64 ny, double scale) {
    Grid g;
    DS_CHECK(make_grid(g, nt, nx, ny));
    DS_ARG(z && q, "NULL pointer");
    DS_CHECK(require_device());
    DevBuf dz, dq;
    DS_CHECK(dz.alloc(10 * g.Nz));
    DS_CHECK(dq.alloc(g.NqAlloc));
    DS_HIP(hipMemcpy(dz.p, z, sizeof(double) * 10 * g.Nz, hipMemcpyHostToDevice));
    DS_CHECK(launch_bfd_conj(g, dq.p, dz.p, scale, nullptr));
    DS_HIP(hipDeviceSynchronize());
    DS_HIP(hipMemcpy(q, dq.p, sizeof(double) * g.NqAlloc, hipMemcpyDeviceToHost));
    return 0;
}

// 1-D operators: the 1-D grid nx x nt is the 2-D grid with ny = nx1d, nx = 1 (no bx edges);
// cone columns [1 | bx(x-,t) bx(x+,t) bx(x-,t+1) bx(x+,t+1) | 6] live in planes {0, 5,6,7,8, 9}.
static const int k1dCols[6] = {0, 5, 6, 7, 8, 9};

int dotsocp_bfd1d(double *z, const double *q, dotsocp_i64 nt, dotsocp_i64 nx, double scale, double dF) {
    Grid g;
    DS_CHECK(make_grid(g, nt, 1, nx));
    DS_ARG(z && q, "NULL pointer");
    DS_CHECK(require_device());
    DevBuf dz, dq;
    DS_CHECK(dz.alloc(10 * g.Nz));
    DS_CHECK(dq.alloc(g.NqAlloc));
    for (int j = 0; j < 6; ++j)
        DS_HIP(hipMemcpy(dz.p + k1dCols[j] * g.Nz, z + j * g.Nz, sizeof(double) * g.Nz, hipMemcpyHostToDevice));
    DS_HIP(hipMemcpy(dq.p, q, sizeof(double) * g.NqAlloc, hipMemcpyHostToDevice));
    DS_CHECK(launch_bfd(g, dz.p, dq.p, scale, dF, nullptr));
    DS_HIP(hipDeviceSynchronize());
    for (int j = 0; j < 6; ++j)
        DS_HIP(hipMemcpy(z + j * g.Nz, dz.p + k1dCols[j] * g.Nz, sizeof(double) * g.Nz, hipMemcpyDeviceToHost));
    return 0;
}

int dotsocp_bfd_conj1d(double *q, const double *z, dotsocp_i64 nt, dotsocp_i64 nx, double scale) {
    Grid g;
    DS_CHECK(make_grid(g, nt, 1, nx));
    DS_ARG(z && q, "NULL pointer");
    DS_CHECK(require_device());
    DevBuf dz, dq;
    DS_CHECK(dz.alloc(10 * g.Nz));
    DS_CHECK(dq.alloc(g.NqAlloc));
    DS_HIP(hipMemset(dz.p, 0, sizeof(double) * 10 * g.Nz));
    for (int j = 0; j < 6; ++j)
        DS_HIP(hipMemcpy(dz.p + k1dCols[j] * g.Nz, z + j * g.Nz, sizeof(double) * g.Nz, hipMemcpyHostToDevice));
    DS_CHECK(launch_bfd_conj(g, dq.p, dz.p, scale, nullptr));
    DS_HIP(hipDeviceSynchronize());
    DS_HIP(hipMemcpy(q, dq.p, sizeof(double) * g.NqAlloc, hipMemcpyDeviceToHost));
    return 0;
}

static int dctn_dev(double *a, double *b, i64 ny, i64 nx, i64 nt, int inverse, DctPlan *py, DctPlan *px, DctPlan *pt) {
    // forward: y, x, t ; inverse: t, x, y (separable, any order gives the same transform).
    // result ends in `b`.
    if (!inverse) {
        DS_CHECK(launch_dct_axis(py, a, b, ny, nx, nt, 0, 0, nullptr));
        DS_CHECK(launch_dct_axis(px, b, a, ny, nx, nt, 1, 0, nullptr));
        DS_CHECK(launch_dct_axis(pt, a, b, ny, nx, nt, 2, 0, nullptr));
    } else {
        DS_CHECK(launch_dct_axis(pt, a, b, ny, nx, nt, 2, 1, nullptr));
        DS_CHECK(launch_dct_axis(px, b, a, ny, nx, nt, 1, 1, nullptr));
        DS_CHECK(launch_dct_axis(py, a, b, ny, nx, nt, 0, 1, nullptr));
    }
    return 0;
}

struct Plans {
    DctPlan *py = nullptr, *px = nullptr, *pt = nullptr;
    ~Plans() { dct_plan_destroy(py); dct_plan_destroy(px); dct_plan_destroy(pt); }
    int make(i64 ny, i64 nx, i64 nt) {
        py = dct_plan_create(ny); px = dct_plan_create(nx); pt = dct_plan_create(nt);
        if (!py || !px || !pt) { set_error("DCT plan allocation failed"); return DOTSOCP_EHIP; }
        return 0;
    }
};

int dotsocp_dctn(double *a, dotsocp_i64 ny, dotsocp_i64 nx, dotsocp_i64 nt, int inverse) {
    DS_ARG(a && ny >= 1 && nx >= 1 && nt >= 1, "bad array");
    DS_CHECK(require_device());
    const i64 n = ny * nx * nt;
    DevBuf da, db;
    Plans pl;
    DS_CHECK(da.alloc(n));
    DS_CHECK(db.alloc(n));
    DS_CHECK(pl.make(ny, nx, nt));
    DS_HIP(hipMemcpy(da.p, a, sizeof(double) * n, hipMemcpyHostToDevice));
    DS_CHECK(dctn_dev(da.p, db.p, ny, nx, nt, inverse, pl.py, pl.px, pl.pt));
    DS_HIP(hipDeviceSynchronize());
    DS_HIP(hipMemcpy(a, db.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

int dotsocp_oper_poisson(double *res, const double *rhs, dotsocp_i64 ny, dotsocp_i64 nx, dotsocp_i64 nt,
                         double kernelScale) {
    DS_ARG(res && rhs && ny >= 1 && nx >= 1 && nt >= 1, "bad array");
    DS_CHECK(require_device());
    const i64 n = ny * nx * nt;
    DevBuf da, db, cy, cx, ct;
    Plans pl;
    DS_CHECK(da.alloc(n));
    DS_CHECK(db.alloc(n));
    DS_CHECK(pl.make(ny, nx, nt));
    const double pi = 3.14159265358979323846;
    auto table = [&](DevBuf &d, i64 m) -> int {
        std::vector<double> t((size_t)m);
        for (i64 k = 0; k < m; ++k) t[k] = (2.0 * (double)(m - 1) * (double)(m - 1)) * (1.0 - cos(pi * (double)k / (double)m));
        DS_CHECK(d.alloc(m));
        DS_HIP(hipMemcpy(d.p, t.data(), sizeof(double) * m, hipMemcpyHostToDevice));
        return 0;
    };
    DS_CHECK(table(cy, ny));
    DS_CHECK(table(cx, nx));
    DS_CHECK(table(ct, nt));
    DS_HIP(hipMemcpy(da.p, rhs, sizeof(double) * n, hipMemcpyHostToDevice));
    const char *ts = getenv("DOTSOCP_TSOLVE");
    if (!(ts && strcmp(ts, "dct") == 0) && tsolve_tri_preferred(nt, dct_plan_is_pow2(pl.pt), ny * nx)) {
        // the sequence of Solver::poisson_all where the t axis is solved as tridiagonal systems (tri.hip): no transform along t
        Grid g;
        g.set(ny, nx, nt, 0, nt);
        DS_CHECK(launch_dct_axis(pl.py, da.p, db.p, ny, nx, nt, 0, 0, nullptr));
        DS_CHECK(launch_dct_axis(pl.px, db.p, da.p, ny, nx, nt, 1, 0, nullptr));
        DS_CHECK(launch_tsolve_tri(g, nt, kernelScale, cy.p, cx.p, da.p, nullptr));
        DS_CHECK(launch_dct_axis(pl.px, da.p, db.p, ny, nx, nt, 1, 1, nullptr));
        DS_CHECK(launch_dct_axis(pl.py, db.p, da.p, ny, nx, nt, 0, 1, nullptr));
    } else if (dct_plan_has_tsolve(pl.pt)) {
        // the sequence of Solver::poisson_all: y, x forward, the fused t pass (forward, division, inverse), x, y inverse
        DS_CHECK(launch_dct_axis(pl.py, da.p, db.p, ny, nx, nt, 0, 0, nullptr));
        DS_CHECK(launch_dct_axis(pl.px, db.p, da.p, ny, nx, nt, 1, 0, nullptr));
        DS_CHECK(launch_dct_t_solve(pl.pt, da.p, da.p, ny, ny * nx, 0, ny * nx, nt, kernelScale, cy.p, cx.p, ct.p, nullptr));
        DS_CHECK(launch_dct_axis(pl.px, da.p, db.p, ny, nx, nt, 1, 1, nullptr));
        DS_CHECK(launch_dct_axis(pl.py, db.p, da.p, ny, nx, nt, 0, 1, nullptr));
    } else {
        DS_CHECK(dctn_dev(da.p, db.p, ny, nx, nt, 0, pl.py, pl.px, pl.pt));
        DS_CHECK(launch_spectral_divide(db.p, ny, nx, nt, 0, nx, kernelScale, cy.p, cx.p, ct.p, nullptr));
        DS_CHECK(dctn_dev(db.p, da.p, ny, nx, nt, 1, pl.py, pl.px, pl.pt));
    }
    DS_HIP(hipDeviceSynchronize());
    DS_HIP(hipMemcpy(res, da.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------ B1
dotsocp_ctx *dotsocp_create(const dotsocp_problem *prob, int device, int nslabs) {
    dotsocp_ctx *c = new (std::nothrow) dotsocp_ctx();
    if (!c) { set_error("out of host memory"); return nullptr; }
    if (c->s.init(prob, device, nslabs) != 0) {
        delete c;
        return nullptr;
    }
    return c;
}

dotsocp_ctx *dotsocp_create_multi(const dotsocp_problem *prob, int first_device, int ngpu) {
    dotsocp_ctx *c = new (std::nothrow) dotsocp_ctx();
    if (!c) { set_error("out of host memory"); return nullptr; }
    if (c->s.init(prob, first_device, ngpu, true) != 0) {
        delete c;
        return nullptr;
    }
    return c;
}

void dotsocp_destroy(dotsocp_ctx *ctx) { delete ctx; }

int dotsocp_slab_range(dotsocp_i64 nt, int world, int rank, dotsocp_i64 *t0, dotsocp_i64 *t1) {
    DS_ARG(t0 && t1 && world >= 1 && rank >= 0 && rank < world && nt >= 2 * world, "bad slab request");
    return dotsocp_slab_range_impl(nt, world, rank, t0, t1);
}

int dotsocp_rccl_unique_id(unsigned char id[128]) {
    DS_ARG(id != nullptr, "id is NULL");
    Rccl &api = rccl_api();
    DS_CHECK(api.load());
    ncclUniqueId uid;
    static_assert(sizeof(uid) == 128, "ncclUniqueId is expected to be 128 bytes");
    DS_NCCL(api.GetUniqueId(&uid));
    memcpy(id, &uid, sizeof uid);
    return 0;
}

int dotsocp_attach_rccl(dotsocp_ctx *ctx, const unsigned char id[128], int rank, int world) {
    if (!ctx) { set_error("ctx is NULL"); return DOTSOCP_EINVAL; }
    return ctx->s.attach_rccl(id, rank, world);
}

#define CTX_OR_FAIL()                                      \
    do {                                                   \
        if (!ctx) { set_error("ctx is NULL"); return DOTSOCP_EINVAL; } \
        (void)hipGetLastError();   /* a stale error of an earlier, failed call must not be blamed on this one */ \
    } while (0)

dotsocp_i64 dotsocp_field_len(const dotsocp_problem *p, int field) {
    if (!p || (p->dim != 1 && p->dim != 2) || p->nt < 2 || p->nx < 1 || (p->dim == 2 && p->ny < 1)) return -1;
    const i64 ny = p->dim == 1 ? p->nx : p->ny, nx = p->dim == 1 ? 1 : p->nx, nt = p->nt;
    const i64 Nphi = ny * nx * nt, Nz = ny * nx * (nt - 1);
    const i64 Nq = Nz + ny * (nx - 1) * nt + (ny - 1) * nx * nt;
    switch (field) {
        case DOTSOCP_F_PHI: case DOTSOCP_F_C: return Nphi;
        case DOTSOCP_F_Q: case DOTSOCP_F_ALPHA: case DOTSOCP_F_WEIGHT: return Nq;
        case DOTSOCP_F_Z: case DOTSOCP_F_BETA: return Nz * (p->dim == 1 ? 6 : 10);
        default: return -1;
    }
}

int dotsocp_upload(dotsocp_ctx *ctx, int field, const double *host) { CTX_OR_FAIL(); return ctx->s.upload(field, host); }
dotsocp_i64 dotsocp_release_cache(void) { return (dotsocp_i64)device_cache_release(); }

int dotsocp_upload_layers(dotsocp_ctx *ctx, int field, const double *host, dotsocp_i64 t0, dotsocp_i64 n) {
    CTX_OR_FAIL();
    return ctx->s.upload_layers(field, host, t0, n);
}
int dotsocp_download(dotsocp_ctx *ctx, int field, double *host) { CTX_OR_FAIL(); return ctx->s.download(field, host); }
int dotsocp_begin(dotsocp_ctx *ctx, const dotsocp_opts *opts) { CTX_OR_FAIL(); return ctx->s.begin(opts); }
int dotsocp_begin_method(dotsocp_ctx *ctx, const dotsocp_opts *opts, int method, const dotsocp_acc_opts *acc) {
    CTX_OR_FAIL();
    return ctx->s.begin_method(opts, method, acc);
}
int dotsocp_recover_outputs(dotsocp_ctx *ctx, const double *rho0, const double *rho1, double *rho, double *Ex,
                            double *Ey, double *q0, double *bx, double *by) {
    CTX_OR_FAIL();
    if (ctx->s.prob.dim == 1)     // 1-D problems live in the y slot of the engine (ny = nx1d, nx = 1)
        return ctx->s.recover_outputs(rho0, rho1, rho, nullptr, Ex, q0, nullptr, bx);
    return ctx->s.recover_outputs(rho0, rho1, rho, Ex, Ey, q0, bx, by);
}

int dotsocp_jump_next_level(dotsocp_ctx *coarse, dotsocp_ctx *fine) {
    if (!coarse || !fine) { set_error("context is NULL"); return DOTSOCP_EINVAL; }
    return fine->s.jump_from(coarse->s);
}

int dotsocp_run(dotsocp_ctx *ctx, dotsocp_i64 n_iters, dotsocp_i64 *done) { CTX_OR_FAIL(); return ctx->s.run(n_iters, done); }
int dotsocp_finish(dotsocp_ctx *ctx, dotsocp_result *res) { CTX_OR_FAIL(); return ctx->s.finish(res); }

int dotsocp_get_history(dotsocp_ctx *ctx, double *kkt, double *time, double *iter, double *pdGap) {
    CTX_OR_FAIL();
    const Solver &s = ctx->s;
    const size_t n = s.hist_iter.size();
    for (size_t i = 0; i < n; ++i) {
        if (kkt) for (int j = 0; j < 7; ++j) kkt[j * n + i] = s.hist_kkt[i * 7 + j];   // len x 7, column-major
        if (time) time[i] = s.hist_time[i];
        if (iter) iter[i] = s.hist_iter[i];
        if (pdGap) pdGap[i] = s.hist_gap[i];
    }
    return 0;
}

int dotsocp_set_profiling(dotsocp_ctx *ctx, int on) {
    CTX_OR_FAIL();
    ctx->s.profiling = on != 0;
    return 0;
}

int dotsocp_kernel_time(dotsocp_ctx *ctx, const char *name, double *avg_ms, dotsocp_i64 *launches) {
    CTX_OR_FAIL();
    DS_ARG(name != nullptr, "name is NULL");
    static const char *names[PH_COUNT] = {"rhs", "poisson", "cone_proj", "qstep", "beta", "kkt",
                                          "cone_fused_a", "cone_fused_b", "materialise", "comm",
                                          "interp", "acc_cone", "acc_gather", "qstep_first", "transpose"};
    for (int i = 0; i < PH_COUNT; ++i)
        if (strcmp(name, names[i]) == 0) {
            const i64 n = ctx->s.phase_launches[i];
            if (avg_ms) *avg_ms = n ? ctx->s.phase_ms[i] / (double)n : 0.0;
            if (launches) *launches = n;
            return 0;
        }
    set_error("unknown kernel family '%s'", name);
    return DOTSOCP_EINVAL;
}

int dotsocp_canary_check(void) {
    std::string rep;
    const int bad = canary_check(&rep);
    if (bad) set_error("canary: %d device buffer(s) written out of bounds: %s", bad, rep.c_str());
    return bad;
}

int dotsocp_synchronize(dotsocp_ctx *ctx) {
    CTX_OR_FAIL();
    ctx->s.cur_dev = -1;
    return ctx->s.sync_all();
}

}  // extern "C"
