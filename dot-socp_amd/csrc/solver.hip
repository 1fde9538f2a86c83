// Host side of the device-resident inPALM / ALG2 loop.  The scalar control flow (sigma rule,
// rescale triggers, KKT ratios, stop test) restates socp/dot2d/algorithms/solver_socp_inPALM.m
// (and solver_wsocp_inPALM.m for the weighted variant) line by line; all array work is done by the
// kernels of cone.hip / fused.hip / stencil.hip / dct.hip / kkt.hip on the slab's HIP streams.
//
// Time-slab mode (world > 1): the grid is cut along t (common.h: Grid).  Per iteration a slab
// exchanges six ny x nx layers with its neighbours (u0 tail, phi head, adjoint tails, bx/by heads)
// and the Poisson solve couples the slabs along t (tri.hip, or slab <-> pencil transposes).  The same code
// runs with all slabs in one process -- each slab on its own device with its own streams, peer copies between
// them (dotsocp_create_multi; on one device: dotsocp_create(..., nslabs)) -- or with one slab per process (RCCL).
#include "solver.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "comm.h"

namespace dotsocp {

thread_local std::string g_last_error;

static int make_eig_table(double **dev, i64 n, i64 len = 0);

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int Solver::use_dev(int d) {
    if (cur_dev == d) return 0;
    DS_HIP(hipSetDevice(d));
    cur_dev = d;
    return 0;
}

int Solver::use(const Slab &s) {
    DS_CHECK(use_dev(s.dev));
    if (stream_stress_enabled() && begun && !finished) {       // race detector: perturb the relative timing of the streams
        if (s.st) stream_stress(s.st);
        if (s.st_z) stream_stress(s.st_z);
    }
    return 0;
}

int Solver::sync_all() {
    for (auto &s : slabs) {
        DS_CHECK(use(s));
        if (s.st_z) DS_HIP(ds_stream_synchronize(s.st_z));
        if (s.st) DS_HIP(ds_stream_synchronize(s.st));
    }
    if (slabs.empty() && stream) {
        DS_CHECK(use_dev(device));
        DS_HIP(ds_stream_synchronize(stream));
    }
    return 0;
}

// The slab's next ordering event (round-robin; a wait captures the record that precedes it, so reuse is safe)
static hipEvent_t next_xev(Slab &s) {
    hipEvent_t e = s.xev[s.xev_next];
    s.xev_next = (s.xev_next + 1) % DS_XEV;
    return e;
}

int Solver::xcopy(Slab &from, const double *src, Slab &to, double *dst, i64 count) {
    if (count <= 0) return 0;
    const size_t bytes = sizeof(double) * (size_t)count;
    DS_CHECK(comm_enter());
    const hipStream_t fs = cst(from), ts = cst(to);
    if (fs == ts) {
        DS_CHECK(use(to));
        DS_HIP(ds_memcpy_async(dst, src, bytes, hipMemcpyDeviceToDevice, ts));
        return comm_leave();
    }
    hipEvent_t a = next_xev(from), b = next_xev(to);
    DS_CHECK(use(from));
    DS_HIP(ds_event_record(a, fs));
    DS_CHECK(use(to));
    DS_HIP(ds_stream_wait_event(ts, a, 0));
    if (from.dev == to.dev) DS_HIP(ds_memcpy_async(dst, src, bytes, hipMemcpyDeviceToDevice, ts));
    else DS_HIP(ds_memcpy_peer_async(dst, to.dev, src, from.dev, bytes, ts));
    DS_HIP(ds_event_record(b, ts));
    DS_CHECK(use(from));
    DS_HIP(ds_stream_wait_event(fs, b, 0));
    return comm_leave();
}

int Solver::xcopy2d(Slab &from, const double *src, size_t spitch, Slab &to, double *dst, size_t dpitch, size_t width,
                    size_t height) {
    if (width == 0 || height == 0) return 0;
    DS_CHECK(comm_enter());
    const hipStream_t fs = cst(from), ts = cst(to);
    if (fs == ts) {
        DS_CHECK(use(to));
        DS_HIP(ds_memcpy2d_async(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, ts));
        return comm_leave();
    }
    hipEvent_t a = next_xev(from), b = next_xev(to);
    DS_CHECK(use(from));
    DS_HIP(ds_event_record(a, fs));
    DS_CHECK(use(to));
    DS_HIP(ds_stream_wait_event(ts, a, 0));
    if (from.dev == to.dev || peer_ok) {
        // different devices: peer access was enabled in both directions when the slabs were placed (alloc_slabs)
        DS_HIP(ds_memcpy2d_async(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, ts));
    } else {
        // peer access refused: row by row through hipMemcpyPeerAsync, which stages through the host by itself
        for (size_t r = 0; r < height; ++r)
            DS_HIP(ds_memcpy_peer_async((char *)dst + r * dpitch, to.dev, (const char *)src + r * spitch, from.dev, width, ts));
    }
    DS_HIP(ds_event_record(b, ts));
    DS_CHECK(use(from));
    DS_HIP(ds_stream_wait_event(fs, b, 0));
    return comm_leave();
}

// ---- communication on the second streams (solver.h: comm_z) ----
// (slabs that share a pair of streams -- dotsocp_create(.., nslabs) -- are served by the first of them)
int Solver::comm_fork() {
    if (!comm_z) return 0;
    FOR_SLABS(s) {
        if (&s != &slabs[0] && s.st == slabs[0].st) continue;
        DS_HIP(ds_event_record(s.ev_fork, s.st));
        DS_HIP(ds_stream_wait_event(s.st_z, s.ev_fork, 0));
    }
    return 0;
}

int Solver::comm_mark(hipEvent_t Slab::*ev) {
    if (!comm_z) return 0;
    FOR_SLABS(s) {
        if (&s != &slabs[0] && s.st == slabs[0].st) continue;
        DS_HIP(ds_event_record(s.*ev, s.st_z));
    }
    return 0;
}

int Solver::comm_wait(hipEvent_t Slab::*ev) {
    if (!comm_z) return 0;
    FOR_SLABS(s) {
        if (&s != &slabs[0] && s.st == slabs[0].st) continue;
        DS_HIP(ds_stream_wait_event(s.st, s.*ev, 0));
    }
    return 0;
}

int Solver::comm_enter() {
    if (!comm_z || comm_async) return 0;
    if (comm_depth++ == 0) DS_CHECK(comm_fork());
    return 0;
}

int Solver::comm_leave() {
    if (!comm_z || comm_async) return 0;
    if (--comm_depth == 0) {
        DS_CHECK(comm_mark(&Slab::ev_cjoin));
        DS_CHECK(comm_wait(&Slab::ev_cjoin));
    }
    return 0;
}

DevRes *Solver::res_for(int dev) {
    for (auto *r : devres)
        if (r->dev == dev) return r;
    if (use_dev(dev) != 0) return nullptr;
    DevRes *r = new DevRes();
    r->dev = dev;
    r->py = dct_plan_create(ny);
    r->px = dct_plan_create(nx);
    r->pt = dct_plan_create(nt);
    // (cy as long as a pitched row: the t-solves of a time-slab context treat the pad entries of a row as modes of their own)
    if (!r->py || !r->px || !r->pt || make_eig_table(&r->cy, ny, row_pitch()) != 0 || make_eig_table(&r->cx, nx) != 0 ||
        make_eig_table(&r->ct, nt) != 0) {
        set_error("DCT plan allocation failed on device %d", dev);
        dct_plan_destroy(r->py); dct_plan_destroy(r->px); dct_plan_destroy(r->pt);
        dfree(r->cy); dfree(r->cx); dfree(r->ct);
        delete r;
        return nullptr;
    }
    devres.push_back(r);
    return r;
}

void Solver::free_slabs() {
    defer.reset();                   // joins the slab threads (their queues are empty outside run())
    for (auto &s : slabs) {
        (void)use(s);
        if (s.st_z) (void)ds_stream_synchronize(s.st_z);
        if (s.st) (void)ds_stream_synchronize(s.st);
        for (auto &e : s.xev) if (e) (void)hipEventDestroy(e);
        if (s.ev_tri) (void)hipEventDestroy(s.ev_tri);
        if (s.ev_msg) (void)hipEventDestroy(s.ev_msg);
        if (s.ev_got) (void)hipEventDestroy(s.ev_got);
        if (s.ev_cjoin) (void)hipEventDestroy(s.ev_cjoin);
        if (s.st != stream) {        // slab 0 borrows the solver's own streams / events
            if (s.ev_fork) (void)hipEventDestroy(s.ev_fork);
            if (s.ev_join) (void)hipEventDestroy(s.ev_join);
            if (s.ev_halo) (void)hipEventDestroy(s.ev_halo);
            if (s.st_z) (void)hipStreamDestroy(s.st_z);
            if (s.st) (void)hipStreamDestroy(s.st);
        }
        if (s.h_sums) (void)hipHostFree(s.h_sums);
        dfree(s.phi); dfree(s.q); dfree(s.alpha); dfree(s.z); dfree(s.beta); dfree(s.c); dfree(s.weight);
        dfree(s.w0); dfree(s.w1); dfree(s.pencil); dfree(s.pencil2); dfree(s.stage);
        dfree(s.u0_prev); dfree(s.tail_bx); dfree(s.tail_by);
        dfree(s.a0_prev); dfree(s.a0w_prev); dfree(s.btail_bx); dfree(s.btail_by);
        dfree(s.send_plane); dfree(s.send_plane2); dfree(s.send_bx); dfree(s.send_by);
        dfree(s.kw.partials); dfree(s.kw.sums);
        dfree(s.q_old); dfree(s.q2); dfree(s.beta2); dfree(s.sx); dfree(s.sy); dfree(s.alpha2);
        dfree(s.q3); dfree(s.p2); dfree(s.sxp); dfree(s.syp);
        dfree(s.send_pbx); dfree(s.send_pby); dfree(s.ptail_bx); dfree(s.ptail_by);
        dfree(s.carry);
        dfree(s.tri_send); dfree(s.tri_recv); dfree(s.tri_bsend); dfree(s.tri_brecv); dfree(s.tri_zero);
        dfree(s.phi_p); dfree(s.alpha_p); dfree(s.z_p);
        dfree(s.phi_a); dfree(s.q_a); dfree(s.alpha_a); dfree(s.z_a); dfree(s.beta_a);
    }
    slabs.clear();
}

Solver::~Solver() {
    cur_dev = -1;
    if (stream) (void)sync_all();   // init() got as far as the device: release what lives there
    if (stream && canary_enabled()) {
        std::string rep;
        const int bad = canary_check(&rep);
        cur_dev = -1;
        if (bad) fprintf(stderr, "libdotsocp: canary: %d device buffer(s) written out of bounds: %s\n", bad, rep.c_str());
    }
    if (nccl) (void)rccl_api().CommDestroy((ncclComm_t)nccl);
    free_slabs();
    for (auto *r : devres) {
        (void)use_dev(r->dev);
        dct_plan_destroy(r->py); dct_plan_destroy(r->px); dct_plan_destroy(r->pt);
        dfree(r->cy); dfree(r->cx); dfree(r->ct);
        delete r;
    }
    devres.clear();
    (void)use_dev(device);
    dfree(d_red);
    if (h_sums) (void)hipHostFree(h_sums);
    for (auto &p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : event_pool) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_halo) (void)hipEventDestroy(ev_halo);
    if (stream_z) (void)hipStreamDestroy(stream_z);
    if (stream) (void)hipStreamDestroy(stream);
}

static int make_eig_table(double **dev, i64 n, i64 len) {
    // (2 (n-1)^2) (1 - cos(pi k / n))   -- initialize_FFTkernel.m:6-8; entries n .. len-1 (pads of a pitched row, whose data
    // are zeros) repeat the last one: any positive number keeps their systems regular
    if (len < n) len = n;
    std::vector<double> t((size_t)len);
    const double pi = 3.14159265358979323846;
    for (i64 k = 0; k < n; ++k) t[k] = (2.0 * (double)(n - 1) * (double)(n - 1)) * (1.0 - cos(pi * (double)k / (double)n));
    for (i64 k = n; k < len; ++k) t[k] = (n > 1) ? t[n - 1] : 1.0;
    DS_CHECK(dmalloc(dev, len));
    DS_HIP(hipMemcpy(*dev, t.data(), sizeof(double) * len, hipMemcpyHostToDevice));
    return 0;
}

int dotsocp_slab_range_impl(i64 nt, int world, int rank, i64 *t0, i64 *t1) {
    // nodes are dealt as evenly as possible; the last slab owns one cell layer fewer than nodes
    const i64 base = nt / world, rem = nt % world;
    const i64 a = rank * base + std::min<i64>(rank, rem);
    const i64 b = a + base + (rank < rem ? 1 : 0);
    *t0 = a;
    *t1 = b;
    return 0;
}

// pencil j of `world`: columns [l0, l1) of the ny*nx (y, x) columns, boundaries on even columns
static void pencil_range(i64 plane, int world, int j, i64 *l0, i64 *l1) {
    auto cut = [&](int k) -> i64 { return (k >= world) ? plane : 2 * ((plane / 2) * k / world); };
    *l0 = cut(j);
    *l1 = cut(j + 1);
}

// The second stream of a slab carries its messages and the small kernels between them (solver.h: comm_z): highest
// priority, so that their workgroups are placed ahead of the queued workgroups of the bulk kernel on the main stream
static int make_second_stream(hipStream_t *st) {
    int least = 0, greatest = 0;
    DS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    DS_HIP(hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest));
    return 0;
}

int Solver::init(const dotsocp_problem *p, int dev, int nslabs, bool multi_dev) {
    DS_ARG(p != nullptr, "prob is NULL");
    DS_ARG(p->dim == 1 || p->dim == 2, "prob.dim must be 1 or 2");
    DS_ARG(p->nt >= 2 && p->nx >= 1, "grid too small");
    prob = *p;
    device = dev;
    if (p->dim == 1) { ny = p->nx; nx = 1; } else { ny = p->ny; nx = p->nx; }
    nt = p->nt;
    DS_ARG(ny >= 1 && nx >= 1, "grid too small");
    DS_ARG(nslabs >= 1 && nslabs <= nt / 2, "nslabs must be in [1, nt/2]");
    if (const char *e = getenv("DOTSOCP_FUSED")) fused = (atoi(e) != 0);
    if (nslabs > 1 && !fused) {
        set_error("time slabs need the fused dataflow (unset DOTSOCP_FUSED=0)");
        return DOTSOCP_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (libdotsocp has no CPU fallback)");
        return DOTSOCP_ENODEVICE;
    }
    DS_ARG(dev >= 0 && dev < ndev, "device ordinal out of range");
    ndev_visible = ndev;
    multi_device = multi_dev && nslabs > 1;
    cur_dev = -1;
    DS_CHECK(use_dev(dev));
    DS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    DS_CHECK(make_second_stream(&stream_z));
    DS_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    DS_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    DS_HIP(hipEventCreateWithFlags(&ev_halo, hipEventDisableTiming));
    overlap = nslabs > 1;                         // pays when there is communication to hide
    if (const char *e = getenv("DOTSOCP_OVERLAP")) overlap = (atoi(e) != 0);
    if (const char *e = getenv("DOTSOCP_KKT_FOLD")) kkt_fold = (atoi(e) != 0);
    if (const char *e = getenv("DOTSOCP_NORM_CACHE")) norm_cache = (atoi(e) != 0);
    if (const char *e = getenv("DOTSOCP_TSOLVE")) tri_tsolve = (strcmp(e, "dct") != 0);
    DS_HIP(hipHostMalloc((void **)&h_sums, sizeof(double) * (S_COUNT + 1)));
    DS_CHECK(dmalloc(&d_red, S_COUNT + 1));
    if (!res_for(dev)) return DOTSOCP_EHIP;
    world = nslabs;
    rank = 0;
    // device arrays are allocated on first use (upload / begin) or by attach_rccl(), so that a process
    // that is about to become one rank of many never allocates the whole grid
    return 0;
}

int Solver::ensure_alloc() {
    if (!slabs.empty()) return 0;
    DS_CHECK(alloc_slabs(remote() ? rank : 0, remote() ? 1 : world));
    DS_CHECK(sync_all());
    return 0;
}

// Row pitch of the device arrays (common.h: Grid::py).  The single slab of a one-GPU context stores rows whose length is
// no multiple of 16 doubles -- the 2^k+1 grids of the reference's multilevel driver -- padded to the next multiple of
// 128 bytes; time-slab contexts keep the reference layout (their messages and the partitioned t-solve index the
// (y, x) columns of a layer linearly).  DOTSOCP_PITCH=0: never.
// pad between the ten columns of z and beta (common.h: Grid::Nc)
i64 Solver::column_pad() const { return (ny * nx >= 4096) ? 48 : 0; }

i64 Solver::row_pitch() const {
    static const bool on = !(getenv("DOTSOCP_PITCH") && atoi(getenv("DOTSOCP_PITCH")) == 0);
    if (!on || ny <= 16) return ny;
    if (ny % 16 == 0) {
        // Rows whose length in bytes is a multiple of 2 KB: the x lines of the Poisson solve (one 64-byte piece per row, rows a
        // power of two apart) keep hitting the same DRAM banks -- with rows 128 bytes longer the x passes of the pipelined DCT
        // kernels take 0.41 / 0.47 instead of 0.50 / 0.52 ms at 1024 x 1024 x 128 (rocprofv3, same box).  DOTSOCP_PITCH2=0: off.
        const char *e = getenv("DOTSOCP_PITCH2");
        const bool on2 = !(e && atoi(e) == 0);
        return (on2 && ny >= 512 && ny % 256 == 0) ? ny + 16 : ny;
    }
    return (ny + 15) / 16 * 16;
}

int Solver::alloc_slabs(int first, int count) {
    free_slabs();
    peer_ok = true;
    cross_device = false;
    comm_z = overlap && world > 1 && fused;      // messages on the second streams (solver.h)
    comm_depth = 0;
    comm_async = false;
    slabs.resize(count);
    const i64 plane = row_pitch() * nx;          // doubles per layer as stored (Grid::plane)
    for (int r = 0; r < count; ++r) {
        Slab &s = slabs[r];
        s.index = first + r;
        // placement: dotsocp_create_multi deals the slabs round-robin over the visible devices, starting at `device`
        s.dev = (multi_device && !remote()) ? (device + r) % ndev_visible : device;
        DS_CHECK(use(s));
        // dotsocp_create(.., nslabs): all slabs on ONE device share its pair of streams -- their kernels would only compete
        // for the same HBM (8 slabs of 1024 x 1024 x 16 on concurrent streams: 15.8 ms per iteration, one after the other
        // 8 x 1.63); dotsocp_create_multi gives every slab its own pair, whichever device it lands on
        if (r == 0 || !multi_device) {
            s.st = stream; s.st_z = stream_z;
            s.ev_fork = ev_fork; s.ev_join = ev_join; s.ev_halo = ev_halo;
        } else {
            DS_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
            DS_CHECK(make_second_stream(&s.st_z));
            DS_HIP(hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming));
            DS_HIP(hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming));
            DS_HIP(hipEventCreateWithFlags(&s.ev_halo, hipEventDisableTiming));
        }
        for (auto &e : s.xev) DS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DS_HIP(hipEventCreateWithFlags(&s.ev_tri, hipEventDisableTiming));
        DS_HIP(hipEventCreateWithFlags(&s.ev_msg, hipEventDisableTiming));
        DS_HIP(hipEventCreateWithFlags(&s.ev_got, hipEventDisableTiming));
        DS_HIP(hipEventCreateWithFlags(&s.ev_cjoin, hipEventDisableTiming));
        DS_HIP(hipHostMalloc((void **)&s.h_sums, sizeof(double) * S_COUNT));
        s.res = res_for(s.dev);
        if (!s.res) return DOTSOCP_EHIP;
        i64 t0, t1;
        dotsocp_slab_range_impl(nt, world, s.index, &t0, &t1);
        s.g.set(ny, nx, nt, t0, t1 - t0, row_pitch(), column_pad());
        const Grid &g = s.g;
        DS_CHECK(dzalloc(&s.phi, g.NphiAlloc, s.st));
        DS_CHECK(dzalloc(&s.q, g.NqAlloc, s.st));
        DS_CHECK(dzalloc(&s.alpha, g.NqAlloc, s.st));
        DS_CHECK(dzalloc(&s.z, 10 * g.Nc, s.st));
        DS_CHECK(dzalloc(&s.beta, 10 * g.Nc, s.st));
        DS_CHECK(dzalloc(&s.c, g.Nphi, s.st));
        // pitched rows: the pad entries are never written by the tile kernels, so they are zeroed once here -- the few
        // kernels that stream over whole arrays (scalings, sums of squares) then leave them zero / add nothing
        if (g.py > g.ny) {
            DS_CHECK(dzalloc(&s.w0, g.Nphi, s.st));
            DS_CHECK(dzalloc(&s.w1, g.Nphi, s.st));
        } else {
            DS_CHECK(dmalloc(&s.w0, g.Nphi));
            DS_CHECK(dmalloc(&s.w1, g.Nphi));
        }
        if (prob.weighted) {
            DS_CHECK(dmalloc(&s.weight, g.NqAlloc));
            DS_CHECK(launch_fill(s.weight, g.NqAlloc, 1.0, s.st));      // pad entries of a weight are ones (x ./ w stays finite)
        }
        if (fused) {
            fused_geometry(g, s.fg);
            DS_CHECK(dzalloc(&s.q_old, g.NqAlloc, s.st));
            DS_CHECK(dzalloc(&s.q2, g.NqAlloc, s.st));
            if (g.py > g.ny || g.Nc > g.Nz) DS_CHECK(dzalloc(&s.beta2, 10 * g.Nc, s.st));
            else DS_CHECK(dmalloc(&s.beta2, 10 * g.Nc));
            DS_CHECK(dzalloc(&s.sx, s.fg.sx_len, s.st));
            DS_CHECK(dzalloc(&s.sy, s.fg.sy_len, s.st));
            DS_CHECK(dzalloc(&s.alpha2, g.NqAlloc, s.st));
        }
        s.kw.maxBlocks = kkt_partials_needed(g);
        DS_CHECK(dzalloc(&s.kw.partials, s.kw.maxBlocks * S_COUNT, s.st));
        DS_CHECK(dmalloc(&s.kw.sums, S_COUNT * (1 + KKT_SLICES)));
        pencil_range(plane, world, s.index, &s.l0, &s.nl);
        s.nl -= s.l0;
        if (multi()) {
            DS_CHECK(dmalloc(&s.pencil, s.nl * nt));
            DS_CHECK(dmalloc(&s.pencil2, s.nl * nt));
            DS_CHECK(dmalloc(&s.stage, g.Nphi));
            if (fused) DS_CHECK(dzalloc(&s.carry, 4 * g.plane, s.st));
            if (!g.first) {
                DS_CHECK(dzalloc(&s.u0_prev, plane, s.st));
                DS_CHECK(dzalloc(&s.a0_prev, plane, s.st));
                DS_CHECK(dzalloc(&s.a0w_prev, plane, s.st));
                DS_CHECK(dzalloc(&s.tail_bx, g.bxLayer, s.st));
                DS_CHECK(dzalloc(&s.btail_bx, g.bxLayer, s.st));
                DS_CHECK(dzalloc(&s.tail_by, g.byLayer, s.st));
                DS_CHECK(dzalloc(&s.btail_by, g.byLayer, s.st));
            }
            if (!g.last) {
                DS_CHECK(dzalloc(&s.send_plane, plane, s.st));
                DS_CHECK(dzalloc(&s.send_plane2, plane, s.st));
                DS_CHECK(dzalloc(&s.send_bx, g.bxLayer, s.st));
                DS_CHECK(dzalloc(&s.send_by, g.byLayer, s.st));
            }
        }
    }
    // neighbours on different devices copy layers into each other's memory
    for (auto &a : slabs)
        for (auto &b : slabs) {
            if (a.dev == b.dev) continue;
            cross_device = true;
            DS_CHECK(use(a));
            hipError_t e = hipDeviceEnablePeerAccess(b.dev, 0);
            // a refusal is not fatal: hipMemcpyPeerAsync stages through the host without peer access -- but the launches
            // that PULL messages through peer pointers (flush_msgs, tri_exchange) must then stay off
            if (e != hipSuccess) {
                (void)hipGetLastError();
                if (e != hipErrorPeerAccessAlreadyEnabled) peer_ok = false;
            }
        }
    // several slabs with their own streams in this process (dotsocp_create_multi): one issuing thread per slab (defer.h) --
    // OPT-IN (DOTSOCP_HOST_THREADS=1) until a box with several devices has shown both bit-equal results and a gain: the one
    // configuration that could be measured, all slabs on one device, is slower with the threads (8 slabs on 64^3: 1.26 vs
    // 1.06-1.17 ms per iteration -- they contend for the one submission queue and add hand-off latency)
    {
        const char *e = getenv("DOTSOCP_HOST_THREADS");
        const bool want = e && atoi(e) != 0;
        if (want && count > 1 && !remote() && multi_device) {
            defer.reset(new DeferCtx());
            for (auto &s : slabs) {
                const int w = defer->add_worker(s.dev);
                defer->map_stream(s.st, w);
                defer->map_stream(s.st_z, w);
            }
        }
    }
    return 0;
}

int Solver::attach_rccl(const unsigned char *id, int rk, int wd) {
    DS_ARG(id != nullptr, "unique id is NULL");
    DS_ARG(wd >= 1 && rk >= 0 && rk < wd, "bad rank / world");
    DS_ARG(wd <= DS_MAX_WORLD, "at most 64 slabs");
    DS_ARG(wd <= nt / 2, "world must not exceed nt/2 time slabs");
    if (begun || world != 1 || !slabs.empty()) {
        set_error("attach_rccl() must directly follow create(..., nslabs = 1)");
        return DOTSOCP_ESTATE;
    }
    if (!fused) { set_error("time slabs need the fused dataflow (unset DOTSOCP_FUSED=0)"); return DOTSOCP_EINVAL; }
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    Rccl &api = rccl_api();
    DS_CHECK(api.load());
    ncclUniqueId uid;
    static_assert(sizeof(uid) == 128, "ncclUniqueId is expected to be 128 bytes");
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    DS_NCCL(api.CommInitRank(&comm, wd, uid, rk));
    nccl = comm;
    world = wd;
    rank = rk;
    if (!getenv("DOTSOCP_OVERLAP")) overlap = wd > 1;
    DS_CHECK(ensure_alloc());
    DS_HIP(ds_stream_synchronize(stream));
    if (wd > 1) {
        // handshake: the communicator spans `wd` ranks and the neighbours are the ranks this slab expects (also opens
        // the neighbour connections before the first timed iteration)
        // (on the stream that carries every later message of this communicator: RCCL sees ONE stream)
        const hipStream_t cs = comm_z ? stream_z : stream;
        double h[4] = {1.0, (double)rk, 0.0, -1.0};
        double *d = nullptr;
        DS_CHECK(dmalloc(&d, 4));
        DS_HIP(ds_memcpy_async(d, h, sizeof h, hipMemcpyHostToDevice, cs));
        DS_NCCL(api.AllReduce(d, d + 2, 1, ncclDouble, ncclSum, comm, cs));
        DS_NCCL(api.GroupStart());
        ++open_groups;
        if (rk + 1 < wd) DS_NCCL_G(api.Send(d + 1, 1, ncclDouble, rk + 1, comm, cs));
        if (rk > 0) DS_NCCL_G(api.Recv(d + 3, 1, ncclDouble, rk - 1, comm, cs));
        --open_groups;
        DS_NCCL(api.GroupEnd());
        DS_HIP(ds_memcpy_async(h, d, sizeof h, hipMemcpyDeviceToHost, cs));
        DS_HIP(ds_stream_synchronize(cs));
        dfree(d);
        if (h[2] != (double)wd || (rk > 0 && h[3] != (double)(rk - 1))) {
            set_error("RCCL handshake failed: %g ranks answered (expected %d), left neighbour says %g (expected %d)", h[2], wd,
                      h[3], rk - 1);
            return DOTSOCP_ECOMM;
        }
    }
    return 0;
}

// --------------------------------------------------------------------------------------
// neighbour exchanges
// --------------------------------------------------------------------------------------
int Solver::shift(int dir, const Sel &src, const Sel &dst, i64 count) {
    if (!multi() || count <= 0) return 0;
    DS_CHECK(comm_enter());
    if (!remote()) {
        if (msg_batching()) {
            for (size_t i = 0; i + 1 < slabs.size(); ++i) {
                const int f = (int)((dir > 0) ? i : i + 1), t = (int)((dir > 0) ? i + 1 : i);
                msgs.push_back(Msg{f, t, src(slabs[f]), dst(slabs[t]), count});
            }
            if (msg_depth == 0) DS_CHECK(flush_msgs());       // a lone shift() is a group of one
            return comm_leave();
        }
        for (size_t i = 0; i + 1 < slabs.size(); ++i) {
            Slab &from = (dir > 0) ? slabs[i] : slabs[i + 1];
            Slab &to = (dir > 0) ? slabs[i + 1] : slabs[i];
            DS_CHECK(xcopy(from, src(from), to, dst(to), count));
        }
        return comm_leave();
    }
    Rccl &api = rccl_api();
    Slab &s = slabs[0];
    const hipStream_t cs = cst(s);
    const int to = rank + dir, from = rank - dir;
    DS_NCCL(api.GroupStart());
    ++open_groups;
    if (to >= 0 && to < world) DS_NCCL_G(api.Send(src(s), (size_t)count, ncclDouble, to, (ncclComm_t)nccl, cs));
    if (from >= 0 && from < world) DS_NCCL_G(api.Recv(dst(s), (size_t)count, ncclDouble, from, (ncclComm_t)nccl, cs));
    --open_groups;
    DS_NCCL(api.GroupEnd());
    return comm_leave();
}

int Solver::shift_edge_halo(const Sel &base) {
    if (!multi()) return 0;
    DS_CHECK(shift(-1, [&](Slab &s) { return base(s) + s.g.offBx; },
                   [&](Slab &s) { return base(s) + s.g.offBx + s.g.bxLayer * s.g.ntl; }, slabs[0].g.bxLayer));
    DS_CHECK(shift(-1, [&](Slab &s) { return base(s) + s.g.offBy; },
                   [&](Slab &s) { return base(s) + s.g.offBy + s.g.byLayer * s.g.ntl; }, slabs[0].g.byLayer));
    return 0;
}

// Several shift() calls issued as ONE RCCL group (nested groups are legal): traffic to the left and
// to the right neighbour then shares the bidirectional links instead of queueing behind each other.
int Solver::group_begin() {
    DS_CHECK(comm_enter());
    if (remote()) {
        DS_NCCL(rccl_api().GroupStart());
        ++open_groups;
    } else {
        ++msg_depth;
    }
    return 0;
}

int Solver::group_end() {
    if (remote()) {
        --open_groups;
        DS_NCCL(rccl_api().GroupEnd());
    } else if (--msg_depth == 0) {
        DS_CHECK(flush_msgs());
    }
    return comm_leave();
}

// Pull launches (a kernel on the receiver's device reading the sender's memory through a peer pointer): the default
// between slabs of ONE device.  Between different devices the event-ordered hipMemcpyPeerAsync copies are the default
// and the pull launches are opt-in (DOTSOCP_MSG_BATCH=1 / DOTSOCP_TRI_GATHER=1): whether the reading device's L2
// returns fresh lines of another device's coarse-grained memory behind nothing but an event wait has never been
// observed on two devices (the build's boxes have one).
bool Solver::pull_default(const char *var) const {
    const char *e = getenv(var);                                  // read per call: the tests switch it inside one process
    if (!peer_ok) return false;
    if (e) return atoi(e) != 0;
    return !cross_device;
}

bool Solver::msg_batching() const { return pull_default("DOTSOCP_MSG_BATCH"); }

// The collected copies of a group, as the event-ordered copies of xcopy() would do them but with one set of events and
// one launch per slab: every sender records "written", every receiver waits for its senders, pulls all its messages
// with one launch (peer pointers) and records "pulled", every sender waits for its receivers (its buffers are free).
int Solver::flush_msgs() {
    if (msgs.empty()) return 0;
    const size_t P = slabs.size();
    std::vector<char> sends(P, 0), gets(P, 0);
    for (const Msg &m : msgs)
        if (m.count > 0) { sends[m.from] = 1; gets[m.to] = 1; }
    for (size_t i = 0; i < P; ++i)
        if (sends[i]) {
            bool other = false;
            for (const Msg &m : msgs) other = other || (m.from == (int)i && m.count > 0 && cst(slabs[m.to]) != cst(slabs[i]));
            if (!other) continue;
            DS_CHECK(use(slabs[i]));
            DS_HIP(ds_event_record(slabs[i].ev_msg, cst(slabs[i])));
        }
    for (size_t t = 0; t < P; ++t) {
        if (!gets[t]) continue;
        Slab &to = slabs[t];
        DS_CHECK(use(to));
        GatherMsgs g{};
        g.n = 0;
        std::vector<char> waited(P, 0);
        for (const Msg &m : msgs) {
            if (m.to != (int)t || m.count <= 0) continue;
            if (!waited[m.from] && cst(slabs[m.from]) != cst(to)) {
                DS_HIP(ds_stream_wait_event(cst(to), slabs[m.from].ev_msg, 0));
                waited[m.from] = 1;
            }
            if (g.n == DS_MAX_WORLD) {
                DS_CHECK(launch_gather_msgs(g, cst(to)));
                g.n = 0;
            }
            g.src[g.n] = m.src; g.dst[g.n] = m.dst; g.count[g.n] = m.count;
            ++g.n;
        }
        DS_CHECK(launch_gather_msgs(g, cst(to)));
        bool other = false;
        for (const Msg &m : msgs) other = other || (m.to == (int)t && m.count > 0 && cst(slabs[m.from]) != cst(to));
        if (other) DS_HIP(ds_event_record(to.ev_got, cst(to)));
    }
    for (size_t f = 0; f < P; ++f) {
        if (!sends[f]) continue;
        Slab &from = slabs[f];
        DS_CHECK(use(from));
        std::vector<char> waited(P, 0);
        for (const Msg &m : msgs)
            if (m.from == (int)f && m.count > 0 && !waited[m.to] && cst(slabs[m.to]) != cst(from)) {
                DS_HIP(ds_stream_wait_event(cst(from), slabs[m.to].ev_got, 0));
                waited[m.to] = 1;
            }
    }
    msgs.clear();
    return 0;
}

// u0 = w.*q0 - alpha0 of every slab's last cell layer -> right neighbour (first node layer of its rhs)
int Solver::make_u0_tail() {
    if (!multi() || u0_made) return 0;        // (u0_made: the q-step wrote it)
    for (auto &s : slabs)
        if (!s.g.last) {
            DS_CHECK(use(s));
            DS_CHECK(launch_u0_tail(s.g, s.q, s.alpha, s.weight, s.send_plane, s.st));
        }
    return 0;
}

// (make_u0_tail() first: its kernel runs on the main streams)
int Solver::exchange_u0_tail() {
    if (!multi()) return 0;
    DS_CHECK(shift(+1, [](Slab &s) { return s.send_plane; }, [](Slab &s) { return s.u0_prev; }, slabs[0].g.plane));
    u0_fresh = true;
    return 0;
}

// first owned bx / by layers of every slab -> halo layer of its left neighbour; with_u0: the u0 tail
// of the new iterate travels to the right in the same group
int Solver::exchange_q_halo(bool with_u0) {
    if (!multi()) return 0;
    if (with_u0 && !comm_async) DS_CHECK(make_u0_tail());     // (an asynchronous caller has run it before its fork)
    prof_begin(PH_COMM, comm_z);
    const i64 bxL = slabs[0].g.bxLayer, byL = slabs[0].g.byLayer;
    DS_CHECK(group_begin());
    DS_CHECK(shift(-1, [](Slab &s) { return s.q + s.g.offBx; },
                   [](Slab &s) { return s.q + s.g.offBx + s.g.bxLayer * s.g.ntl; }, bxL));
    DS_CHECK(shift(-1, [](Slab &s) { return s.q + s.g.offBy; },
                   [](Slab &s) { return s.q + s.g.offBy + s.g.byLayer * s.g.ntl; }, byL));
    if (with_u0) DS_CHECK(exchange_u0_tail());
    DS_CHECK(group_end());
    prof_end(PH_COMM, comm_z);
    return 0;
}

int Solver::ensure_halo() {
    if (!halo_pending) return 0;
    halo_pending = false;
    return exchange_q_halo(true);
}

// slabs [y][x][t_local] <-> pencils [columns l0..l0+nl)[all t]; data in w0 resp. pencil
int Solver::transpose(bool forward) {
    const i64 plane = slabs[0].g.plane;
    if (!remote()) {
        for (auto &s : slabs)
            for (auto &p : slabs) {
                double *slabPtr = s.w0 + p.l0;                    // layer pitch plane
                double *penPtr = p.pencil + p.nl * s.g.t0;        // layer pitch p.nl
                if (p.nl <= 0) continue;
                if (forward)
                    DS_CHECK(xcopy2d(s, slabPtr, sizeof(double) * plane, p, penPtr, sizeof(double) * p.nl,
                                     sizeof(double) * p.nl, (size_t)s.g.ntl));
                else
                    DS_CHECK(xcopy2d(p, penPtr, sizeof(double) * p.nl, s, slabPtr, sizeof(double) * plane,
                                     sizeof(double) * p.nl, (size_t)s.g.ntl));
            }
        return 0;
    }
    // one slab per process: pack the part of every peer contiguously (one kernel), then one send/recv per peer;
    // the own part is a plain device copy
    Rccl &api = rccl_api();
    Slab &s = slabs[0];
    std::vector<i64> off(world + 1, 0), pl0(world), pnl(world), pt0(world), pntl(world);
    PencilCuts pc{};
    pc.world = world;
    for (int j = 0; j < world; ++j) {
        i64 a, b;
        pencil_range(plane, world, j, &a, &b);
        pl0[j] = a;
        pnl[j] = b - a;
        pc.cut[j] = a;
        pc.cut[j + 1] = b;
        dotsocp_slab_range_impl(nt, world, j, &a, &b);
        pt0[j] = a;
        pntl[j] = b - a;
        off[j + 1] = off[j] + pnl[j] * s.g.ntl;
    }
    auto self_copy = [&](bool fwd) -> int {
        if (s.nl <= 0) return 0;
        double *st_ = s.stage + off[rank], *pe = s.pencil + s.nl * pt0[rank];
        const size_t bytes = sizeof(double) * (size_t)(s.nl * s.g.ntl);
        DS_HIP(ds_memcpy_async(fwd ? pe : st_, fwd ? st_ : pe, bytes, hipMemcpyDeviceToDevice, stream));
        return 0;
    };
    if (forward) {
        DS_CHECK(launch_pencil_pack(true, pc, plane, s.g.ntl, s.w0, s.stage, stream));
        DS_CHECK(self_copy(true));
        DS_CHECK(comm_enter());
        const hipStream_t cs = cst(s);
        DS_NCCL(api.GroupStart());
        ++open_groups;
        for (int j = 0; j < world; ++j) {
            if (j == rank) continue;
            if (pnl[j] > 0)
                DS_NCCL_G(api.Send(s.stage + off[j], (size_t)(pnl[j] * s.g.ntl), ncclDouble, j, (ncclComm_t)nccl, cs));
            if (s.nl > 0)
                DS_NCCL_G(api.Recv(s.pencil + s.nl * pt0[j], (size_t)(s.nl * pntl[j]), ncclDouble, j, (ncclComm_t)nccl, cs));
        }
        --open_groups;
        DS_NCCL(api.GroupEnd());
        DS_CHECK(comm_leave());
    } else {
        DS_CHECK(self_copy(false));
        DS_CHECK(comm_enter());
        const hipStream_t cs = cst(s);
        DS_NCCL(api.GroupStart());
        ++open_groups;
        for (int j = 0; j < world; ++j) {
            if (j == rank) continue;
            if (s.nl > 0)
                DS_NCCL_G(api.Send(s.pencil + s.nl * pt0[j], (size_t)(s.nl * pntl[j]), ncclDouble, j, (ncclComm_t)nccl, cs));
            if (pnl[j] > 0)
                DS_NCCL_G(api.Recv(s.stage + off[j], (size_t)(pnl[j] * s.g.ntl), ncclDouble, j, (ncclComm_t)nccl, cs));
        }
        --open_groups;
        DS_NCCL(api.GroupEnd());
        DS_CHECK(comm_leave());
        DS_CHECK(launch_pencil_pack(false, pc, plane, s.g.ntl, s.w0, s.stage, stream));
    }
    return 0;
}

// --------------------------------------------------------------------------------------
// Time-slab Poisson solve without transposes (tri.hip): local eliminations, 2 numbers per mode to the mode's owner,
// reduced systems there, 2 numbers per mode back, local solves.
// --------------------------------------------------------------------------------------
static void tri_layout(i64 plane, i64 nt, int world, PencilCuts &pc, std::vector<i64> &slab_n) {
    pc.world = world;
    slab_n.assign(world, 0);
    for (int j = 0; j < world; ++j) {
        i64 a, b;
        pencil_range(plane, world, j, &a, &b);
        pc.cut[j] = a;
        pc.cut[j + 1] = b;
        dotsocp_slab_range_impl(nt, world, j, &a, &b);
        slab_n[j] = b - a;
    }
}

int Solver::tri_alloc() {
    const i64 plane = slabs[0].g.plane;
    for (auto &s : slabs) {
        if (s.tri_send) continue;
        DS_CHECK(use(s));
        DS_CHECK(dzalloc(&s.tri_send, 2 * plane + (i64)TRI_EXTRA * world, s.st));
        DS_CHECK(dzalloc(&s.tri_brecv, 2 * plane + (i64)TRI_EXTRA * world, s.st));
        DS_CHECK(dzalloc(&s.tri_recv, (2 * s.nl + TRI_EXTRA) * world, s.st));
        DS_CHECK(dzalloc(&s.tri_bsend, (2 * s.nl + TRI_EXTRA) * world, s.st));
        DS_CHECK(dzalloc(&s.tri_zero, nt, s.st));
    }
    return 0;
}

// back == false: every slab's message for owner j -> owner j (slot of the sending slab); back == true: the way back
int Solver::tri_exchange(bool back) {
    const i64 plane = slabs[0].g.plane;
    PencilCuts pc{};
    std::vector<i64> slab_n;
    tri_layout(plane, nt, world, pc, slab_n);
    auto off = [&](int j) { return 2 * pc.cut[j] + (i64)TRI_EXTRA * j; };                  // in tri_send / tri_brecv
    auto cnt = [&](int j) { return 2 * (pc.cut[j + 1] - pc.cut[j]) + (i64)TRI_EXTRA; };  // message for / from owner j
    DS_CHECK(comm_enter());
    if (!remote()) {
        // Slabs of one process: every receiver pulls all its messages with ONE launch (peer pointers; P launches and
        // P * P stream waits instead of P * P event-ordered copies, whose host cost grew to 2.8 ms per iteration at
        // eight slabs).  "Message written" is one event per slab; the buffers need no event for their reuse: a sender
        // overwrites its message only behind its own next gather, which waits for every receiver of this one.
        const bool gather = pull_default("DOTSOCP_TRI_GATHER");
        if (gather) {
            bool one = true;
            for (auto &s : slabs) one = one && cst(s) == cst(slabs[0]);
            if (!one) {
                FOR_SLABS(s) DS_HIP(ds_event_record(s.ev_tri, cst(s)));
            }
            FOR_SLABS(sd) {                 // receiver: owner j (forward), slab p (back)
                GatherMsgs m{};
                m.n = 0;
                for (auto &ss : slabs) {    // sender
                    if (cst(ss) != cst(sd)) DS_HIP(ds_stream_wait_event(cst(sd), ss.ev_tri, 0));
                    const int d = sd.index, q = ss.index;
                    if (!back) {            // slab q's message for owner d
                        m.src[m.n] = ss.tri_send + off(d);
                        m.dst[m.n] = sd.tri_recv + (i64)q * cnt(d);
                        m.count[m.n] = cnt(d);
                    } else {                // owner q's answer for slab d
                        m.src[m.n] = ss.tri_bsend + (i64)d * cnt(q);
                        m.dst[m.n] = sd.tri_brecv + off(q);
                        m.count[m.n] = cnt(q);
                    }
                    ++m.n;
                }
                DS_CHECK(launch_gather_msgs(m, cst(sd)));
            }
            return comm_leave();
        }
        for (auto &sp : slabs)             // slab p
            for (auto &sj : slabs) {       // owner j
                const int p = sp.index, j = sj.index;
                double *a = sp.tri_send + off(j), *b = sj.tri_recv + (i64)p * cnt(j);
                if (back) { a = sj.tri_bsend + (i64)p * cnt(j); b = sp.tri_brecv + off(j); }
                if (back) DS_CHECK(xcopy(sj, a, sp, b, cnt(j)));
                else DS_CHECK(xcopy(sp, a, sj, b, cnt(j)));
            }
        return comm_leave();
    }
    // one slab per process.  The rank's own part does not travel: k_tri_reduced reads it where k_tri_local wrote it and
    // k_tri_final reads the answer where k_tri_reduced left it (launch_tri_reduced / _final: `own`)
    Rccl &api = rccl_api();
    Slab &s = slabs[0];
    const hipStream_t cs = cst(s);
    DS_NCCL(api.GroupStart());
    ++open_groups;
    for (int j = 0; j < world; ++j) {
        if (j == rank) continue;
        if (!back) {
            DS_NCCL_G(api.Send(s.tri_send + off(j), (size_t)cnt(j), ncclDouble, j, (ncclComm_t)nccl, cs));
            DS_NCCL_G(api.Recv(s.tri_recv + (i64)j * cnt(rank), (size_t)cnt(rank), ncclDouble, j, (ncclComm_t)nccl, cs));
        } else {
            DS_NCCL_G(api.Send(s.tri_bsend + (i64)j * cnt(rank), (size_t)cnt(rank), ncclDouble, j, (ncclComm_t)nccl, cs));
            DS_NCCL_G(api.Recv(s.tri_brecv + off(j), (size_t)cnt(j), ncclDouble, j, (ncclComm_t)nccl, cs));
        }
    }
    --open_groups;
    DS_NCCL(api.GroupEnd());
    return comm_leave();
}

// hooks (asynchronous schedule of step(), messages on the second streams): the latency-bound middle of the solve -- local
// eliminations, interface exchange, reduced systems, interface exchange: two small kernels and two rounds of messages --
// runs on the SECOND streams while hooks->fill (the last cone chunk) keeps the main streams busy; hooks->behind is called
// once both have been joined (more messages for the second streams).
int Solver::poisson_t_tridiag(const PhiHooks *hooks) {
    const i64 plane = slabs[0].g.plane;
    DS_CHECK(tri_alloc());
    PencilCuts pc{};
    std::vector<i64> slab_n;
    tri_layout(plane, nt, world, pc, slab_n);
    const double kscale = D * D;
    const bool async = hooks != nullptr && comm_z;
    auto own_off = [&](const Slab &s) { return 2 * pc.cut[s.index] + (i64)TRI_EXTRA * s.index; };
    if (async) {
        DS_CHECK(comm_fork());
        comm_async = true;
    }
    // (synchronous form: kernels on the main streams, every exchange forks and joins by itself)
    FOR_SLABS(s) DS_CHECK(launch_tri_local(s.g, nt, kscale, s.res->cy, s.res->cx, pc, s.w0, s.tri_send, async ? cst(s) : s.st));
    int rc = 0;
    prof_begin(PH_TRANSPOSE, async);
    rc = tri_exchange(false);
    prof_end(PH_TRANSPOSE, async);
    if (rc == 0) {
        for (auto &s : slabs) {
            if ((rc = use(s)) != 0) break;
            const bool own = remote();       // one slab per process: the own message stays where it is (tri_exchange)
            rc = launch_tri_reduced(s.g, nt, kscale, s.res->cy, s.res->cx, pc, s.index, s.l0, s.nl, slab_n.data(), s.tri_recv,
                                    s.tri_bsend, s.tri_zero, async ? cst(s) : s.st, own ? s.tri_send + own_off(s) : nullptr,
                                    own ? s.tri_brecv + own_off(s) : nullptr);
            if (rc != 0) break;
        }
    }
    if (rc == 0) {
        prof_begin(PH_TRANSPOSE, async);
        rc = tri_exchange(true);
        prof_end(PH_TRANSPOSE, async);
    }
    comm_async = false;
    DS_CHECK(rc);
    if (async) DS_CHECK(comm_mark(&Slab::ev_halo));
    if (hooks && hooks->fill) {
        prof_end(PH_POISSON);
        DS_CHECK(hooks->fill());
        prof_begin(PH_POISSON);
    }
    if (async) DS_CHECK(comm_wait(&Slab::ev_halo));
    if (hooks && hooks->behind) DS_CHECK(hooks->behind());
    FOR_SLABS(s) DS_CHECK(launch_tri_final(s.g, nt, kscale, s.res->cy, s.res->cx, pc, s.tri_brecv, s.w0, s.st));
    return 0;
}

// --------------------------------------------------------------------------------------
// upload / download.  Host pointers hold the GLOBAL field in the reference layout; with an RCCL
// communicator attached they hold this process's slab of it (same layout restricted to the owned
// layers: q = [q0 cells | bx layers | by layers]).
// --------------------------------------------------------------------------------------
static const int k1dCols[6] = {0, 5, 6, 7, 8, 9};   // 1-D cone columns inside the 10-plane layout

i64 Solver::field_len(int field) const {
    i64 ntn = nt, ntc = nt - 1;
    if (remote()) { ntn = slabs[0].g.ntl; ntc = slabs[0].g.ncl; }
    const i64 Nz = ny * nx * ntc, Nphi = ny * nx * ntn;
    const i64 Nq = Nz + ny * (nx - 1) * ntn + (ny - 1) * nx * ntn;
    switch (field) {
        case DOTSOCP_F_PHI: case DOTSOCP_F_C: return Nphi;
        case DOTSOCP_F_Q: case DOTSOCP_F_ALPHA: case DOTSOCP_F_WEIGHT: return Nq;
        case DOTSOCP_F_Z: case DOTSOCP_F_BETA: return Nz * (prob.dim == 1 ? 6 : 10);
        default: return -1;
    }
}

// rows of `rowlen` doubles: device rows `pitch` apart, host rows contiguous (reference layout)
int Solver::copy_rows(double *dev, double *host, i64 rowlen, i64 pitch, i64 nrows, bool up, hipStream_t st) {
    if (rowlen <= 0 || nrows <= 0) return 0;
    if (pitch == rowlen) {
        if (up) DS_HIP(ds_memcpy_async(dev, host, sizeof(double) * rowlen * nrows, hipMemcpyHostToDevice, st));
        else DS_HIP(ds_memcpy_async(host, dev, sizeof(double) * rowlen * nrows, hipMemcpyDeviceToHost, st));
        return 0;
    }
    if (up) DS_HIP(ds_memcpy2d_async(dev, sizeof(double) * pitch, host, sizeof(double) * rowlen, sizeof(double) * rowlen,
                                    (size_t)nrows, hipMemcpyHostToDevice, st));
    else DS_HIP(ds_memcpy2d_async(host, sizeof(double) * rowlen, dev, sizeof(double) * pitch, sizeof(double) * rowlen,
                                 (size_t)nrows, hipMemcpyDeviceToHost, st));
    return 0;
}

static int copy_field(Solver &S, int field, double *host, bool up) {
    const i64 ny = S.ny, nx = S.nx;
    i64 ntn = S.nt, ntc = S.nt - 1;
    if (S.remote()) { ntn = S.slabs[0].g.ntl; ntc = S.slabs[0].g.ncl; }
    // host side: the reference layout q = [q0 (ny, nx, nt-1) ; bx (ny, nx-1, nt) ; by (ny-1, nx, nt)]; device side: rows
    // py (by: pyb) doubles apart (common.h)
    const i64 hplane = ny * nx, hbx = ny * (nx - 1), hby = (ny - 1) * nx;
    const i64 NzG = hplane * ntc;
    const i64 bxG = NzG, byG = NzG + hbx * ntn;
    for (auto &s : S.slabs) {
        DS_CHECK(S.use(s));
        hipStream_t cur = s.st;
        const Grid &g = s.g;
        const i64 t0 = S.remote() ? 0 : g.t0;
        auto nodes = [&](double *dev, double *h, i64 layers) { return S.copy_rows(dev, h, ny, g.py, nx * layers, up, cur); };
        switch (field) {
            case DOTSOCP_F_PHI: DS_CHECK(nodes(s.phi, host + hplane * t0, g.ntl)); break;
            case DOTSOCP_F_C: DS_CHECK(nodes(s.c, host + hplane * t0, g.ntl)); break;
            case DOTSOCP_F_Q: case DOTSOCP_F_ALPHA: case DOTSOCP_F_WEIGHT: {
                double *d = field == DOTSOCP_F_Q ? s.q : (field == DOTSOCP_F_ALPHA ? s.alpha : s.weight);
                DS_CHECK(nodes(d, host + hplane * t0, g.ncl));
                DS_CHECK(S.copy_rows(d + g.offBx, host + bxG + hbx * t0, ny, g.py, (nx - 1) * g.ntl, up, cur));
                DS_CHECK(S.copy_rows(d + g.offBy, host + byG + hby * t0, ny - 1, g.pyb, nx * g.ntl, up, cur));
                break;
            }
            case DOTSOCP_F_Z: case DOTSOCP_F_BETA: {
                double *d = field == DOTSOCP_F_Z ? s.z : s.beta;
                const int K = S.prob.dim == 1 ? 6 : 10;
                if (up && S.prob.dim == 1) DS_HIP(ds_memset_async(d, 0, sizeof(double) * 10 * g.Nc, cur));
                for (int j = 0; j < K; ++j) {
                    const int pj = S.prob.dim == 1 ? k1dCols[j] : j;
                    DS_CHECK(nodes(d + pj * g.Nc, host + j * NzG + hplane * t0, g.ncl));
                }
                break;
            }
        }
    }
    DS_CHECK(S.sync_all());
    return 0;
}

int Solver::upload(int field, const double *host) {
    DS_ARG(host != nullptr, "host pointer is NULL");
    DS_ARG(field_len(field) >= 0, "unknown field");
    DS_ARG(field != DOTSOCP_F_WEIGHT || prob.weighted, "weight uploaded to an unweighted problem");
    if (begun) { set_error("upload() after begin()"); return DOTSOCP_ESTATE; }
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    DS_CHECK(ensure_alloc());
    return copy_field(*this, field, const_cast<double *>(host), true);
}

// time layers [t0, t0 + n) of a node field (phi, c) from a host buffer that holds only those layers; the other layers
// keep what they have (zeros after create).  model.c of initialize.m:42-50 is zero except for its first and last layer:
// a driver uploads those two instead of a vector as long as the grid (1 GB at 1025 x 1025 x 129).
int Solver::upload_layers(int field, const double *host, i64 t0, i64 n) {
    DS_ARG(host != nullptr, "host pointer is NULL");
    DS_ARG(field == DOTSOCP_F_PHI || field == DOTSOCP_F_C, "layer uploads serve the node fields (phi, c)");
    const i64 ntn = remote() ? slabs[0].g.ntl : nt;
    DS_ARG(t0 >= 0 && n >= 0 && t0 + n <= ntn, "layer range outside the field");
    if (begun) { set_error("upload() after begin()"); return DOTSOCP_ESTATE; }
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    DS_CHECK(ensure_alloc());
    const i64 hplane = ny * nx;
    for (auto &s : slabs) {
        DS_CHECK(use(s));
        const Grid &g = s.g;
        const i64 base = remote() ? 0 : g.t0;
        const i64 lo = std::max(t0, base), hi = std::min(t0 + n, base + g.ntl);
        if (lo >= hi) continue;
        double *dev = (field == DOTSOCP_F_PHI ? s.phi : s.c) + g.plane * (lo - base);
        DS_CHECK(copy_rows(dev, const_cast<double *>(host) + hplane * (lo - t0), ny, g.py, nx * (hi - lo), true, s.st));
    }
    DS_CHECK(sync_all());
    return 0;
}

int Solver::download(int field, double *host) {
    DS_ARG(host != nullptr, "host pointer is NULL");
    DS_ARG(field_len(field) >= 0, "unknown field");
    DS_ARG(field != DOTSOCP_F_WEIGHT || prob.weighted, "no weight in an unweighted problem");
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    DS_CHECK(ensure_alloc());
    if (field == DOTSOCP_F_Z || field == DOTSOCP_F_BETA) {
        DS_CHECK(ensure_z());
        DS_CHECK(flush_beta());
    }
    if (field == DOTSOCP_F_ALPHA) DS_CHECK(flush_alpha());
    host_first_touch(host, sizeof(double) * (size_t)field_len(field));
    DS_CHECK(copy_field(*this, field, host, false));
    // after finish(): var.alpha = sigma * alpha, var.beta = sigma * beta  (solver_socp_inPALM.m:335-336)
    if (finished && (field == DOTSOCP_F_ALPHA || field == DOTSOCP_F_BETA)) host_scale(host, field_len(field), sigma);
    return 0;
}

// --------------------------------------------------------------------------------------
// profiling helpers
// --------------------------------------------------------------------------------------
void Solver::prof_begin(int phase, bool on_z) {
    if (!profiling) return;
    (void)use_dev(device);
    hipStream_t st = on_z ? stream_z : stream;
    Pending p;
    p.phase = phase;
    auto get = [&]() {
        hipEvent_t e;
        if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    p.a = get();
    p.b = get();
    (void)ds_event_record(p.a, st);
    pending.push_back(p);
}

void Solver::prof_end(int phase, bool on_z) {
    if (!profiling) return;
    (void)use_dev(device);
    hipStream_t st = on_z ? stream_z : stream;
    for (auto it2 = pending.rbegin(); it2 != pending.rend(); ++it2)
        if (it2->phase == phase) { (void)ds_event_record(it2->b, st); break; }
}

int Solver::prof_flush() {
    if (!profiling || pending.empty()) return 0;
    DS_CHECK(use_dev(device));
    DS_HIP(ds_stream_synchronize(stream_z));
    DS_HIP(ds_stream_synchronize(stream));
    for (auto &p : pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            phase_ms[p.phase] += ms;
            phase_launches[p.phase] += 1;
        }
        event_pool.push_back(p.a);
        event_pool.push_back(p.b);
    }
    pending.clear();
    return 0;
}

double Solver::elapsed() const {
    return elapsed_prev + std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
}

// --------------------------------------------------------------------------------------
// begin: solver_socp_inPALM.m:11-135
// --------------------------------------------------------------------------------------
void Solver::update_coef() {
    lc.s = E / D;                       // scaleBF (:58)
    lc.sf = lc.s / sqrt(2.0);
    lc.dF = E / dScale;                 // scaleD (:59,183)
    const double ht = 1.0 / (double)(nt - 1);
    lc.at = D * (1.0 / ht);             // D .* grad, entries 1/ht (initialize.m:68; solver_dotsocp2d.m:338)
    lc.ax = (nx > 1) ? D * (1.0 / (1.0 / (double)(nx - 1))) : 0.0;
    lc.ay = (ny > 1) ? D * (1.0 / (1.0 / (double)(ny - 1))) : 0.0;
    lc.tau = opts.tau;
    const double tmp = (E / D) * (E / D);   // oper_q.m:14
    if (prob.weighted) { lc.c1 = 2.0 * tmp; lc.c2 = tmp; }
    else { lc.c1 = 1.0 + 2.0 * tmp; lc.c2 = 1.0 + tmp; }
    lc.dinv1 = 1.0 / lc.c1;
    lc.dinv2 = 1.0 / lc.c2;
}

int Solver::flush_beta() {
    if (!bpend) return 0;
    FOR_SLABS(s) {
        DS_CHECK(launch_scale(s.beta, 10 * s.g.Nc, bmul, bdiv, s.st));
        if (bpend > 1) DS_CHECK(launch_scale(s.beta, 10 * s.g.Nc, bmul2, bdiv2, s.st));
    }
    bpend = 0;
    return 0;
}

// beta <- beta * mul / div, executed by the next pass that reads beta (two operations can wait)
int Solver::push_beta_op(double mul, double div) {
    if (bpend >= 2) DS_CHECK(flush_beta());
    if (bpend == 0) { bmul = mul; bdiv = div; }
    else { bmul2 = mul; bdiv2 = div; }
    bpend += 1;
    return 0;
}

int Solver::flush_alpha() {
    if (!apend) return 0;
    FOR_SLABS(s) DS_CHECK(launch_scale(s.alpha, s.g.NqAlloc, amul, adiv, s.st));
    apend = false;
    return 0;
}

// sigma update on the folded KKT path (one slab, fused dataflow): alpha, beta, c <- x / factor (solver_socp_inPALM.m:
// 312-314) without a pass over alpha or q -- beta and alpha stay as they are and are divided on load by their next
// reader (cone pass resp. q-step), and the right-hand side of the next phi-step is corrected with the r = A' alpha - c
// the q-step stored: A'(w.*q - alpha / f) + c / f = (rhs + r) - r / f.  c is divided in that same small pass.
int Solver::sigma_scale_folded(double factor) {
    DS_CHECK(flush_alpha());
    DS_CHECK(ensure_halo());
    u0_made = false;
    DS_CHECK(push_beta_op(1.0, factor));
    if (multi()) {
        // time slabs: the u0 tail for the right neighbour is formed from alpha in memory before the next q-step runs
        FOR_SLABS(s) DS_CHECK(launch_scale(s.alpha, s.g.NqAlloc, 1.0, factor, s.st));
    } else {
        apend = true; amul = 1.0; adiv = factor;
    }
    u0_fresh = false;
    FOR_SLABS(s) DS_CHECK(launch_rhs_sigma_fix(s.w0, s.w1, s.c, s.g.Nphi, factor, s.st));
    return 0;
}

int Solver::scale_state(double a_mul, double a_div, double q_div, bool with_c) {
    DS_CHECK(flush_alpha());
    if (begun) DS_CHECK(ensure_halo());
    u0_made = false;
    u0_fresh = false;      // q0 / alpha0 change: the u0 tail held by the right neighbour is stale
    rhs_valid = false;     // ... and so is the right-hand side the last q-step left in w0
    if (fused && begun) {
        // beta: applied by the next pass that reads it
        DS_CHECK(push_beta_op(a_mul, a_div));
    }
    FOR_SLABS(s) {
        const Grid &g = s.g;
        if (with_c) DS_CHECK(launch_scale(s.c, g.Nphi, a_mul, a_div, s.st));
        if (!acc_light) DS_CHECK(launch_scale(s.alpha, g.NqAlloc, a_mul, a_div, s.st));
        // (acc_light: acc-ADMM's sigma update with the extrapolating cone pass to follow, which divides beta itself)
        if (!(fused && begun) && !acc_light) DS_CHECK(launch_scale(s.beta, 10 * g.Nc, a_mul, a_div, s.st));
        if (q_div != 1.0) {
            DS_CHECK(launch_scale(s.q, g.NqAlloc, 1.0, q_div, s.st));
            // fused dataflow: a z that is not materialised is not scaled either -- it is regenerated from the scaled
            // q and beta when somebody asks for it (the kept beta^k / q^k pair of the last KKT pass no longer matches)
            if (!(fused && begun) || z_valid) DS_CHECK(launch_scale(s.z, 10 * g.Nc, 1.0, q_div, s.st));
        }
    }
    if (q_div != 1.0 && fused && begun && !z_valid) z_prev_ok = false;
    return 0;
}

int Solver::begin_method(const dotsocp_opts *o, int m, const dotsocp_acc_opts *acc) {
    if (begun) { set_error("begin() called twice"); return DOTSOCP_ESTATE; }
    if (m == DOTSOCP_METHOD_INPALM) return begin(o);
    DS_ARG(m == DOTSOCP_METHOD_ACCADMM || m == DOTSOCP_METHOD_PALM, "unknown method");
    DS_ARG(prob.dim == 2, "the reference has PALM and acc-ADMM loops for 2-D problems only");
    DS_ARG(!(m == DOTSOCP_METHOD_PALM && prob.weighted), "the reference has no weighted PALM loop");
    method = m;
    int rc = begin(o);
    if (rc != 0) { method = DOTSOCP_METHOD_INPALM; return rc; }
    return (m == DOTSOCP_METHOD_PALM) ? palm_begin() : acc_begin(acc);
}

int Solver::begin(const dotsocp_opts *o) {
    DS_ARG(o != nullptr, "opts is NULL");
    DS_ARG(o->maxit >= 0, "opts.maxit < 0");
    DS_ARG(o->sigma > 0, "opts.sigma must be positive");
    if (begun) { set_error("begin() called twice"); return DOTSOCP_ESTATE; }
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    DS_CHECK(ensure_alloc());
    if (method == DOTSOCP_METHOD_ACCADMM) {
        DS_CHECK(acc_alloc());
        fused = false;       // z is a stored state variable of this loop: the generic helpers take their "z in memory" paths
    }
    opts = *o;
    checkPrimDualFeas = (o->checkPrimDualFeas < 0) ? !prob.weighted : (o->checkPrimDualFeas != 0);   // :20-24 / wsocp :25-29
    time_limit = (o->time_limit > 0) ? o->time_limit : 3600.0;                                        // :26-30
    sigma = o->sigma;
    lastSigmaIt = -INFINITY;
    cScale = prob.cScale; dScale = prob.dScale; D = prob.D; E = prob.E;                                // :54-59
    use_feasOrg = 0;
    tol_feasOrg = 5 * o->tol;
    rescale = o->scaling ? 1 : 0;                                                                      // :64-68
    maxFeas = INFINITY; relGap = INFINITY;
    h = 1.0 / ((double)nx * (double)ny * (double)nt);                                                  // :84
    norm_c = prob.normc; norm_d = prob.normd;                                                          // :100-101
    update_coef();
    // alpha /= sigma, beta /= sigma, c /= sigma                                                       // :102-104
    DS_CHECK(scale_state(1.0, sigma, 1.0, true));
    DS_CHECK(exchange_q_halo(false));
    sigmaScale = 1.0;
    it = 0;
    stopped = false;
    deferred = false;
    z_valid = true;
    z_prev_ok = false;
    rhs_valid = false;
    last_S_it = -1;
    hist_kkt.clear(); hist_time.clear(); hist_iter.clear(); hist_gap.clear();
    for (int i = 0; i < PH_COUNT; ++i) { phase_ms[i] = 0; phase_launches[i] = 0; }
    if (canary_enabled() && getenv("DOTSOCP_CANARY_SELFTEST")) {
        // test hook: one double written right behind model.c, the way a kernel overrunning its last tile would
        Slab &s0 = slabs[0];
        DS_CHECK(use(s0));
        const double v = 1.0;
        DS_HIP(ds_memcpy_async(s0.c + s0.g.Nphi, &v, sizeof v, hipMemcpyHostToDevice, s0.st));
        DS_HIP(ds_stream_synchronize(s0.st));
    }
    begun = true;
    elapsed_prev = 0.0;
    elapsed_agreed = 0.0;
    t_begin = std::chrono::steady_clock::now();
    return 0;
}

// --------------------------------------------------------------------------------------
// the four steps of one iteration
// --------------------------------------------------------------------------------------
// phi = idctn(dctn(rhs) ./ kernel), kernel = D^2 * initialize_FFTkernel  (:96,194); rhs is in w0
int Solver::poisson_all(const PhiHooks *hooks) {
    const i64 plane = slabs[0].g.plane;
    const bool tp2 = dct_plan_has_tsolve(devres[0]->pt);
    FOR_SLABS(s) {
        const Grid &g = s.g;
        DS_CHECK(launch_dct_axis(s.res->py, s.w0, s.w1, g.ny, g.nx, g.ntl, 0, 0, s.st, g.py));
        DS_CHECK(launch_dct_axis(s.res->px, s.w1, s.w0, g.ny, g.nx, g.ntl, 1, 0, s.st, g.py));
    }
    bool tri = multi() && tri_tsolve && world <= DS_MAX_WORLD;
    for (auto &s : slabs) tri = tri && s.g.ntl <= TRI_EXTRA;
    if (tri) {
        DS_CHECK(poisson_t_tridiag(hooks));
    } else {
    if (hooks && hooks->fill) {
        prof_end(PH_POISSON);
        DS_CHECK(hooks->fill());
        prof_begin(PH_POISSON);
    }
    if (hooks && hooks->behind) DS_CHECK(hooks->behind());
    if (multi()) {       // timed on its own (inside "poisson") so that the scaling runs show what the all-to-alls cost
        prof_begin(PH_TRANSPOSE);
        DS_CHECK(transpose(true));
        prof_end(PH_TRANSPOSE);
    }
    FOR_SLABS(s) {
        double *p = multi() ? s.pencil : s.w0;
        double *p2 = multi() ? s.pencil2 : s.w1;
        if (!multi()) {
            // the single slab: rows may be pitched (common.h), the (y, x) columns of a layer are ny lines in each of nx rows
            const Grid &g = s.g;
            if (tri_tsolve && tsolve_tri_preferred(nt, dct_plan_is_pow2(s.res->pt), g.plane)) {
                // no transform along t: the (ky, kx) modes are tridiagonal systems in t (tri.hip: k_tsolve_single / _pipe);
                // DOTSOCP_TSOLVE=dct: the transform passes below for every length
                DS_CHECK(launch_tsolve_tri(g, nt, D * D, s.res->cy, s.res->cx, p, s.st));
            } else if (tp2) {
                DS_CHECK(launch_dct_t_solve(s.res->pt, p, p, ny, ny * nx, 0, ny * nx, nt, D * D, s.res->cy, s.res->cx, s.res->ct, s.st, g.py));
            } else {
                DS_CHECK(launch_dct_axis(s.res->pt, p, p2, g.ny, g.nx, nt, 2, 0, s.st, g.py));
                DS_CHECK(launch_spectral_divide(p2, ny, nx, nt, 0, nx, D * D, s.res->cy, s.res->cx, s.res->ct, s.st, g.py));
                DS_CHECK(launch_dct_axis(s.res->pt, p2, p, g.ny, g.nx, nt, 2, 1, s.st, g.py));
            }
        } else if (tp2) {
            DS_CHECK(launch_dct_t_solve(s.res->pt, p, p, s.g.py, plane, s.l0, s.nl, nt, D * D, s.res->cy, s.res->cx, s.res->ct, s.st));
        } else {
            DS_CHECK(launch_dct_axis(s.res->pt, p, p2, s.nl, 1, nt, 2, 0, s.st));
            DS_CHECK(launch_spectral_divide_pencil(p2, s.g.py, plane, s.l0, s.nl, nt, D * D, s.res->cy, s.res->cx, s.res->ct, s.st));
            DS_CHECK(launch_dct_axis(s.res->pt, p2, p, s.nl, 1, nt, 2, 1, s.st));
        }
    }
    if (multi()) {
        prof_begin(PH_TRANSPOSE);
        DS_CHECK(transpose(false));
        prof_end(PH_TRANSPOSE);
    }
    }
    FOR_SLABS(s) {
        const Grid &g = s.g;
        DS_CHECK(launch_dct_axis(s.res->px, s.w0, s.w1, g.ny, g.nx, g.ntl, 1, 1, s.st, g.py));
        DS_CHECK(launch_dct_axis(s.res->py, s.w1, s.phi, g.ny, g.nx, g.ntl, 0, 1, s.st, g.py));
    }
    return 0;
}

int Solver::phase_phi(const PhiHooks *hooks) {
    DS_CHECK(ensure_halo());
    if (multi() && !u0_fresh) {      // normally shipped with the q halo at the end of the previous iteration
        DS_CHECK(make_u0_tail());
        prof_begin(PH_COMM, comm_z);
        DS_CHECK(exchange_u0_tail());
        prof_end(PH_COMM, comm_z);
    }
    prof_begin(PH_RHS);
    if (!rhs_valid) {
        DS_CHECK(flush_alpha());
        FOR_SLABS(s) DS_CHECK(launch_rhs(s.g, lc, s.q, s.alpha, s.c, s.weight, s.u0_prev, s.w0, s.st));
    } else if (multi()) {
        // the q-step left rhs in w0; its first layer still lacks the left neighbour's last cell
        FOR_SLABS(s)
            if (!s.g.first) DS_CHECK(launch_rhs_fixup(s.g, lc, s.u0_prev, s.w0, s.st));
    }
    rhs_valid = false;
    prof_end(PH_RHS);
    prof_begin(PH_POISSON);
    DS_CHECK(poisson_all(hooks));
    prof_end(PH_POISSON);
    return 0;                        // the phi head travels with the adjoint tails (phase_z_tails)
}

// The cone pass needs q^k and beta only -- not phi^{k+1}.  Time slabs: it runs in chunks of time cells, one launch per
// chunk in ascending order (the "t + 1" cone entries of a chunk's last cell travel to the next launch through s.carry),
// and only the last chunk reads the q halo -- step() puts the others in front of the halo's arrival (part 1) and the
// last one beside the first interface exchange of the Poisson solve (part 2).
int Solver::phase_z(int part) {
    if (part != 1) DS_CHECK(ensure_halo());     // the last chunk reads the q halo (part 1 never does)
    if (!fused) {
        prof_begin(PH_PROJ);
        FOR_SLABS(s) DS_CHECK(launch_cone_proj(s.g, lc, s.q, s.beta, s.z, s.st));
        prof_end(PH_PROJ);
        return 0;
    }
    const int ph = deferred ? PH_FUSED_B : PH_FUSED_A;
    z_valid = false;          // the fused pass forms z^{k+1} in registers only
    z_prev_ok = false;        // ... and (mode B) overwrites the kept beta^{k-1}
    prof_begin(ph);
    FOR_SLABS(s) {
        FusedArgs a{};
        a.q = s.q;
        a.q2 = s.q2;
        a.sx = s.sx;
        a.sy = s.sy;
        a.beta_in = s.beta;
        set_pending(a);
        const i64 C = s.fg.chunks;
        const i64 z0 = (part == 2) ? C - 1 : 0;
        const i64 zc = (part == 0) ? C : ((part == 1) ? C - 1 : 1);
        if (deferred) {
            // beta^k = beta^{k-1} + tau (z^k - BF q^k - d) folded into this iteration's projection
            a.q_old = s.q_old;
            a.beta_out = s.beta2;
        }
        const int mode = deferred ? 1 : 0;
        if (s.carry && C > 1) {
            for (i64 z = z0; z < z0 + zc; ++z) {
                a.carry_in = (z > 0) ? s.carry : nullptr;
                a.carry_out = (z + 1 < C) ? s.carry : nullptr;
                DS_CHECK(launch_cone_fused(mode, s.g, lc, s.fg, a, s.st, z, 1));
            }
        } else {
            DS_CHECK(launch_cone_fused(mode, s.g, lc, s.fg, a, s.st, z0, zc));
        }
        if (deferred && part != 1) std::swap(s.beta, s.beta2);
    }
    prof_end(ph);
    if (part == 1) return 0;
    if (deferred) bpend = 0;          // mode B rewrote beta with the scaling applied
    return 0;
}

// time-slab mode: ship the adjoint sums of every slab's last cell to its right neighbour (main stream)
int Solver::phase_z_tails() {
    return fused ? ship_tails() : 0;
}

// adjoint sums of every slab's last cell for the first edge layer of its right neighbour (kernel, main streams)
int Solver::make_tails() {
    if (!multi()) return 0;
    FOR_SLABS(s)
        if (!s.g.last) DS_CHECK(launch_tail_finalize(s.g, lc, s.fg, s.q2, s.sx, s.sy, s.send_bx, s.send_by, s.st));
    return 0;
}

int Solver::send_tails() {
    if (!multi()) return 0;
    prof_begin(PH_COMM, comm_z);
    DS_CHECK(group_begin());
    DS_CHECK(shift(+1, [](Slab &s) { return s.send_bx; }, [](Slab &s) { return s.tail_bx; }, slabs[0].g.bxLayer));
    DS_CHECK(shift(+1, [](Slab &s) { return s.send_by; }, [](Slab &s) { return s.tail_by; }, slabs[0].g.byLayer));
    DS_CHECK(group_end());
    prof_end(PH_COMM, comm_z);
    return 0;
}

// first phi layer of every slab -> halo layer of its left neighbour (forward time difference of the q-step)
int Solver::send_phi_head() {
    if (!multi()) return 0;
    prof_begin(PH_COMM, comm_z);
    DS_CHECK(shift(-1, [](Slab &s) { return s.phi; }, [](Slab &s) { return s.phi + s.g.plane * s.g.ntl; }, slabs[0].g.plane));
    prof_end(PH_COMM, comm_z);
    return 0;
}

int Solver::ship_tails() {
    if (multi()) {
        DS_CHECK(make_tails());
        DS_CHECK(group_begin());        // one group: traffic in both directions at once
        DS_CHECK(send_phi_head());
        DS_CHECK(send_tails());
        DS_CHECK(group_end());
    }
    return 0;
}

KktCoef Solver::kkt_coef() const {
    KktCoef k;
    k.sigma = sigma;
    k.kappa = sigma * cScale * D;
    k.dsD = dScale / D;
    k.dsE = dScale / E;
    return k;
}

// kkt: the iteration ends with a KKT check and the q-step runs in its KKT variant (one slab: part == 0)
int Solver::phase_q(int part, bool kkt) {
    if (!(fused && qrhs)) DS_CHECK(flush_alpha());
    prof_begin(PH_QSTEP);
    FOR_SLABS(s) {
        hipStream_t st = s.st;
        if (!fused) {
            DS_CHECK(launch_qstep(s.g, lc, s.phi, s.z, s.beta, s.weight, s.tail_bx, s.tail_by, s.q, s.alpha, s.st));
        } else {
            // q^{k+1} goes to the buffer that held q^{k-1}; q^k is kept for the deferred beta update
            if (qrhs) {
                // ... and the right-hand side of the next phi-step is formed in the same pass (alpha ping-pongs)
                const i64 C = qstep_rhs_chunks(s.g, s.fg);
                i64 z0 = 0, zc = C, zs = 1;
                if (part == 1) { zc = C - 1; }                     // all but the last chunk
                else if (part == 2) { z0 = C - 1; zc = 1; }        // the last chunk
                QStepExtra ex{};
                ex.apend = apend ? 1 : 0; ex.amul = amul; ex.adiv = adiv;
                if (multi() && !s.g.last && part != 1) ex.u0_tail = s.send_plane;
                if (kkt) {
                    const KktCoef k = kkt_coef();
                    ex.partials = kkt_qstep_partials(s.g, s.kw);
                    ex.resid = s.w1;                   // free between the Poisson solves
                    ex.kappa = k.kappa; ex.dsD = k.dsD;
                }
                DS_CHECK(launch_qstep_rhs(s.g, lc, s.fg, s.phi, s.q2, s.sx, s.sy, s.weight, s.tail_bx, s.tail_by, s.c,
                                          s.q_old, s.alpha, s.alpha2, s.w0, st, z0, zc, zs, &ex));
                if (part != 1) std::swap(s.alpha, s.alpha2);
            } else {
                DS_CHECK(launch_qstep_fused(s.g, lc, s.fg, s.phi, s.q2, s.sx, s.sy, s.weight, s.tail_bx, s.tail_by,
                                            s.q_old, s.alpha, s.st));
            }
            if (part != 1) std::swap(s.q, s.q_old);
        }
    }
    prof_end(PH_QSTEP);
    if (part == 1) return 0;
    u0_made = multi() && fused && qrhs;
    if (fused && qrhs) apend = false;        // the q-step wrote the scaled alpha into the ping-pong partner
    rhs_valid = fused && qrhs;
    // the halo exchange waits for the next consumer: the next step() runs it beside the first cone chunks
    if (multi() && fused && comm_z && cone_split_enabled()) halo_pending = true;
    else DS_CHECK(exchange_q_halo(true));
    return 0;
}

int Solver::phase_mult() {
    if (fused) {
        deferred = true;     // the multiplier step is executed by the next fused pass (or by materialise())
        return 0;
    }
    prof_begin(PH_BETA);
    FOR_SLABS(s) DS_CHECK(launch_beta_update(s.g, lc, s.q, s.z, s.beta, s.st));
    prof_end(PH_BETA);
    return 0;
}

// Fused path only: execute the pending multiplier step and store z (solver_socp_inPALM.m:199,
// 212-215) so that beta, z are the iterates the KKT block, the rescale block and the outputs see.
int Solver::materialise() {
    if (!fused || !deferred) return 0;
    DS_CHECK(ensure_halo());
    prof_begin(PH_MATERIALISE);
    FOR_SLABS(s) {
        FusedArgs a{};
        a.q_old = s.q_old;
        a.q = s.q;
        a.beta_in = s.beta;
        a.beta_out = s.beta;
        a.z_out = s.z;
        set_pending(a);
        DS_CHECK(launch_cone_fused(2, s.g, lc, s.fg, a, s.st));
    }
    prof_end(PH_MATERIALISE);
    bpend = 0;
    deferred = false;
    z_valid = true;
    return 0;
}

// z of the last completed iteration in s.z (outputs, rescale block, unfused-style cell sums)
int Solver::ensure_z() {
    if (!fused || z_valid) return 0;
    if (deferred) return materialise();
    if (!z_prev_ok) {
        set_error("internal: z cannot be regenerated");
        return DOTSOCP_ESTATE;
    }
    DS_CHECK(ensure_halo());
    prof_begin(PH_MATERIALISE);
    FOR_SLABS(s) {
        FusedArgs a{};
        a.q_old = s.q_old;
        a.q = s.q;
        a.beta_in = s.beta2;      // beta^k, kept by the KKT pass
        a.z_out = s.z;
        a.bpend = zp_pend;
        a.bmul = zp_mul; a.bdiv = zp_div; a.bmul2 = zp_mul2; a.bdiv2 = zp_div2;
        DS_CHECK(launch_cone_fused(3, s.g, lc, s.fg, a, s.st));
    }
    prof_end(PH_MATERIALISE);
    z_valid = true;
    return 0;
}

// folded: this iteration's q-step ran in its KKT variant (phase_q(.., true)): region 0 of the partial sums holds its
// share, the cell pass adds the F*B*beta terms of the edges, and no node / edge launch follows
int Solver::kkt_sums(double *S, bool folded) {
    DS_CHECK(ensure_halo());
    const KktCoef k = kkt_coef();
    if (method == DOTSOCP_METHOD_ACCADMM && folded) {
        // acc-ADMM, one slab: the cone pass of the iteration has taken the cell sums and the F*B*beta^+ terms of every entry
        // (launch_acc_cone_kkt: regions 1-3, buffer cleared before it); left are the sums made of phi^+, q^+, alpha^+, c
        FOR_SLABS(s) DS_CHECK(launch_kkt_nodual(s.g, lc, k, s.phi, s.q, s.alpha, s.c, s.weight, s.kw, s.st));
        return reduce_sums(S);
    }
    if (!folded) {
        DS_CHECK(flush_alpha());
        // the launches below write per-workgroup partial sums into four regions; grids of different
        // shapes may use a region on different calls, so stale entries are cleared first
        FOR_SLABS(s)
            DS_HIP(ds_memset_async(s.kw.partials, 0, sizeof(double) * s.kw.maxBlocks * S_COUNT, s.st));
    }
    // ---- cell part (region 1 of the partial sums) ----
    int rest = folded ? 0 : (1 | 4 | 8);
    if (fused && deferred) {
        // pending multiplier step + cell sums in one pass; beta^k stays in beta2 so that z can be regenerated
        FOR_SLABS(s) {
            FusedArgs a{};
            a.q_old = s.q_old;
            a.q = s.q;
            a.beta_in = s.beta;
            a.beta_out = s.beta2;
            set_pending(a);
            if (folded) { a.q2 = s.q2; a.sx = s.sx; a.sy = s.sy; }      // scratch for the gather of beta on tile borders
            DS_CHECK(launch_kkt_cells_update(s.g, lc, k, s.fg, a, s.phi, s.alpha, s.weight, s.kw, s.st, folded, s.q));
            std::swap(s.beta, s.beta2);
        }
        // the kept beta^k (now in beta2) is still unscaled in memory: remember its pending op for MODE_Z
        zp_pend = bpend; zp_mul = bmul; zp_div = bdiv; zp_mul2 = bmul2; zp_div2 = bdiv2;
        bpend = 0;
        deferred = false;
        z_valid = false;
        z_prev_ok = true;
    } else {
        if (folded) { set_error("internal: folded KKT sums without a pending multiplier step"); return DOTSOCP_ESTATE; }
        DS_CHECK(ensure_z());
        DS_CHECK(flush_beta());
        rest |= 2;
    }
    if (multi()) {
        const i64 plane = slabs[0].g.plane;
        u0_made = false;                      // launch_kkt_tail reuses send_plane
        FOR_SLABS(s)
            if (!s.g.last)
                DS_CHECK(launch_kkt_tail(s.g, s.alpha, s.beta, s.weight, s.send_plane, s.send_plane2, s.send_bx, s.send_by,
                                         s.st));
        DS_CHECK(group_begin());
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_plane; }, [](Slab &s) { return s.a0_prev; }, plane));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_plane2; }, [](Slab &s) { return s.a0w_prev; }, plane));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_bx; }, [](Slab &s) { return s.btail_bx; }, slabs[0].g.bxLayer));
        DS_CHECK(shift(+1, [](Slab &s) { return s.send_by; }, [](Slab &s) { return s.btail_by; }, slabs[0].g.byLayer));
        DS_CHECK(group_end());
    }
    FOR_SLABS(s) {
        KktHalo halo{s.a0_prev, s.a0w_prev, s.btail_bx, s.btail_by};
        if (rest) DS_CHECK(launch_kkt(s.g, lc, k, s.phi, s.q, s.alpha, s.z, s.beta, s.c, s.weight, halo, s.kw, rest, s.st));
        // folded path on a slab that is not the first: its first node / edge layer, now that the left neighbour's last
        // cell has arrived (the q-step, the cell pass and the border launches skipped that layer)
        if (folded && !s.g.first)
            DS_CHECK(launch_kkt(s.g, lc, k, s.phi, s.q, s.alpha, s.z, s.beta, s.c, s.weight, halo, s.kw, 1 | 4 | 8, s.st, true,
                                s.w1));
    }
    return reduce_sums(S);
}

// The partial sums every slab holds -> S[0 .. S_COUNT) summed over the slabs (host, in slab order) resp. over the ranks
// (all-reduce); S[S_COUNT] = the wall clock (one slab per process: the maximum over the ranks)
int Solver::reduce_sums(double *S) {
    for (int i = 0; i <= S_COUNT; ++i) S[i] = 0.0;
    FOR_SLABS(s) {
        DS_CHECK(launch_kkt_final(s.g, s.kw, s.st));
        if (remote()) break;
        DS_HIP(ds_memcpy_async(s.h_sums, s.kw.sums, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, s.st));
    }
    if (!remote()) {        // all slabs are enqueued before the host waits for the first; summed in slab order
        FOR_SLABS(s) {
            DS_HIP(ds_stream_synchronize(s.st));
            for (int i = 0; i < S_COUNT; ++i) S[i] += s.h_sums[i];
        }
    }
    S[S_COUNT] = elapsed();
    if (remote()) {
        // sum the partial sums over the ranks; slot S_COUNT carries the wall clock (max via a second reduce)
        Rccl &api = rccl_api();
        Slab &s = slabs[0];
        DS_CHECK(comm_enter());
        const hipStream_t cs = cst(s);
        DS_NCCL(api.AllReduce(s.kw.sums, d_red, S_COUNT, ncclDouble, ncclSum, (ncclComm_t)nccl, cs));
        h_sums[S_COUNT] = S[S_COUNT];
        DS_HIP(ds_memcpy_async(d_red + S_COUNT, h_sums + S_COUNT, sizeof(double), hipMemcpyHostToDevice, cs));
        DS_NCCL(api.AllReduce(d_red + S_COUNT, d_red + S_COUNT, 1, ncclDouble, ncclMax, (ncclComm_t)nccl, cs));
        DS_HIP(ds_memcpy_async(h_sums, d_red, sizeof(double) * (S_COUNT + 1), hipMemcpyDeviceToHost, cs));
        DS_CHECK(comm_leave());
        DS_HIP(ds_stream_synchronize(stream));
        for (int i = 0; i <= S_COUNT; ++i) S[i] = h_sums[i];
    }
    elapsed_agreed = S[S_COUNT];
    return 0;
}

// The five norms of solver_socp_inPALM.m:140-141 for an iterate whose multiplier step is still pending (the state between
// two iterations of the fused loop): one pass over beta that stores nothing, three sums of squares.  S as from kkt_sums().
int Solver::norms_light(double *S) {
    DS_CHECK(ensure_halo());
    DS_CHECK(flush_alpha());
    const KktCoef k = kkt_coef();
    FOR_SLABS(s) {
        DS_HIP(ds_memset_async(s.kw.partials, 0, sizeof(double) * s.kw.maxBlocks * S_COUNT, s.st));
        FusedArgs a{};
        a.q_old = s.q_old;
        a.q = s.q;
        a.beta_in = s.beta;
        set_pending(a);
        DS_CHECK(launch_norms(s.g, lc, k, s.fg, a, s.phi, s.alpha, s.weight, s.kw, s.st));
    }
    return reduce_sums(S);
}

// solver_socp_inPALM.m:138-190
int Solver::rescale_block() {
    bool scaleYes = false;
    double normPhis = 0, normAlps = 0;
    auto norms = [&](double &nPhis, double &nAlps) -> int {
        double S[S_COUNT + 1];
        double sig = sigma;
        if (method == DOTSOCP_METHOD_INPALM && fused && last_S_it == it - 1 && norm_cache) {
            // The previous iteration ended with a KKT check: its sums ARE the squared norms of phi, q, z, alpha, beta
            // of the current iterate.  A sigma update in between divided alpha and beta by `factor` and multiplied
            // sigma by it, so sigma * ||alpha|| is what it was with the sigma of the check -- no pass over the state.
            for (int i = 0; i < S_COUNT; ++i) S[i] = last_S[i];
            sig = last_S_sigma;
        } else if (method == DOTSOCP_METHOD_INPALM && fused && deferred && norm_cache) {
            DS_CHECK(norms_light(S));          // nothing is materialised, the multiplier step stays pending
        } else {
            DS_CHECK(materialise());
            DS_CHECK(ensure_z());
            DS_CHECK(kkt_sums(S));
        }
        const double sh = sqrt(h);
        const double normPhi = sh * sqrt(S[S_PHI2]), normQ = sh * sqrt(S[S_Q2]), normZ = sh * sqrt(S[S_Z2]);
        const double normAlpha = sig * (sh * sqrt(S[S_ALPHA2])), normBeta = sig * (sh * sqrt(S[S_BETA2]));
        nPhis = std::max(std::max(normPhi, normQ), normZ);
        nAlps = std::max(normAlpha, normBeta);
        return 0;
    };
    if (rescale >= 3 && (it % 100) == 0) {
        DS_CHECK(norms(normPhis, normAlps));
        const double ratio = std::max(normAlps, normPhis) / std::min(normAlps, normPhis);
        if (ratio > 1.2) scaleYes = true;
    }
    const bool first = (rescale == 1) && (maxFeas < 2e-2) && (it >= 10) && (relGap < 5e-2);
    const bool second = (rescale == 2) && (maxFeas < 5e-3) && (it >= 50) && (relGap < 1e-2);
    if (!(first || second || scaleYes)) return 0;
    if (!scaleYes) DS_CHECK(norms(normPhis, normAlps));
    // A multiplier step that is still pending belongs to the OLD scaling (z^k = Pi(BF q^{k-1} + d - beta^{k-1}) with the old
    // d and the unscaled q, beta): it is executed before anything is rescaled.  (Right after a KKT check nothing is
    // pending; the every-100-iterations check took its norms without touching the state.)
    DS_CHECK(materialise());
    const double dScale2 = normPhis, cScale2 = normAlps;
    sigma = sigma * (cScale2 / dScale2);
    norm_c = norm_c / cScale2;
    if (!prob.weighted) norm_d = norm_d / dScale2;      // solver_wsocp_inPALM.m has no norm_d
    // c, alpha, beta <- x * dScale2 / cScale2^2 ; q, z <- x / dScale2
    DS_CHECK(scale_state(dScale2, cScale2 * cScale2, dScale2, true));
    if (method == DOTSOCP_METHOD_PALM) {                 // solver_socp_PALM.m:191 tmp_q = A phi is scaled: scale phi
        FOR_SLABS(s) DS_CHECK(launch_scale(s.phi, s.g.NphiAlloc, 1.0, dScale2, s.st));
    }
    dScale = dScale2 * dScale;
    cScale = cScale2 * cScale;
    sigmaScale = sigmaScale * (cScale2 / dScale2);
    update_coef();                                       // scaleD = E / dScale (:183); z2 is regenerated on the fly
    rescale += 1;
    return 0;
}

bool if_adjust_sigma(double iter, double last_iter) {   // :361-379
    const double passed = iter - last_iter;
    if (iter < 20 && passed >= 3) return true;
    if (iter < 50 && passed >= 6) return true;
    if (iter < 100 && passed >= 10) return true;
    if (iter < 200 && passed >= 15) return true;
    if (iter < 500 && passed >= 25) return true;
    return passed >= 40;
}

static const double kUpdateRule[11][2] = {   // :39-51
    {1.1, 1.10}, {1.2, 1.15}, {1.5, 1.20}, {2, 1.26}, {2.5, 1.28}, {3.33, 1.32},
    {5, 1.35}, {10, 1.40}, {20, 1.60}, {40, 1.80}, {50, 2.00}};

static double get_factor(double xi) {   // adjust_lagrangianParam.m:49-60
    double factor = 1.0;
    for (int i = 0; i < 11; ++i) {
        if (xi >= kUpdateRule[i][0]) factor = kUpdateRule[i][1];
        else break;
    }
    return factor;
}

static void adjust_lagrangian_param(double &sigma, double xi, double &factor) {   // adjust_lagrangianParam.m:14-39
    factor = 1.0;
    if (xi >= 1) factor = get_factor(xi);
    else if (xi < 1) factor = 1.0 / get_factor(1.0 / xi);
    if (factor != 1.0) {
        const double old = sigma;
        sigma = std::max(std::min(sigma * factor, 1e3), 1e-3);
        factor = sigma / old;
    }
}

// solver_socp_inPALM.m:222-323
int Solver::kkt_block(bool adjustSigmaYes, bool timed_out, bool *brk, bool folded) {
    double S[S_COUNT + 1];
    prof_begin(PH_KKT);
    DS_CHECK(kkt_sums(S, folded));
    prof_end(PH_KKT);
    // one slab per process: the ranks must take the time-limit decision together, so it is taken
    // here from the maximum of their wall clocks (time-outs are detected at KKT checks only)
    if (remote()) timed_out = elapsed_agreed > time_limit;
    for (int i = 0; i < S_COUNT; ++i) last_S[i] = S[i];
    last_S_sigma = sigma;
    last_S_it = it;
    const double sh = sqrt(h);
    auto nrm = [&](int i) { return sh * sqrt(S[i]); };
    const double norm_q = nrm(S_Q2), norm_z = nrm(S_Z2), norm_Aphi = nrm(S_APHI2);
    const double norm_alpha = sigma * nrm(S_ALPHA2), norm_beta = sigma * nrm(S_BETA2);
    const double norm_FBbeta = sigma * nrm(S_FBBETA2);
    const double primFea1 = nrm(S_PRIM1), primFea2 = nrm(S_PRIM2);
    const double dualFea1 = sigma * nrm(S_DUAL1), dualFea2 = sigma * nrm(S_DUAL2);
    const double complem = nrm(S_COMPLEM);
    const double dotcomplem = nrm(S_DOTCOMP), normRho = nrm(S_RHO2), norm_rhoFq = nrm(S_RHOFQ2);
    const double mRhoB = nrm(S_MRHOB), normM = nrm(S_M2), normRhoB = nrm(S_RHOB2);
    const double kc = 1.0;
    const double den2 = prob.weighted ? (norm_q + norm_z) : norm_d;        // wsocp :256,265
    double org[7], res[5];
    org[0] = primFea1 / (kc * D / dScale + norm_Aphi + norm_q);
    org[1] = primFea2 / (kc * E / dScale + den2);
    org[2] = dualFea1 / (kc / cScale + norm_c);
    org[3] = complem / (kc * E / dScale + norm_z + norm_beta);
    org[4] = dualFea2 / (kc / cScale / D + norm_FBbeta + norm_alpha);
    org[5] = dotcomplem / (kc + normRho + norm_rhoFq);
    org[6] = mRhoB / (kc + normM + normRhoB);
    res[0] = primFea1 / (kc + norm_Aphi + norm_q);
    res[1] = primFea2 / (kc + den2);
    res[2] = dualFea1 / (kc + norm_c);
    res[3] = complem / (kc + norm_z + norm_beta);
    res[4] = dualFea2 / (kc + norm_FBbeta + norm_alpha);
    const double priVal = (sigma * cScale * dScale * h) * S[S_QALPHA];
    const double dualVal = (sigma * cScale * dScale * h) * S[S_CPHI];
    const double pdGap = fabs(priVal - dualVal) / (1 + fabs(priVal) + fabs(dualVal));
    for (int i = 0; i < 7; ++i) hist_kkt.push_back(org[i]);
    hist_time.push_back(remote() ? elapsed_agreed : elapsed());
    hist_iter.push_back((double)it);
    hist_gap.push_back(pdGap);
    // stop criterion (:287-290); stopCondition = [1,3,6,7] or [1,3,6] (:117-121)
    double mstop = std::max(std::max(org[0], org[2]), org[5]);
    if (checkPrimDualFeas) mstop = std::max(mstop, org[6]);
    if (mstop < opts.tol || timed_out) {
        *brk = true;
        return 0;
    }
    const double maxRes = *std::max_element(res, res + 5);
    if (maxRes < tol_feasOrg) use_feasOrg = 1;                              // :293-295
    if (adjustSigmaYes) {                                                   // :298-316
        lastSigmaIt = (double)it;
        double resiPri, resiDual;
        if (use_feasOrg) { resiPri = std::max(org[0], org[1]); resiDual = std::max(org[2], org[4]); }
        else { resiPri = std::max(res[0], res[1]); resiDual = std::max(res[2], res[4]); }
        double factor;
        adjust_lagrangian_param(sigma, resiPri / resiDual, factor);
        if (factor != 1.0) {
            if (folded && rhs_valid && method != DOTSOCP_METHOD_ACCADMM) DS_CHECK(sigma_scale_folded(factor));
            else DS_CHECK(scale_state(1.0, factor, 1.0, true));
            if (method == DOTSOCP_METHOD_ACCADMM) DS_CHECK(acc_on_sigma_factor(factor));
        }
    }
    if (rescale > 0) {                                                      // :319-322
        maxFeas = maxRes;
        relGap = pdGap;
    }
    return 0;
}

int Solver::step(bool *brk) {
    *brk = false;
    if (method == DOTSOCP_METHOD_ACCADMM) return acc_step(brk);
    if (method == DOTSOCP_METHOD_PALM) return palm_step(brk);
    it += 1;
    DS_CHECK(rescale_block());
    const bool adjustSigmaYes = if_adjust_sigma((double)it, lastSigmaIt);                  // :220
    // known before the q-step (the time limit is the one trigger that is not: such a check takes the unfolded path)
    const bool kkt_due = opts.ifCheckStepByStep || adjustSigmaYes || it == opts.maxit;
    // fused dataflow: the q-step of a checking iteration accumulates its share of the KKT sums itself
    const bool fold = kkt_due && kkt_fold && fused && qrhs;
    // Time slabs, messages on the second streams (solver.h: comm_z): kernels on the main streams in an order that leaves
    // every message time to travel while kernels that do not need it run.
    const bool inter = comm_z && multi() && fused && qrhs;
    bool split = inter && cone_split_enabled();
    for (auto &s : slabs) split = split && s.fg.chunks >= 2;
    bool split_q = inter && cone_split_enabled();
    for (auto &s : slabs) split_q = split_q && qstep_rhs_chunks(s.g, s.fg) >= 2;
    // an exchange issued on the second streams without the join: fork, messages, mark
    auto async_comm = [&](const std::function<int()> &fn, hipEvent_t Slab::*ev) -> int {
        DS_CHECK(comm_fork());
        comm_async = true;
        const int rc = fn();
        comm_async = false;
        DS_CHECK(rc);
        return comm_mark(ev);
    };
    if (inter) {
        // [E2 + E1] the q halo and the u0 tail of the last q-step travel while the cone chunks in front of the last one
        // -- which alone reads the halo -- run
        const bool pend = halo_pending;
        if (pend) {
            DS_CHECK(make_u0_tail());
            DS_CHECK(async_comm([&]() { return ensure_halo(); }, &Slab::ev_halo));
        }
        if (split) DS_CHECK(phase_z(1));
        if (pend) DS_CHECK(comm_wait(&Slab::ev_halo));
        if (!split) DS_CHECK(phase_z(0));
        // The phi-step.  The last cone chunk (and the finalising of its adjoint tails) runs while the second streams carry
        // the latency-bound middle of the Poisson solve (poisson_t_tridiag); the tails [E4] leave right behind that and
        // travel beside the rest of the solve and the front of the q-step
        PhiHooks hooks;
        hooks.fill = [&]() -> int {
            if (split) DS_CHECK(phase_z(2));
            return make_tails();
        };
        hooks.behind = [&]() -> int { return async_comm([&]() { return send_tails(); }, &Slab::ev_join); };
        DS_CHECK(phase_phi(&hooks));
    } else {
        DS_CHECK(phase_phi());
        DS_CHECK(phase_z(0));
    }
    if (fold) {
        // every region of partial sums is cleared before the q-step writes region 0; kkt_sums() then only adds the cell,
        // border and (time slabs) first-layer launches
        FOR_SLABS(s) DS_HIP(ds_memset_async(s.kw.partials, 0, sizeof(double) * s.kw.maxBlocks * S_COUNT, s.st));
    }
    if (inter) {
        // [E3] the phi head travels (behind the tails) while every chunk of the q-step but the last -- the only reader of
        // the phi halo -- runs; the tails (read by the first chunk) have had the second half of the phi-step to arrive
        DS_CHECK(async_comm([&]() { return send_phi_head(); }, &Slab::ev_halo));
        DS_CHECK(comm_wait(&Slab::ev_join));
        if (split_q) DS_CHECK(phase_q(1, fold));
        DS_CHECK(comm_wait(&Slab::ev_halo));
        DS_CHECK(phase_q(split_q ? 2 : 0, fold));
    } else {
        DS_CHECK(phase_z_tails());
        DS_CHECK(phase_q(0, fold));
    }
    DS_CHECK(phase_mult());
    // with one slab per process a per-rank clock could split the ranks: see kkt_block()
    const bool timed_out = remote() ? false : (elapsed() > time_limit);
    if (kkt_due || timed_out)                                                             // :221
        DS_CHECK(kkt_block(adjustSigmaYes, timed_out, brk, fold));
    return 0;
}

int Solver::run(i64 n_iters, i64 *done) {
    if (!begun || finished) { set_error("run() needs begin() and must precede finish()"); return DOTSOCP_ESTATE; }
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    // while the loop runs, whatever enqueues work on a slab's streams is issued by that slab's own host thread (defer.h)
    struct DeferScope {
        DeferCtx *ctx;
        int err = 0;
        explicit DeferScope(DeferCtx *c) : ctx(c) { if (ctx) { g_defer = ctx; ctx->begin(); } }
        int close() { if (ctx) { err = ctx->end(); g_defer = nullptr; ctx = nullptr; } return err; }
        ~DeferScope() { (void)close(); }
    } scope(defer.get());
    i64 n = 0;
    while (it < opts.maxit && !stopped) {
        if (n_iters >= 0 && n >= n_iters) break;
        bool brk = false;
        DS_CHECK(step(&brk));
        if (brk) stopped = true;
        ++n;
    }
    DS_CHECK(ensure_halo());          // callers between run() calls see exchanged halos
    DS_CHECK(sync_all());
    DS_CHECK(prof_flush());
    if (const int derr = scope.close()) {
        set_error("a HIP call issued by a slab thread failed: %s", hipGetErrorString((hipError_t)derr));
        return DOTSOCP_EHIP;
    }
    if (done) *done = n;
    return 0;
}

int Solver::finish(dotsocp_result *res) {
    if (!begun) { set_error("finish() before begin()"); return DOTSOCP_ESTATE; }
    cur_dev = -1;
    DS_CHECK(use_dev(device));
    DS_CHECK(ensure_z());
    DS_CHECK(flush_beta());
    DS_CHECK(flush_alpha());
    DS_CHECK(sync_all());
    DS_CHECK(prof_flush());
    if (canary_enabled()) {        // DOTSOCP_CANARY=1: no kernel of this solve wrote outside its buffers
        std::string rep;
        const int bad = canary_check(&rep);
        cur_dev = -1;
        if (bad) {
            set_error("canary: %d device buffer(s) written out of bounds: %s", bad, rep.c_str());
            return DOTSOCP_EHIP;
        }
    }
    finished = true;
    if (res) {
        memset(res, 0, sizeof *res);
        res->sigma = sigma / sigmaScale;        // :357
        res->sigma_internal = sigma;
        res->cScale = cScale;
        res->dScale = dScale;
        // device time per step (HIP events) when profiling is on; Total_Time is host wall time
        res->times[0] = (phase_ms[PH_RHS] + phase_ms[PH_POISSON]) * 1e-3;
        res->times[1] = (phase_ms[PH_PROJ] + phase_ms[PH_FUSED_A] + phase_ms[PH_FUSED_B]) * 1e-3;
        res->times[2] = phase_ms[PH_QSTEP] * 1e-3;
        res->times[3] = (phase_ms[PH_BETA] + phase_ms[PH_MATERIALISE]) * 1e-3;
        res->times[4] = phase_ms[PH_KKT] * 1e-3;
        res->times[5] = elapsed();
        res->times[6] = (double)it;
        res->iters = it;
        res->hist_len = (i64)hist_iter.size();
        res->stopped = stopped ? 1 : 0;
        if (method == DOTSOCP_METHOD_PALM) res->time_extra = phase_ms[PH_QSTEP0] * 1e-3;   // 'Step_1_Q_Step'
        if (method == DOTSOCP_METHOD_ACCADMM) {
            // Step_1_Q_Step, Step_2_Multiplier (folded into the cone pass), Step_3_1_FFT, Step_3_2_ProjSOC, KKT, Interp
            res->times[0] = (phase_ms[PH_RHS] + phase_ms[PH_POISSON]) * 1e-3;
            res->times[1] = phase_ms[PH_ACC_CONE] * 1e-3;
            res->times[2] = (phase_ms[PH_QSTEP] + phase_ms[PH_ACC_GATHER]) * 1e-3;
            res->times[3] = 0.0;
            res->time_extra = phase_ms[PH_INTERP] * 1e-3;
        }
    }
    return 0;
}

}  // namespace dotsocp
