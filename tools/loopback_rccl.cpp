/* MEASUREMENT INFRASTRUCTURE ONLY -- a stand-in for librccl in which ONE rank talks to itself: every ncclRecv of a
 * group is served, as an asynchronous device-to-device copy on the caller's stream, from the ncclSend of the same
 * group that has the same byte count (in call order), and ncclAllReduce(sum) returns N times the send buffer (N ranks
 * holding the same partial sums).  Like librccl, which runs ALL sends and receives of a group in one kernel launch, the
 * copies of a group go out as ONE launch per stream (k_copy_group: up to 16 messages per launch).
 * `bench.py --rank-share N` uses it to run ONE rank's slab of an N-way time-slab split on a single GPU with exactly
 * the kernels, launch sequence, stream structure and message sizes of a real rank -- "neighbour messages as local
 * copies" -- to measure what bounds the N-GPU scaling curve before any byte crosses xGMI.
 * The iterates are NOT a valid solve (the slab sees itself as both neighbours); only the timing is meaningful.
 * Selected with DOTSOCP_RCCL_LIB=<this .so>; never used by the product or by parity tests.
 * Build: hipcc -shared -fPIC --offload-arch=gfx950 -o libloopback_rccl.so loopback_rccl.cpp   (__graft_entry__.build() does it). */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ncclComm {
    int rank, nranks;
};

typedef struct { int is_send; void *buf; size_t bytes; hipStream_t st; int used; int peer; } op_t;

/* out[i] = f * in[i]: the sum over N ranks that all hold this rank's numbers */
__global__ void k_scaled_copy(double *out, const double *in, size_t n, double f) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f * in[i];
}
/* all device-to-device messages of one group on one stream: message m = blockIdx.y */
#define LB_MAX 16
struct copy_group { const double *src[LB_MAX]; double *dst[LB_MAX]; size_t n[LB_MAX]; int count; };
__global__ void __launch_bounds__(256) k_copy_group(copy_group g) {
    const int m = blockIdx.y;
    const double2 *__restrict__ s = (const double2 *)g.src[m];
    double2 *__restrict__ d = (double2 *)g.dst[m];
    const size_t n2 = g.n[m] / 2;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) d[i] = s[i];
    if ((g.n[m] & 1) && blockIdx.x == 0 && threadIdx.x == 0) g.dst[m][g.n[m] - 1] = g.src[m][g.n[m] - 1];
}
static int launch_group(const copy_group &g, hipStream_t st) {
    if (g.count <= 0) return 0;
    size_t nmax = 0;
    for (int i = 0; i < g.count; ++i) nmax = g.n[i] > nmax ? g.n[i] : nmax;
    size_t bx = (nmax / 2 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k_copy_group, dim3((unsigned)bx, (unsigned)g.count), dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
static op_t g_ops[8192];
static int g_nops = 0, g_depth = 0;

static ncclResult_t flush_ops(void) {
    int bad = 0;
    copy_group cg;
    cg.count = 0;
    hipStream_t cg_st = nullptr;
    for (int i = 0; i < g_nops; ++i) {
        if (g_ops[i].is_send) continue;
        op_t *r = &g_ops[i];
        if (r->bytes == 8) {
            /* the one-double message of the attach handshake carries the sender's rank (Solver::attach_rccl) */
            double v = (double)r->peer;
            if (hipMemcpyAsync(r->buf, &v, 8, hipMemcpyHostToDevice, r->st) != hipSuccess) bad = 1;
            if (hipStreamSynchronize(r->st) != hipSuccess) bad = 1;      /* v lives on this stack frame */
            continue;
        }
        int src = -1;
        for (int j = 0; j < g_nops && src < 0; ++j)            /* first unused send of the same size ... */
            if (g_ops[j].is_send && !g_ops[j].used && g_ops[j].bytes == r->bytes) src = j;
        for (int j = 0; j < g_nops && src < 0; ++j)            /* ... else any send that is long enough */
            if (g_ops[j].is_send && g_ops[j].bytes >= r->bytes) src = j;
        if (src < 0) {
            if (r->bytes && hipMemsetAsync(r->buf, 0, r->bytes, r->st) != hipSuccess) bad = 1;
            continue;
        }
        g_ops[src].used = 1;
        if (!r->bytes || r->buf == g_ops[src].buf) continue;
        const bool aligned = (((size_t)r->buf | (size_t)g_ops[src].buf) & 15) == 0;
        if (!aligned) {
            if (hipMemcpyAsync(r->buf, g_ops[src].buf, r->bytes, hipMemcpyDeviceToDevice, r->st) != hipSuccess) bad = 1;
            continue;
        }
        if (cg.count == LB_MAX || (cg.count && cg_st != r->st)) { bad |= launch_group(cg, cg_st); cg.count = 0; }
        cg_st = r->st;
        cg.src[cg.count] = (const double *)g_ops[src].buf; cg.dst[cg.count] = (double *)r->buf; cg.n[cg.count] = r->bytes / 8;
        ++cg.count;
    }
    bad |= launch_group(cg, cg_st);
    g_nops = 0;
    return bad ? ncclSystemError : ncclSuccess;
}

static ncclResult_t enqueue(int is_send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble || peer < 0 || peer >= comm->nranks || g_nops >= 8192) return ncclInvalidArgument;
    op_t o = {is_send, buf, count * 8, st, 0, peer};
    g_ops[g_nops++] = o;
    return g_depth ? ncclSuccess : flush_ops();
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    (void)id;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    c->rank = rank;
    c->nranks = nranks;
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return (--g_depth == 0) ? flush_ops() : ncclSuccess; }
ncclResult_t ncclSend(const void *b, size_t n, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) { return enqueue(1, (void *)b, n, dt, peer, c, s); }
ncclResult_t ncclRecv(void *b, size_t n, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) { return enqueue(0, b, n, dt, peer, c, s); }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "loopback_rccl failure"; }
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble) return ncclInvalidArgument;
    const double f = (op == ncclSum) ? (double)comm->nranks : 1.0;      /* N ranks holding the same partial sums */
    hipLaunchKernelGGL(k_scaled_copy, dim3(1), dim3(64), 0, st, (double *)recvbuff, (const double *)sendbuff, count, f);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclSystemError;
}
}
