/* TEST INFRASTRUCTURE ONLY -- a stand-in for librccl that moves messages between PROCESSES THAT SHARE
 * ONE GPU through POSIX shared memory, so that the one-process-per-GPU code path of libdotsocp
 * (dot-socp_amd/csrc/solver.hip: shift(), transpose(), the KKT all-reduce) can be executed with a real
 * world size > 1 on a single-GPU test box (RCCL itself refuses several ranks on one device).
 * Selected with DOTSOCP_RCCL_LIB=<this .so>; never used by the product.
 *
 * Semantics kept: point-to-point operations between a pair of ranks match in call order; operations
 * inside ncclGroupStart/End are issued together (all sends, then all receives, so a group never
 * deadlocks); everything is ordered after prior work of the given stream (the stream is drained).
 * Build: hipcc -shared -fPIC -o libfake_rccl.so fake_rccl.cpp  (tests/test_gpu_multiprocess.py does it). */
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define MAXR 16
struct ncclComm {
    int rank, nranks;
    char tag[40];
    unsigned long sent[MAXR], rcvd[MAXR];
};

typedef struct { int is_send; void *buf; size_t bytes; int peer; struct ncclComm *comm; hipStream_t st; } op_t;
static op_t g_ops[4096];
static int g_nops = 0, g_depth = 0;

static void msg_name(char *out, struct ncclComm *c, int src, int dst, unsigned long k) {
    snprintf(out, 128, "/dsf_%s_%d_%d_%lu", c->tag, src, dst, k);
}

static int do_send(op_t *o) {
    struct ncclComm *c = o->comm;
    char name[128];
    msg_name(name, c, c->rank, o->peer, c->sent[o->peer]++);
    char tmp[140];
    snprintf(tmp, sizeof tmp, "%s.tmp", name);
    int fd = shm_open(tmp, O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0) return 1;
    size_t total = o->bytes + 16;
    if (ftruncate(fd, (off_t)total) != 0) return 1;
    char *p = (char *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) return 1;
    *(size_t *)p = o->bytes;
    if (hipStreamSynchronize(o->st) != hipSuccess) return 1;
    if (o->bytes && hipMemcpy(p + 16, o->buf, o->bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    munmap(p, total);
    close(fd);
    /* publish atomically: the receiver only ever sees complete messages */
    char a[160], b[160];
    snprintf(a, sizeof a, "/dev/shm%s", tmp);
    snprintf(b, sizeof b, "/dev/shm%s", name);
    return rename(a, b) != 0;
}

static int do_recv(op_t *o) {
    struct ncclComm *c = o->comm;
    char name[128];
    msg_name(name, c, o->peer, c->rank, c->rcvd[o->peer]++);
    int fd = -1;
    for (long spin = 0; spin < 600000; ++spin) {          /* up to ~60 s */
        fd = shm_open(name, O_RDWR, 0600);
        if (fd >= 0) break;
        struct timespec ts = {0, 100000};
        nanosleep(&ts, NULL);
    }
    if (fd < 0) { fprintf(stderr, "fake_rccl: rank %d timed out waiting for %s\n", c->rank, name); return 1; }
    struct stat sb;
    fstat(fd, &sb);
    char *p = (char *)mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) return 1;
    size_t bytes = *(size_t *)p;
    if (bytes != o->bytes) { fprintf(stderr, "fake_rccl: size mismatch on %s: sent %zu, expected %zu\n", name, bytes, o->bytes); return 1; }
    if (hipStreamSynchronize(o->st) != hipSuccess) return 1;
    if (bytes && hipMemcpy(o->buf, p + 16, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    munmap(p, (size_t)sb.st_size);
    close(fd);
    shm_unlink(name);
    return 0;
}

static ncclResult_t flush_ops(void) {
    int bad = 0;
    for (int i = 0; i < g_nops; ++i) if (g_ops[i].is_send) bad |= do_send(&g_ops[i]);
    for (int i = 0; i < g_nops; ++i) if (!g_ops[i].is_send) bad |= do_recv(&g_ops[i]);
    g_nops = 0;
    return bad ? ncclSystemError : ncclSuccess;
}

static ncclResult_t enqueue(int is_send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble || peer < 0 || peer >= comm->nranks || g_nops >= 4096) return ncclInvalidArgument;
    op_t o = {is_send, buf, count * 8, peer, comm, st};
    g_ops[g_nops++] = o;
    return g_depth ? ncclSuccess : flush_ops();
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "%x%lx", (unsigned)getpid(), (unsigned long)time(NULL));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks > MAXR) return ncclInvalidArgument;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    c->rank = rank;
    c->nranks = nranks;
    /* the id may come from the real librccl (a test process that loaded it earlier hands out ITS unique ids): arbitrary
     * bytes, '/' included -- the shared-memory names take a hex digest of the first 16 of them */
    for (int i = 0; i < 16; ++i) snprintf(c->tag + 2 * i, 3, "%02x", (unsigned)(unsigned char)id.internal[i]);
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return (--g_depth == 0) ? flush_ops() : ncclSuccess; }
ncclResult_t ncclSend(const void *b, size_t n, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) { return enqueue(1, (void *)b, n, dt, peer, c, s); }
ncclResult_t ncclRecv(void *b, size_t n, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) { return enqueue(0, b, n, dt, peer, c, s); }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "fake_rccl failure"; }

/* every rank sends its vector to every rank and reduces in rank order: identical results everywhere */
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t st) {
    if (dt != ncclDouble || (op != ncclSum && op != ncclMax) || g_depth) return ncclInvalidArgument;
    double *stage = NULL;
    if (hipMalloc((void **)&stage, count * 8 * comm->nranks) != hipSuccess) return ncclSystemError;
    ncclGroupStart();
    for (int r = 0; r < comm->nranks; ++r) {
        ncclSend(sendbuff, count, dt, r, comm, st);
        ncclRecv(stage + (size_t)r * count, count, dt, r, comm, st);
    }
    ncclResult_t rc = ncclGroupEnd();
    double *h = (double *)malloc(count * 8 * comm->nranks), *acc = (double *)malloc(count * 8);
    hipMemcpy(h, stage, count * 8 * comm->nranks, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < count; ++i) {
        double v = h[i];
        for (int r = 1; r < comm->nranks; ++r) {
            double w = h[(size_t)r * count + i];
            v = (op == ncclSum) ? v + w : (w > v ? w : v);
        }
        acc[i] = v;
    }
    hipMemcpy(recvbuff, acc, count * 8, hipMemcpyHostToDevice);
    free(h); free(acc); hipFree(stage);
    return rc;
}
