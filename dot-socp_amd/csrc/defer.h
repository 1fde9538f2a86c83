// One host thread per time slab for the single-process multi-slab modes (dotsocp_create_multi = opts.ngpu of the MEX
// gateways, dotsocp_create(..., nslabs)).  The reference calls the loop synchronously from ONE interpreter thread
// (solver_dotsocp2d.m:208), so the solver's host logic stays on the caller's thread -- but with P slabs that thread
// issued P x 30 launches per iteration one after the other (0.92 ms per iteration at 8 slabs, half of a rank's kernel
// time at N = 8).  While Solver::run() is active every operation that ENQUEUES work on a slab's streams -- kernel
// launches, device copies, memsets, event records, stream waits -- is handed, as a closure, to that slab's worker
// thread; the caller's thread only records them and moves on to the next slab.
//
// Ordering.  A worker executes its closures in the order they were recorded, so everything on one slab's streams
// keeps its order.  Across slabs the only ordering the solver uses is events; for them the host order matters (a
// hipStreamWaitEvent binds to the LAST hipEventRecord the host has issued): a wait is held back until the record
// that preceded it in program order has been executed by the other worker, and a record is held back until every
// wait on that event that preceded it has been executed -- the two counters per event below.  An operation only
// ever waits for operations recorded before it, and every worker runs in recording order, so the earliest
// unexecuted operation can always run: no deadlock.  Host-blocking calls (stream synchronise, the read-back of the
// KKT sums) first drain the worker of the stream they name.
//
// Outside Solver::run(), and whenever the layer is off (one slab, one process per GPU, DOTSOCP_HOST_THREADS=0), every
// wrapper below is the plain HIP call on the caller's thread.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace dotsocp {

struct DeferEvent {
    std::atomic<unsigned long long> rec_done{0}, wait_done{0};
    unsigned long long rec_enq = 0, wait_enq = 0;      // caller's thread only
};

struct DeferOp {
    std::function<hipError_t()> fn;
    DeferEvent *ev = nullptr;
    int kind = 0;                          // 0: plain, 1: event record, 2: stream wait
    unsigned long long need = 0;           // record: waits that must have executed; wait: record generation needed
    unsigned long long gen = 0;            // record: its generation
};

struct DeferCtx;
struct DeferWorker {
    static constexpr unsigned RING = 8192;
    DeferCtx *ctx = nullptr;
    std::vector<DeferOp> ring;
    std::atomic<unsigned long long> head{0}, tail{0};   // head: recorded, tail: executed
    std::atomic<bool> stop{false};
    std::atomic<int> first_error{0};
    int device = 0;
    std::thread th;
    DeferWorker() : ring(RING) {}
};

struct DeferCtx {
    std::vector<std::unique_ptr<DeferWorker>> workers;
    std::map<hipStream_t, int> stream_worker;           // streams of the slabs -> worker index
    std::map<hipEvent_t, std::unique_ptr<DeferEvent>> events;
    bool active = false;                                // between begin() and end(): closures are recorded
    // outside run() the workers sleep on this condition (no polling while a context sits idle)
    std::mutex park_mu;
    std::condition_variable park_cv;
    std::atomic<bool> awake{false};

    ~DeferCtx();
    int add_worker(int device);                         // returns its index
    void map_stream(hipStream_t st, int worker) { stream_worker[st] = worker; }
    void begin();                                       // Solver::run(): start recording
    int end();                                          // drain every worker, stop recording; first error of a closure or 0
    int drain(hipStream_t st);                          // wait until the worker of `st` has executed everything recorded
    int drain_all();
    DeferWorker *worker_of(hipStream_t st) {
        auto it = stream_worker.find(st);
        return it == stream_worker.end() ? nullptr : workers[it->second].get();
    }
    void push(DeferWorker *w, DeferOp &&op);
    DeferEvent *event(hipEvent_t e) {
        auto &p = events[e];
        if (!p) p.reset(new DeferEvent());
        return p.get();
    }
};

// the context whose run() is active on THIS host thread (nullptr: every wrapper is the plain HIP call)
extern thread_local DeferCtx *g_defer;

template <class F>
inline void defer_or_run(hipStream_t st, F &&f) {
    if (g_defer && g_defer->active) {
        if (DeferWorker *w = g_defer->worker_of(st)) {
            DeferOp op;
            op.fn = [f]() mutable -> hipError_t { f(); return hipGetLastError(); };
            g_defer->push(w, std::move(op));
            return;
        }
    }
    f();
}

// kernel launch: hipLaunchKernelGGL's arguments, recorded for the worker of the stream's slab
#define DS_KLAUNCH(K, G, B, L, S, ...) \
    ::dotsocp::defer_or_run((S), [=]() { hipLaunchKernelGGL(K, G, B, L, S, __VA_ARGS__); })

hipError_t ds_memcpy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
hipError_t ds_memcpy_peer_async(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t st);
hipError_t ds_memcpy2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                             hipMemcpyKind kind, hipStream_t st);
hipError_t ds_memset_async(void *dst, int value, size_t bytes, hipStream_t st);
hipError_t ds_event_record(hipEvent_t e, hipStream_t st);
hipError_t ds_stream_wait_event(hipStream_t st, hipEvent_t e, unsigned flags);
hipError_t ds_stream_synchronize(hipStream_t st);

}  // namespace dotsocp
