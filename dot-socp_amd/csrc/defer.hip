// Worker threads of the single-process multi-slab modes (defer.h).
#include "defer.h"

#include <chrono>

namespace dotsocp {

thread_local DeferCtx *g_defer = nullptr;

static inline void cpu_relax(unsigned &spins) {
    // short waits spin (a worker usually waits for microseconds); long ones give the core away
    if (++spins < 2000) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    } else if (spins < 20000) {
        std::this_thread::yield();
    } else {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

static void worker_main(DeferWorker *w) {
    (void)hipSetDevice(w->device);
    unsigned idle = 0;
    for (;;) {
        const unsigned long long t = w->tail.load(std::memory_order_relaxed);
        if (t == w->head.load(std::memory_order_acquire)) {
            if (w->stop.load(std::memory_order_acquire)) return;
            if (!w->ctx->awake.load(std::memory_order_acquire)) {
                // the layer is off (outside Solver::run()): sleep until begin() or the destructor says otherwise
                std::unique_lock<std::mutex> lk(w->ctx->park_mu);
                w->ctx->park_cv.wait(lk, [&]() { return w->ctx->awake.load(std::memory_order_acquire) || w->stop.load(std::memory_order_acquire); });
                idle = 0;
                continue;
            }
            cpu_relax(idle);
            continue;
        }
        idle = 0;
        DeferOp &op = w->ring[t % DeferWorker::RING];
        unsigned spins = 0;
        if (op.kind == 1) {          // record: every wait on this event recorded before it has been executed
            while (op.ev->wait_done.load(std::memory_order_acquire) < op.need) cpu_relax(spins);
        } else if (op.kind == 2) {   // wait: the record that preceded it has been executed
            while (op.ev->rec_done.load(std::memory_order_acquire) < op.need) cpu_relax(spins);
        }
        const hipError_t e = op.fn();
        if (e != hipSuccess) {
            int expected = 0;
            w->first_error.compare_exchange_strong(expected, (int)e);
        }
        if (op.kind == 1) op.ev->rec_done.store(op.gen, std::memory_order_release);
        else if (op.kind == 2) op.ev->wait_done.fetch_add(1, std::memory_order_acq_rel);
        op.fn = nullptr;             // release what the closure holds before the slot is reused
        w->tail.store(t + 1, std::memory_order_release);
    }
}

int DeferCtx::add_worker(int device) {
    std::unique_ptr<DeferWorker> w(new DeferWorker());
    w->device = device;
    w->ctx = this;
    w->th = std::thread(worker_main, w.get());
    workers.push_back(std::move(w));
    return (int)workers.size() - 1;
}

DeferCtx::~DeferCtx() {
    for (auto &w : workers) w->stop.store(true, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(park_mu);
    }
    park_cv.notify_all();
    for (auto &w : workers)
        if (w->th.joinable()) w->th.join();
}

void DeferCtx::push(DeferWorker *w, DeferOp &&op) {
    const unsigned long long h = w->head.load(std::memory_order_relaxed);
    unsigned spins = 0;
    while (h - w->tail.load(std::memory_order_acquire) >= DeferWorker::RING) cpu_relax(spins);      // ring full
    w->ring[h % DeferWorker::RING] = std::move(op);
    w->head.store(h + 1, std::memory_order_release);
}

void DeferCtx::begin() {
    // events recorded while the layer was off are complete facts: a wait on them needs no worker's record
    for (auto &kv : events) {
        DeferEvent &e = *kv.second;
        e.rec_enq = e.wait_enq = 0;
        e.rec_done.store(0);
        e.wait_done.store(0);
    }
    for (auto &w : workers) w->first_error.store(0);
    active = true;
    {
        std::lock_guard<std::mutex> lk(park_mu);
        awake.store(true, std::memory_order_release);
    }
    park_cv.notify_all();
}

int DeferCtx::drain_all() {
    int err = 0;
    for (auto &w : workers) {
        unsigned spins = 0;
        while (w->tail.load(std::memory_order_acquire) != w->head.load(std::memory_order_acquire)) cpu_relax(spins);
        if (!err) err = w->first_error.load();
    }
    return err;
}

int DeferCtx::drain(hipStream_t st) {
    DeferWorker *w = worker_of(st);
    if (!w) return 0;
    unsigned spins = 0;
    while (w->tail.load(std::memory_order_acquire) != w->head.load(std::memory_order_acquire)) cpu_relax(spins);
    return w->first_error.load();
}

int DeferCtx::end() {
    const int err = drain_all();
    active = false;
    awake.store(false, std::memory_order_release);      // the workers find their rings empty and park
    return err;
}

// ---------------------------------------------------------------------------------------------------------------
static inline DeferWorker *recording(hipStream_t st) {
    return (g_defer && g_defer->active) ? g_defer->worker_of(st) : nullptr;
}

hipError_t ds_memcpy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    if (DeferWorker *w = recording(st)) {
        DeferOp op;
        op.fn = [=]() { return hipMemcpyAsync(dst, src, bytes, kind, st); };
        g_defer->push(w, std::move(op));
        return hipSuccess;
    }
    return hipMemcpyAsync(dst, src, bytes, kind, st);
}

hipError_t ds_memcpy_peer_async(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t st) {
    if (DeferWorker *w = recording(st)) {
        DeferOp op;
        op.fn = [=]() { return hipMemcpyPeerAsync(dst, ddev, src, sdev, bytes, st); };
        g_defer->push(w, std::move(op));
        return hipSuccess;
    }
    return hipMemcpyPeerAsync(dst, ddev, src, sdev, bytes, st);
}

hipError_t ds_memcpy2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                             hipMemcpyKind kind, hipStream_t st) {
    if (DeferWorker *w = recording(st)) {
        DeferOp op;
        op.fn = [=]() { return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, st); };
        g_defer->push(w, std::move(op));
        return hipSuccess;
    }
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, st);
}

hipError_t ds_memset_async(void *dst, int value, size_t bytes, hipStream_t st) {
    if (DeferWorker *w = recording(st)) {
        DeferOp op;
        op.fn = [=]() { return hipMemsetAsync(dst, value, bytes, st); };
        g_defer->push(w, std::move(op));
        return hipSuccess;
    }
    return hipMemsetAsync(dst, value, bytes, st);
}

hipError_t ds_event_record(hipEvent_t e, hipStream_t st) {
    if (DeferWorker *w = recording(st)) {
        DeferEvent *ev = g_defer->event(e);
        DeferOp op;
        op.kind = 1;
        op.ev = ev;
        op.need = ev->wait_enq;              // every wait recorded so far binds to an EARLIER record of this event
        op.gen = ++ev->rec_enq;
        op.fn = [=]() { return hipEventRecord(e, st); };
        g_defer->push(w, std::move(op));
        return hipSuccess;
    }
    // a record on a stream without a worker, issued while other streams record closures: nothing of this event is
    // pending on a worker unless it was recorded through one before -- then its waiters have to be let through first
    if (g_defer && g_defer->active) {
        auto it = g_defer->events.find(e);
        if (it != g_defer->events.end()) {
            DeferEvent &ev = *it->second;
            unsigned spins = 0;
            while (ev.wait_done.load(std::memory_order_acquire) < ev.wait_enq) cpu_relax(spins);
            const hipError_t rc = hipEventRecord(e, st);
            ev.rec_done.store(++ev.rec_enq, std::memory_order_release);
            return rc;
        }
    }
    return hipEventRecord(e, st);
}

hipError_t ds_stream_wait_event(hipStream_t st, hipEvent_t e, unsigned flags) {
    if (DeferWorker *w = recording(st)) {
        DeferEvent *ev = g_defer->event(e);
        DeferOp op;
        op.kind = 2;
        op.ev = ev;
        op.need = ev->rec_enq;               // the last record the host has issued (0: recorded before run())
        ++ev->wait_enq;
        op.fn = [=]() { return hipStreamWaitEvent(st, e, flags); };
        g_defer->push(w, std::move(op));
        return hipSuccess;
    }
    if (g_defer && g_defer->active) {
        auto it = g_defer->events.find(e);
        if (it != g_defer->events.end()) {   // the record may still sit in a worker's queue
            DeferEvent &ev = *it->second;
            unsigned spins = 0;
            while (ev.rec_done.load(std::memory_order_acquire) < ev.rec_enq) cpu_relax(spins);
        }
    }
    return hipStreamWaitEvent(st, e, flags);
}

hipError_t ds_stream_synchronize(hipStream_t st) {
    if (g_defer && g_defer->active) {
        const int err = g_defer->drain(st);
        if (err) return (hipError_t)err;
    }
    return hipStreamSynchronize(st);
}

}  // namespace dotsocp
