// TEST INFRASTRUCTURE: ThreadSanitizer run of the slab-thread layer (dot-socp_amd/csrc/defer.h, defer.hip) on the CPU.
// The HIP calls are loggers here.  Scenario: the call pattern of the solver's time-slab loop -- per-slab launches, then
// neighbour exchanges ordered by events (record on the sender's stream, wait on the receiver's, copy, record back, wait),
// with the events reused round-robin -- recorded from one thread for 8 slabs and executed by 8 workers.  Checked:
//   * every stream executes its operations in the order they were recorded;
//   * a stream wait is executed after the record that preceded it in host order and before the NEXT record of the same
//     event (the binding a single host thread would have produced);
//   * no data race (TSAN), no deadlock (the run ends).
#include "defer.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

using namespace dotsocp;

struct stub_stream { int id; };
struct stub_event { int id; };

struct Logged { int kind; int stream; int event; long host_seq; };      // kind 0 op, 1 record, 2 wait
static std::mutex g_mu;
static std::vector<Logged> g_log;                                         // in EXECUTION order
static thread_local long t_seq = -1;                                      // host sequence number of the closure being executed

static void logit(int kind, hipStream_t st, hipEvent_t ev) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_log.push_back({kind, st ? st->id : -1, ev ? ev->id : -1, t_seq});
}

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t st) { logit(0, st, nullptr); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *, int, const void *, int, size_t, hipStream_t st) { logit(0, st, nullptr); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *, size_t, const void *, size_t, size_t, size_t, hipMemcpyKind, hipStream_t st) { logit(0, st, nullptr); return hipSuccess; }
hipError_t hipMemsetAsync(void *, int, size_t, hipStream_t st) { logit(0, st, nullptr); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) { logit(1, st, e); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned) { logit(2, st, e); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

int main() {
    const int P = 8, NEV = 4, ITERS = 300;
    std::vector<stub_stream> st(2 * P);
    std::vector<stub_event> ev(P * NEV);
    for (int i = 0; i < 2 * P; ++i) st[i].id = i;
    for (int i = 0; i < P * NEV; ++i) ev[i].id = i;
    DeferCtx ctx;
    for (int s = 0; s < P; ++s) {
        const int w = ctx.add_worker(0);
        ctx.map_stream(&st[2 * s], w);
        ctx.map_stream(&st[2 * s + 1], w);
    }
    struct Host { int kind, stream, event; };
    std::vector<Host> host;                                               // in RECORDING order
    long seq = 0;
    std::vector<int> next_ev(P, 0);
    auto ev_of = [&](int s) { int i = next_ev[s]; next_ev[s] = (i + 1) % NEV; return &ev[s * NEV + i]; };
    unsigned rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    g_defer = &ctx;
    for (int round = 0; round < 3; ++round) {                            // three run() calls: begin / end reset the event state
        ctx.begin();
        for (int it = 0; it < ITERS; ++it) {
            for (int s = 0; s < P; ++s) {                                 // per-slab work on both streams
                const int n = 1 + (int)(rnd() % 4);
                for (int k = 0; k < n; ++k) {
                    hipStream_t s0 = &st[2 * s + (int)(rnd() & 1)];
                    const long q = seq++;
                    host.push_back({0, s0->id, -1});
                    defer_or_run(s0, [q, s0]() { t_seq = q; logit(0, s0, nullptr); });
                }
            }
            for (int s = 0; s + 1 < P; ++s) {                             // exchange s -> s + 1 (Solver::xcopy)
                hipStream_t a = &st[2 * s], b = &st[2 * (s + 1)];
                hipEvent_t e1 = ev_of(s), e2 = ev_of(s + 1);
                t_seq = seq++; host.push_back({1, a->id, e1->id}); (void)ds_event_record(e1, a);
                t_seq = seq++; host.push_back({2, b->id, e1->id}); (void)ds_stream_wait_event(b, e1, 0);
                t_seq = seq++; host.push_back({0, b->id, -1}); (void)ds_memcpy_async(nullptr, nullptr, 8, hipMemcpyDeviceToDevice, b);
                t_seq = seq++; host.push_back({1, b->id, e2->id}); (void)ds_event_record(e2, b);
                t_seq = seq++; host.push_back({2, a->id, e2->id}); (void)ds_stream_wait_event(a, e2, 0);
            }
            if (it % 7 == 0) (void)ds_stream_synchronize(&st[2 * (int)(rnd() % P)]);   // a KKT check reads sums back
        }
        if (ctx.end() != 0) { printf("closure error\n"); return 1; }
    }
    g_defer = nullptr;
    // ---- checks ----
    int bad = 0;
    if (g_log.size() != host.size()) { printf("executed %zu of %zu operations\n", g_log.size(), host.size()); return 1; }
    // (1) per stream: execution order == recording order (kinds and events match one by one)
    for (int s = 0; s < 2 * P; ++s) {
        std::vector<Host> h;
        std::vector<Logged> l;
        for (auto &x : host) if (x.stream == s) h.push_back(x);
        for (auto &x : g_log) if (x.stream == s) l.push_back(x);
        if (h.size() != l.size()) { ++bad; continue; }
        for (size_t i = 0; i < h.size(); ++i)
            if (h[i].kind != l[i].kind || h[i].event != l[i].event) { ++bad; break; }
    }
    // (2) per event: the executed sequence of records and waits is the recorded one (a wait binds to the record that
    //     preceded it in host order: in the execution log it sits between that record and the next one)
    for (int e = 0; e < P * NEV; ++e) {
        std::vector<int> h, l;
        for (auto &x : host) if (x.event == e) h.push_back(x.kind);
        for (auto &x : g_log) if (x.event == e) l.push_back(x.kind);
        // waits between two records may execute in any order among themselves; records and the wait COUNT between them must match
        auto shape = [](const std::vector<int> &v) { std::vector<int> s; int c = 0; for (int k : v) { if (k == 1) { s.push_back(c); c = 0; } else ++c; } s.push_back(c); return s; };
        if (shape(h) != shape(l)) ++bad;
    }
    printf("%zu operations, %d failed\n", host.size(), bad);
    return bad ? 1 : 0;
}
