// Guard bands around the solver's device buffers ("out-of-bounds canaries around device
// buffers").  Off by default; DOTSOCP_CANARY=1 switches them on for every allocation made afterwards.
//
// Layout of a guarded buffer:  [ GUARD bytes of pattern | payload (rounded up to 256 B) | GUARD bytes of pattern ];
// the pointer handed out is base + GUARD, so the 256-byte alignment hipMalloc gives (and the 16-byte vector
// accesses of the kernels rely on) is kept.  The pattern is a quiet NaN with a recognisable payload.
#include <map>
#include <mutex>
#include <vector>

#include "solver.h"
#include "defer.h"

namespace dotsocp {

namespace {

constexpr size_t GUARD = 4096;                              // bytes on each side (512 doubles: two rows of a 64 x 4 tile)
constexpr unsigned long long PATTERN = 0x7ff8dead5afe0badull;

struct Rec {
    char *base;
    size_t bytes;      // payload bytes as requested
    size_t padded;     // payload rounded up to 256
    int dev;
};

std::mutex g_mu;
std::map<void *, Rec> g_live;      // user pointer -> record

__global__ void k_fill_pattern(unsigned long long *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = PATTERN;
}

// one workgroup that idles for about `ticks` of the 100 MHz wall clock (bounded: every wave leaves after 2^20 polls at most)
__global__ void k_stall(long long ticks) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 20); ++i) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

}  // namespace

// Race detector for the multi-stream paths (DOTSOCP_STRESS_STREAMS=1): a short stall of pseudo-random length (0 .. 300 us)
// in front of whatever is enqueued next on `st`.  Every ordering between streams has to come from an event then, not from
// kernels happening to take as long as they usually do; results must not change (tests/test_gpu_multidevice.py).
bool stream_stress_enabled() {
    static const bool on = getenv("DOTSOCP_STRESS_STREAMS") && atoi(getenv("DOTSOCP_STRESS_STREAMS")) != 0;
    return on;
}

void stream_stress(hipStream_t st) {
    static std::mutex mu;                                                  // contexts may be driven from several host threads
    static unsigned long long shared_state = 0x9e3779b97f4a7c15ull;
    unsigned long long state;
    {
        std::lock_guard<std::mutex> lock(mu);
        shared_state ^= shared_state << 13; shared_state ^= shared_state >> 7; shared_state ^= shared_state << 17;   // xorshift
        state = shared_state;
    }
    if ((state & 3) == 0) return;                                          // a quarter of the calls add nothing
    const long long ticks = (long long)((state >> 8) % 30000);             // up to 300 us at 100 MHz
    DS_KLAUNCH(k_stall, dim3(1), dim3(64), 0, st, ticks);
}

bool canary_enabled() {
    const char *e = getenv("DOTSOCP_CANARY");
    return e && atoi(e) != 0;
}

int guarded_malloc(void **p, size_t bytes) {
    *p = nullptr;
    if (!canary_enabled()) {
        DS_HIP(hipMalloc(p, bytes));
        return 0;
    }
    const size_t padded = (bytes + 255) / 256 * 256;
    char *base = nullptr;
    DS_HIP(hipMalloc((void **)&base, padded + 2 * GUARD));
    // the slack between the payload's end and the rear band is pattern too: an overrun by one element is seen
    const size_t head = GUARD / 8, tail = (padded - bytes + GUARD) / 8;
    (void)hipGetLastError();       // a stale error of an earlier, failed call must not be blamed on the fill launches
    DS_KLAUNCH(k_fill_pattern, dim3(4), dim3(256), 0, nullptr, (unsigned long long *)base, head);
    // payload sizes are multiples of 8 (doubles, double2); the tail starts right behind the payload
    DS_KLAUNCH(k_fill_pattern, dim3(4), dim3(256), 0, nullptr, (unsigned long long *)(base + GUARD + bytes / 8 * 8),
                       tail);
    DS_HIP(hipGetLastError());
    DS_HIP(hipDeviceSynchronize());
    int dev = 0;
    (void)hipGetDevice(&dev);
    *p = base + GUARD;
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[*p] = Rec{base, bytes, padded, dev};
    return 0;
}

void guarded_free(void *p) {
    if (!p) return;
    Rec r{};
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) { r = it->second; found = true; g_live.erase(it); }
    }
    (void)hipFree(found ? (void *)r.base : p);
}

int canary_check(std::string *report) {
    if (report) report->clear();
    std::vector<std::pair<void *, Rec>> recs;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        recs.assign(g_live.begin(), g_live.end());
    }
    if (recs.empty()) return 0;
    int bad = 0, cur = -1;
    (void)hipGetDevice(&cur);
    std::vector<unsigned long long> h;
    for (auto &pr : recs) {
        const Rec &r = pr.second;
        if (hipSetDevice(r.dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { ++bad; continue; }
        const size_t head = GUARD / 8, tail = (r.padded - r.bytes + GUARD) / 8;
        h.resize(head + tail);
        if (hipMemcpy(h.data(), r.base, head * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h.data() + head, r.base + GUARD + r.bytes / 8 * 8, tail * 8, hipMemcpyDeviceToHost) != hipSuccess) {
            ++bad;
            continue;
        }
        long long first = -1;
        size_t count = 0;
        for (size_t i = 0; i < h.size(); ++i)
            if (h[i] != PATTERN) {
                if (first < 0) first = (long long)i;
                ++count;
            }
        if (count) {
            ++bad;
            if (report && bad <= 4) {
                char buf[256];
                const bool front = (size_t)first < head;
                snprintf(buf, sizeof buf, "%sbuffer %p (%zu bytes, device %d): %zu guard words overwritten, first %s the payload at word %lld",
                         report->empty() ? "" : "; ", pr.first, r.bytes, r.dev, count, front ? "in front of" : "behind",
                         front ? (long long)head - first : first - (long long)head);
                *report += buf;
            }
        }
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    return bad;
}

}  // namespace dotsocp
