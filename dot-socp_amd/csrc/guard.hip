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

// ---------------------------------------------------------------------------------------------------------------
// Device block cache.  hipFree of a context's buffers is cheap, but the NEXT hipMalloc pays for it: measured on the
// MI355X boxes, creating a 1025 x 1025 x 129 context (54 GB in ~40 buffers) takes 0.67 s in a fresh process and 1.65 s
// right after a context of that size was destroyed (one hipMalloc call of 1.6 s: the driver hands freed memory out again
// only once it is wiped), and a context created next to such a release stalls once for 30-90 ms in its first iterations
// (DESIGN.md section 6).  A caller that solves one problem size again and again -- the usual MATLAB session -- would pay
// that on every call.  Freed buffers are therefore kept per device and handed out again to requests of (nearly) their
// size; a reused buffer is zero-filled, as fresh device memory is; the cache holds at most half of the device's memory
// (DOTSOCP_DEVICE_CACHE_GB), oldest blocks leave first.  DOTSOCP_DEVICE_CACHE=0 switches the cache off,
// dotsocp_release_cache() returns everything to the driver, an allocation failure does so by itself and retries.
namespace {

struct Block {
    void *p;
    size_t bytes;
    int dev;
};
std::mutex c_mu;
std::map<void *, Block> c_live;      // handed out: pointer -> block (with the size it was allocated with)
std::vector<Block> c_free;

bool cache_enabled() {
    static const bool on = !(getenv("DOTSOCP_DEVICE_CACHE") && atoi(getenv("DOTSOCP_DEVICE_CACHE")) == 0);
    return on;
}

}  // namespace

long long device_cache_release() {
    std::vector<Block> blocks;
    {
        std::lock_guard<std::mutex> lk(c_mu);
        blocks.swap(c_free);
    }
    int cur = -1;
    (void)hipGetDevice(&cur);
    long long bytes = 0;
    for (auto &b : blocks) {
        (void)hipSetDevice(b.dev);
        (void)hipFree(b.p);
        bytes += (long long)b.bytes;
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    return bytes;
}

static int cached_malloc(void **p, size_t bytes) {
    int dev = 0;
    DS_HIP(hipGetDevice(&dev));
    Block got{nullptr, 0, dev};
    {
        // best fit among the free blocks of this device that are at most an eighth larger than the request
        std::lock_guard<std::mutex> lk(c_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < c_free.size(); ++i) {
            const Block &b = c_free[i];
            if (b.dev != dev || b.bytes < bytes || b.bytes > bytes + bytes / 8 + 65536) continue;
            if (best == (size_t)-1 || b.bytes < c_free[best].bytes) best = i;
        }
        if (best != (size_t)-1) {
            got = c_free[best];
            c_free.erase(c_free.begin() + (long)best);
        }
    }
    if (got.p) {
        // like fresh device memory, a reused block reads as zeros (guarded_free made sure nothing is still using it)
        hipError_t e = hipMemsetAsync(got.p, 0, got.bytes, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) {
            (void)hipFree(got.p);
            DS_HIP(e);
        }
    } else {
        hipError_t e = hipMalloc(&got.p, bytes);
        if (e != hipSuccess) {                       // out of memory with blocks of other sizes in the cache: give them back, retry
            (void)hipGetLastError();
            if (device_cache_release() > 0) e = hipMalloc(&got.p, bytes);
        }
        DS_HIP(e);
        got.bytes = bytes;
    }
    *p = got.p;
    std::lock_guard<std::mutex> lk(c_mu);
    c_live[got.p] = got;
    return 0;
}

static void cached_free(void *p) {
    Block b{nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lk(c_mu);
        auto it = c_live.find(p);
        if (it != c_live.end()) { b = it->second; c_live.erase(it); }
    }
    if (!b.p) {
        (void)hipFree(p);
        return;
    }
    // hipFree waits for the device; a block that goes back into the cache must be as idle as a freed one
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != b.dev) (void)hipSetDevice(b.dev);
    const hipError_t e = hipDeviceSynchronize();
    if (cur >= 0 && cur != b.dev) (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        (void)hipFree(b.p);
        return;
    }
    // at most half of the device's memory (DOTSOCP_DEVICE_CACHE_GB) stays in the cache: the oldest blocks go first
    static const size_t cap = [] {
        const char *g = getenv("DOTSOCP_DEVICE_CACHE_GB");
        if (g && atof(g) >= 0) return (size_t)(atof(g) * 1e9);
        size_t fr = 0, tot = 0;
        return hipMemGetInfo(&fr, &tot) == hipSuccess ? tot / 2 : (size_t)64e9;
    }();
    std::vector<Block> evict;
    {
        std::lock_guard<std::mutex> lk(c_mu);
        c_free.push_back(b);
        size_t held = 0;
        for (auto &x : c_free) held += x.bytes;
        while (held > cap && !c_free.empty()) {
            held -= c_free.front().bytes;
            evict.push_back(c_free.front());
            c_free.erase(c_free.begin());
        }
    }
    for (auto &x : evict) {
        if (x.dev != cur) (void)hipSetDevice(x.dev);
        (void)hipFree(x.p);
        if (cur >= 0 && x.dev != cur) (void)hipSetDevice(cur);
    }
}

int guarded_malloc(void **p, size_t bytes) {
    *p = nullptr;
    if (!canary_enabled()) {
        if (cache_enabled()) return cached_malloc(p, bytes);
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
    // which way a buffer goes back is decided by how it was allocated, not by the environment at the time of the free
    Rec r{};
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) { r = it->second; found = true; g_live.erase(it); }
    }
    if (found) (void)hipFree((void *)r.base);
    else if (cache_enabled()) cached_free(p);       // (a pointer the cache does not know is freed there)
    else (void)hipFree(p);
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
