#include "hostmem.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace dotsocp {

int host_threads() {
    static const int n = [] {
        const char *e = getenv("DOTSOCP_HOST_COPY_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            v = (int)std::min(16u, hw ? hw : 1u);
        }
        return std::max(1, std::min(v, 256));
    }();
    return n;
}

// f(begin, end) over [0, n) in contiguous pieces that are multiples of `grain`, one per thread
template <class F>
static void host_parallel(size_t n, size_t grain, F f) {
    const size_t units = (n + grain - 1) / grain;
    const size_t T = std::min<size_t>((size_t)host_threads(), units);
    if (T <= 1) {
        f((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    const size_t per = (units + T - 1) / T;
    for (size_t k = 1; k < T; ++k) {
        const size_t b = std::min(n, k * per * grain), e = std::min(n, (k + 1) * per * grain);
        if (b < e) th.emplace_back([=] { f(b, e); });
    }
    f((size_t)0, std::min(n, per * grain));
    for (auto &t : th) t.join();
}

void host_first_touch(void *p, size_t bytes) {
    if (!p || bytes < ((size_t)32 << 20)) return;
    const size_t page = 4096;
    // whole pages inside the buffer, one aligned 8-byte word each
    const uintptr_t lo = ((uintptr_t)p + 7) & ~(uintptr_t)7, hi = (uintptr_t)p + bytes;
    if (hi < lo + 8) return;
    const size_t span = hi - lo;
    host_parallel(span, page, [=](size_t b, size_t e) {
        // compare-and-swap of a word with itself: a write access whatever the outcome (an atomic `or 0` / `add 0` is
        // turned into a fenced LOAD by the compiler, which maps the shared zero page and allocates nothing)
        for (size_t o = b; o + 8 <= e; o += page) {
            uint64_t *w = (uint64_t *)(lo + o);
            uint64_t v = __atomic_load_n(w, __ATOMIC_RELAXED);
            __atomic_compare_exchange_n(w, &v, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
        }
    });
}

void host_scale(double *p, long long n, double s) {
    if (n <= 0) return;
    host_parallel((size_t)n, (size_t)1 << 16, [=](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) p[i] = s * p[i];
    });
}

}  // namespace dotsocp
