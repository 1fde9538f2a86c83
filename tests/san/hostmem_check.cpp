// Drives dot-socp_amd/csrc/hostmem.hip (plain C++: built with g++ under ThreadSanitizer / AddressSanitizer by
// tests/test_sanitizers.py): first touch leaves every byte as it was, at every alignment of the buffer; the threaded
// scaling equals the serial loop bit for bit.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hostmem.h"

int main() {
    int failed = 0;
    const size_t n = ((size_t)40 << 20) + 4099;          // above the 32 MB threshold, not a multiple of a page
    std::vector<unsigned char> buf(n + 16), ref(n + 16);
    for (size_t i = 0; i < buf.size(); ++i) buf[i] = ref[i] = (unsigned char)(i * 2654435761u >> 13);
    for (size_t off : {(size_t)0, (size_t)1, (size_t)5, (size_t)8, (size_t)13}) {
        dotsocp::host_first_touch(buf.data() + off, n);
        if (memcmp(buf.data(), ref.data(), buf.size()) != 0) { printf("first touch changed the buffer at offset %zu\n", off); ++failed; }
    }
    dotsocp::host_first_touch(nullptr, n);
    dotsocp::host_first_touch(buf.data(), 0);
    dotsocp::host_first_touch(buf.data(), 7);
    const long long m = (1 << 22) + 12345;
    std::vector<double> a((size_t)m), b((size_t)m);
    for (long long i = 0; i < m; ++i) a[(size_t)i] = b[(size_t)i] = 1.0 / (double)(i + 3) - 0.37 * (double)(i % 17);
    const double s = 0.73105857863000487;
    dotsocp::host_scale(a.data(), m, s);
    for (long long i = 0; i < m; ++i) b[(size_t)i] = s * b[(size_t)i];
    if (memcmp(a.data(), b.data(), sizeof(double) * (size_t)m) != 0) { printf("host_scale differs from the serial loop\n"); ++failed; }
    dotsocp::host_scale(a.data(), 0, s);
    dotsocp::host_scale(a.data(), 1, 2.0);
    if (a[0] != 2.0 * b[0]) { printf("host_scale on one element\n"); ++failed; }
    printf("threads %d, %d failed\n", dotsocp::host_threads(), failed);
    return failed ? 1 : 0;
}
