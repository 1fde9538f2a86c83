// Host-side helpers for the boundary that hands over HOST buffers (dotsocp_download, dotsocp_recover_outputs): the arrays
// are as large as the device fields (1 GB per node field at 1025 x 1025 x 129), so what the host does to them is done on a
// few threads.
#pragma once
#include <cstddef>

namespace dotsocp {

// Number of host threads for the helpers below: DOTSOCP_HOST_COPY_THREADS, else min(16, hardware threads).
int host_threads();

// Fault in the pages of [p, p + bytes) for writing WITHOUT changing their contents (one compare-and-swap of a word with itself per page).  A
// device-to-host copy into pages that exist runs at the link rate (55 GB/s measured on the MI355X boxes, pinned or not);
// into pages of a fresh allocation (numpy.empty, mxCreateDoubleMatrix: untouched zero pages) it is bound by the kernel
// zeroing them under the one copying thread (9 GB/s).  Arrays below 32 MB are left alone.
void host_first_touch(void *p, size_t bytes);

// p[i] = s * p[i], i < n  (var.alpha = sigma * alpha, var.beta = sigma * beta, solver_socp_inPALM.m:335-336)
void host_scale(double *p, long long n, double s);

}  // namespace dotsocp
