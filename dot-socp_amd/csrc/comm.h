// RCCL binding for the one-process-per-GPU time-slab mode.  librccl is opened at run time
// (dlopen by SONAME, so a copy already loaded by the host program -- e.g. PyTorch's -- is
// reused) and only when a communicator is attached; single-GPU use never touches it.
#pragma once
#include <rccl/rccl.h>

#include "common.h"

namespace dotsocp {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    int load();   // 0 on success, DOTSOCP_ECOMM otherwise (error text set)
};

Rccl &rccl_api();

// Solver members only: a failing Send / Recv inside open groups closes every open group (Solver::open_groups counts
// GroupStart minus GroupEnd) before the error is returned, so the communicator is not left with a dangling group
#define DS_NCCL_G(call)                                                                        \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess) {                                                              \
            dotsocp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,                   \
                               dotsocp::rccl_api().GetErrorString(r__));                       \
            for (; open_groups > 0; --open_groups) (void)dotsocp::rccl_api().GroupEnd();       \
            return DOTSOCP_ECOMM;                                                              \
        }                                                                                      \
    } while (0)

#define DS_NCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess) {                                                              \
            dotsocp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,                   \
                               dotsocp::rccl_api().GetErrorString(r__));                       \
            return DOTSOCP_ECOMM;                                                              \
        }                                                                                      \
    } while (0)

}  // namespace dotsocp
