#include "comm.h"

#include <dlfcn.h>

namespace dotsocp {

Rccl &rccl_api() {
    static Rccl api;
    return api;
}

int Rccl::load() {
    if (handle) return 0;
    // DOTSOCP_RCCL_LIB: explicit library path (the test suite points it at tests/fake_rccl to run several
    // ranks on one GPU); otherwise the system / already-loaded RCCL
    const char *names[] = {getenv("DOTSOCP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (handle) break;
    }
    if (!handle) {
        set_error("cannot load librccl.so.1: %s", dlerror());
        return DOTSOCP_ECOMM;
    }
#define DS_SYM(field, sym)                                                   \
    field = (decltype(field))dlsym(handle, sym);                             \
    if (!field) {                                                            \
        set_error("librccl: missing symbol %s", sym);                        \
        handle = nullptr;                                                    \
        return DOTSOCP_ECOMM;                                                \
    }
    DS_SYM(GetUniqueId, "ncclGetUniqueId");
    DS_SYM(CommInitRank, "ncclCommInitRank");
    DS_SYM(CommDestroy, "ncclCommDestroy");
    DS_SYM(Send, "ncclSend");
    DS_SYM(Recv, "ncclRecv");
    DS_SYM(AllReduce, "ncclAllReduce");
    DS_SYM(GroupStart, "ncclGroupStart");
    DS_SYM(GroupEnd, "ncclGroupEnd");
    DS_SYM(GetErrorString, "ncclGetErrorString");
#undef DS_SYM
    return 0;
}

}  // namespace dotsocp
