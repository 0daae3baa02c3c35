// RCCL for the slab decomposition inside the library (no reference counterpart: the reference is single-device).
//
// The library does not LINK against librccl (573 MB, and a single-GPU user never needs it): the entry points are resolved with
// dlopen / dlsym at smac_comm_init, typed from <rccl/rccl.h> so that a signature change is a compile error, not a crash.
// Neighbour-only ncclSend / ncclRecv inside one ncclGroup (each GPU pair of an MI355X node has its own xGMI link; a ring all-reduce
// of the grid would move G instead of 2 planes and be per-link bound) + a small ncclAllReduce for the primitives' wrench sums / adjoints.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

namespace smac {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;

    bool load() {
        if (lib) return true;
        // ONE RCCL per process.  A host that already carries one (a PyTorch process: libtorch_hip.so depends on its bundled librccl.so, loaded whether
        // or not torch.distributed ever uses it) gets THAT instance: RTLD_NOLOAD by the names it is known under.  Otherwise the ROCm install's
        // library is loaded with RTLD_DEEPBIND, so that its internal calls bind to itself whatever other copy a later import brings into the
        // process (without it, round 3's first run printed torch's library path from inside /opt/rocm's RCCL: symbols of two copies had mixed).
        // SMAC_RCCL_LIB: explicit path, taken as given - if it does not load, nothing else is tried (tests/test_abi.py provokes the failure path with it).
        const char* forced = getenv("SMAC_RCCL_LIB");
        const bool only_forced = forced && *forced;
        if (only_forced) lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (!lib && !only_forced)
            for (const char* n : {"librccl.so", "librccl.so.1"}) {
                lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
                if (lib) break;
            }
        if (!lib && !only_forced)
            for (const char* n : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
                lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
                if (lib) break;
            }
        if (!lib) {
            const char* why = dlerror();          // (one call: dlerror() clears the message it returns)
            err = std::string("RCCL not found (dlopen librccl.so.1): ") + (why ? why : "no loader message");
            return false;
        }
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(lib, name); if (!p) { ok = false; err = std::string("RCCL symbol missing: ") + name; } return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(lib); lib = nullptr; }
        return ok;
    }
    static Rccl& get() { static Rccl r; return r; }
};

template <class R> struct nccl_type;
template <> struct nccl_type<float> { static constexpr ncclDataType_t v = ncclFloat32; };
template <> struct nccl_type<double> { static constexpr ncclDataType_t v = ncclFloat64; };

}  // namespace smac
