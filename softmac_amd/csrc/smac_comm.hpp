// RCCL for the slab decomposition inside the library (no reference counterpart: the reference is single-device).
//
// The library does not LINK against librccl (573 MB, and a single-GPU user never needs it): the entry points are resolved with
// dlopen / dlsym at smac_comm_init, typed from <rccl/rccl.h> so that a signature change is a compile error, not a crash.
// Neighbour-only ncclSend / ncclRecv inside one ncclGroup (each GPU pair of an MI355X node has its own xGMI link; a ring all-reduce
// of the grid would move G instead of 2 planes and be per-link bound) + a small ncclAllReduce for the primitives' wrench sums / adjoints.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
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

// ------------------------------------------------------------------------------------------------------------------------------------------
// SMAC_COMM_STUB=2: the slab loop between DIFFERENT ranks without RCCL - two or more processes that may share ONE GPU (two RCCL ranks cannot).
// Round 4: the in-library loop had only ever run as a world-1 self exchange; this link executes the distinct-peer code - the
// left / right slot mapping, the two-sided pack and unpack-add, the migration's count and row messages - with real ranks on the one GPU a
// development box has.  It is a TEST transport, host-synchronous and slow by design:
//   * every rank owns a device "mailbox" of two slots (towards its left / right neighbour), exported with hipIpcGetMemHandle; the handles and the
//     barriers live in a POSIX shared-memory segment named by the 128-byte id that smac_comm_unique_id makes on rank 0;
//   * one exchange = copy the outgoing messages into the own mailbox, stream sync, pair barrier with each neighbour involved, copy the incoming
//     messages out of the neighbours' mailboxes (left neighbour's RIGHT slot, right neighbour's LEFT slot), stream sync, pair barrier again (the
//     mailbox may be overwritten from then on);
//   * the small all-reduces go through a host array in the same segment.
// A rank that waits longer than 60 s at a barrier gives up with an error (its peer failed): nothing hangs.
// ------------------------------------------------------------------------------------------------------------------------------------------
struct IpcLink {
    static constexpr int MAX_WORLD = 8;
    static constexpr size_t REDUCE_CAP = 1 << 16;                 // doubles per rank and round of the host-side all-reduce
    struct Barrier { std::atomic<int> count, sense; };
    struct Shm {
        std::atomic<int> aborted;                                  // set by any rank that gives up (a failure, a timeout, smac_comm_abort): every barrier wait ends at once
        Barrier all;
        Barrier pair[MAX_WORLD];                                   // pair[r]: between rank r and rank r + 1
        hipIpcMemHandle_t handle[MAX_WORLD];
        size_t slot_bytes[MAX_WORLD];
        double reduce[MAX_WORLD][REDUCE_CAP];
    };
    Shm* shm = nullptr;
    std::string name, err;
    int rank = 0, world = 1;
    int sense_all = 0, sense_pair[MAX_WORLD] = {};
    char* mailbox = nullptr;                                       // 2 slots of slot_bytes: [to the left | to the right]
    size_t slot_bytes = 0;
    char* peer_box[2] = {nullptr, nullptr};                        // the left / right neighbour's mailbox (own pointer in the world-1 self loop)
    bool opened[2] = {false, false};

    static bool make_id(char id128[128], std::string& err) {       // rank 0: create the segment, the id is its name
        memset(id128, 0, 128);
        snprintf(id128, 128, "/smac-ipc-%d-%ld", (int)getpid(), (long)time(nullptr));
        int fd = shm_open(id128, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { err = std::string("shm_open(create) failed for ") + id128; return false; }
        if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); shm_unlink(id128); err = "ftruncate failed on the IPC segment"; return false; }
        void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { shm_unlink(id128); err = "mmap failed on the IPC segment"; return false; }
        memset(p, 0, sizeof(Shm));                                 // (zeroed barriers; the creator keeps no mapping: it maps again in attach)
        munmap(p, sizeof(Shm));
        return true;
    }
    bool attach(const char* id128, int r, int w) {
        if (w > MAX_WORLD) { err = "SMAC_COMM_STUB=2 links at most 8 ranks"; return false; }
        name.assign(id128, strnlen(id128, 127));
        int fd = -1;
        for (int tries = 0; tries < 200 && fd < 0; ++tries) {      // (the creator may be a moment behind)
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            if (fd < 0) usleep(50000);
        }
        if (fd < 0) { err = "cannot open the IPC segment " + name; return false; }
        void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { err = "mmap failed on the IPC segment"; return false; }
        shm = (Shm*)p;
        rank = r; world = w;
        return true;
    }
    // once every rank holds a mapping the NAME can go: the kernel then frees the segment with the last unmap - also after a crash, a timeout or os._exit
    // (only rank 0's detach() removed it, every failure path left 4 MB in /dev/shm).  Every rank calls it; ENOENT is the expected answer for all but one.
    void unlink_name() { if (!name.empty()) (void)shm_unlink(name.c_str()); }
    // give up on behalf of everybody: a rank waiting at any barrier of this link returns with an error within one poll (50 us)
    void abort_link() { if (shm) shm->aborted.store(1, std::memory_order_release); }
    bool aborted() const { return shm && shm->aborted.load(std::memory_order_acquire) != 0; }
    bool sync(Barrier& b, int parties, int& local_sense) {
        if (aborted()) { err = "IPC link: aborted (a rank of this link failed or called smac_comm_abort)"; return false; }
        local_sense ^= 1;
        if (b.count.fetch_add(1, std::memory_order_acq_rel) == parties - 1) {
            b.count.store(0, std::memory_order_relaxed);
            b.sense.store(local_sense, std::memory_order_release);
            return true;
        }
        for (long waited = 0; b.sense.load(std::memory_order_acquire) != local_sense; waited += 50) {
            if (aborted()) { err = "IPC link: aborted while waiting at a barrier (a rank of this link failed or called smac_comm_abort)"; return false; }
            if (waited > 60L * 1000 * 1000) {
                abort_link();                                      // (this rank's arrival stays counted in the barrier: the link is unusable from here on, and says so to everybody)
                err = "IPC link: a neighbour did not reach the barrier within 60 s (it failed or exchanges a different sequence)";
                return false;
            }
            usleep(50);
        }
        return true;
    }
    bool sync_all() { return world == 1 || sync(shm->all, world, sense_all); }
    bool sync_pair(int side) {                                     // side 0: with the left neighbour, 1: with the right one
        if (world == 1) return true;
        const int pr = side == 0 ? rank - 1 : rank;
        return sync(shm->pair[pr], 2, sense_pair[pr]);
    }
    // the mailbox of this rank (2 x bytes) and its neighbours' (self_loop: this rank's own)
    bool open_boxes(size_t bytes, int peer_l, int peer_r, bool self_loop) {
        close_boxes();
        const bool alone = self_loop || world == 1;
        if (!alone) {                                              // one slot size for everybody: the largest any rank asks for (capacities differ per slab)
            shm->slot_bytes[rank] = bytes;
            if (!sync_all()) return false;
            for (int r = 0; r < world; ++r) bytes = shm->slot_bytes[r] > bytes ? shm->slot_bytes[r] : bytes;
            if (!sync_all()) return false;                         // (everybody has read the sizes before anybody publishes again)
        }
        if (hipMalloc((void**)&mailbox, 2 * bytes) != hipSuccess) { err = "hipMalloc failed for the IPC mailbox"; return false; }
        slot_bytes = bytes;
        if (alone) { peer_box[0] = peer_box[1] = mailbox; return true; }
        if (hipIpcGetMemHandle(&shm->handle[rank], mailbox) != hipSuccess) { err = "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)"; return false; }
        if (!sync_all()) return false;
        const int peers[2] = {peer_l, peer_r};
        for (int s = 0; s < 2; ++s) {
            if (peers[s] < 0) continue;
            if (hipIpcOpenMemHandle((void**)&peer_box[s], shm->handle[peers[s]], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                err = "hipIpcOpenMemHandle failed"; return false;
            }
            opened[s] = true;
        }
        return sync_all();
    }
    void close_boxes() {
        for (int s = 0; s < 2; ++s) {
            if (opened[s]) (void)hipIpcCloseMemHandle(peer_box[s]);
            opened[s] = false; peer_box[s] = nullptr;
        }
        if (mailbox) (void)hipFree(mailbox);
        mailbox = nullptr; slot_bytes = 0;
    }
    void detach() {
        close_boxes();
        if (shm) { munmap(shm, sizeof(Shm)); shm = nullptr; unlink_name(); }
    }
};

template <class R> struct nccl_type;
template <> struct nccl_type<float> { static constexpr ncclDataType_t v = ncclFloat32; };
template <> struct nccl_type<double> { static constexpr ncclDataType_t v = ncclFloat64; };

}  // namespace smac
