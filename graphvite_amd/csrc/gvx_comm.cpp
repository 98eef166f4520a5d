// gvx_comm.cpp — the carriers of gvx_comm.h: RCCL (dlopen), event-ordered device copies, caller-supplied callbacks.
#include "gvx_comm.h"

#if !defined(GVX_HOST_BUILD)  // tests/hostdev compiles this file without a GPU stack: no RCCL there
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is opened at run time
#endif

#include <string.h>

#include <mutex>
#include <set>

#include "gvk_internal.h"

#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

namespace gvx {
namespace {

#if !defined(GVX_HOST_BUILD)
// ---- RCCL through dlopen -------------------------------------------------------------------------------------------

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl *rccl() {
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, []() {
        // a process that already holds RCCL (PyTorch ships its own copy) keeps using that one: same SONAME
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *name : names)
            if ((lib.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib.handle) {
            const char *why = dlerror();
            lib.error = std::string("librccl is not loadable: ") + (why ? why : "unknown dlopen error");
            return;
        }
        bool ok = true;
#define GVX_SYM(field, symbol)                                                          \
    lib.field = reinterpret_cast<decltype(lib.field)>(dlsym(lib.handle, #symbol)); \
    ok = ok && lib.field != nullptr;
        GVX_SYM(GetUniqueId, ncclGetUniqueId)
        GVX_SYM(CommInitRank, ncclCommInitRank)
        GVX_SYM(CommInitAll, ncclCommInitAll)
        GVX_SYM(CommDestroy, ncclCommDestroy)
        GVX_SYM(AllGather, ncclAllGather)
        GVX_SYM(Send, ncclSend)
        GVX_SYM(Recv, ncclRecv)
        GVX_SYM(GroupStart, ncclGroupStart)
        GVX_SYM(GroupEnd, ncclGroupEnd)
        GVX_SYM(GetErrorString, ncclGetErrorString)
#undef GVX_SYM
        if (!ok) {
            lib.error = "librccl lacks an entry point this engine needs";
            lib.handle = nullptr;
        }
    });
    return &lib;  // usable when ->handle is set; ->error says why not
}

#define RCCL_TRY(lib, call)                                                                          \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) return gvk_fail(GVK_EHIP, "%s: %s", #call, (lib)->GetErrorString(r_)); \
    } while (0)

class RcclComm : public Comm {
public:
    Rccl *lib;
    int world;
    std::vector<ncclComm_t> comms;  // one per local worker, in the order of the engine's local workers
    RcclComm(Rccl *l, int w) : lib(l), world(w) {}
    ~RcclComm() override {
        for (ncclComm_t c : comms)
            if (c) lib->CommDestroy(c);
    }
    const char *name() const override { return "RCCL"; }

    int all_gather(const std::vector<Peer> &local, const std::vector<void *> &slabs, size_t bytes) override {
        // one collective: W in-place ncclAllGather calls, grouped when this process drives several GPUs
        RCCL_TRY(lib, lib->GroupStart());
        for (size_t i = 0; i < local.size(); i++) {
            hipSetDevice(local[i].device);
            const char *part = static_cast<const char *>(slabs[i]) + (size_t)local[i].rank * bytes;
            ncclResult_t r = lib->AllGather(part, slabs[i], bytes, ncclChar, comms[i], local[i].stream);
            if (r != ncclSuccess) {
                lib->GroupEnd();
                return gvk_fail(GVK_EHIP, "ncclAllGather: %s", lib->GetErrorString(r));
            }
        }
        RCCL_TRY(lib, lib->GroupEnd());
        return GVK_OK;
    }

    int all_to_all(const std::vector<Peer> &local, const std::vector<const void *> &send, const std::vector<void *> &recv,
                   size_t bytes) override {
        RCCL_TRY(lib, lib->GroupStart());
        for (size_t i = 0; i < local.size(); i++) {
            hipSetDevice(local[i].device);
            for (int q = 0; q < world; q++) {
                ncclResult_t r = lib->Send(static_cast<const char *>(send[i]) + (size_t)q * bytes, bytes, ncclChar, q, comms[i],
                                           local[i].stream);
                if (r == ncclSuccess)
                    r = lib->Recv(static_cast<char *>(recv[i]) + (size_t)q * bytes, bytes, ncclChar, q, comms[i], local[i].stream);
                if (r != ncclSuccess) {
                    lib->GroupEnd();
                    return gvk_fail(GVK_EHIP, "ncclSend / ncclRecv: %s", lib->GetErrorString(r));
                }
            }
        }
        RCCL_TRY(lib, lib->GroupEnd());
        return GVK_OK;
    }
};

#endif  // !GVX_HOST_BUILD

// ---- one process, workers that may share GPUs: device copies ordered by events ---------------------------------------

class CopyComm : public Comm {
public:
    int world;
    std::vector<hipEvent_t> entered, left;  // per local worker, created lazily on its device
    explicit CopyComm(int w) : world(w) {}
    ~CopyComm() override {
        for (hipEvent_t e : entered)
            if (e) hipEventDestroy(e);
        for (hipEvent_t e : left)
            if (e) hipEventDestroy(e);
    }
    const char *name() const override { return "device copies"; }

    int prepare(const std::vector<Peer> &local) {
        if (entered.size() == local.size()) return GVK_OK;
        entered.assign(local.size(), nullptr), left.assign(local.size(), nullptr);
        for (size_t i = 0; i < local.size(); i++) {
            hipSetDevice(local[i].device);
            if (hipEventCreateWithFlags(&entered[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&left[i], hipEventDisableTiming) != hipSuccess)
                return gvk_fail(GVK_EHIP, "exchange: cannot create events");
        }
        return GVK_OK;
    }

    // Nobody writes into a peer's buffer before EVERY worker has reached the exchange (a worker may still be reading the
    // part of its buffer that a peer is about to overwrite), and nobody leaves it before every copy has landed.
    template <class Copies>
    int run(const std::vector<Peer> &local, Copies copies) {
        int rc = prepare(local);
        if (rc != GVK_OK) return rc;
        for (size_t i = 0; i < local.size(); i++) {
            hipSetDevice(local[i].device);
            hipEventRecord(entered[i], local[i].stream);
        }
        for (size_t i = 0; i < local.size(); i++) {
            hipSetDevice(local[i].device);
            for (size_t j = 0; j < local.size(); j++)
                if (j != i) hipStreamWaitEvent(local[i].stream, entered[j], 0);
            rc = copies(i);
            if (rc != GVK_OK) return rc;
            hipEventRecord(left[i], local[i].stream);
        }
        for (size_t i = 0; i < local.size(); i++) {
            hipSetDevice(local[i].device);
            for (size_t j = 0; j < local.size(); j++)
                if (j != i) hipStreamWaitEvent(local[i].stream, left[j], 0);
        }
        return GVK_OK;
    }

    static int copy(void *to, int to_device, const void *from, int from_device, size_t bytes, hipStream_t stream) {
        hipError_t e = to_device == from_device ? hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, stream)
                                                : hipMemcpyPeerAsync(to, to_device, from, from_device, bytes, stream);
        return e == hipSuccess ? GVK_OK : gvk_fail(GVK_EHIP, "exchange copy: %s", hipGetErrorString(e));
    }

    int all_gather(const std::vector<Peer> &local, const std::vector<void *> &slabs, size_t bytes) override {
        return run(local, [&](size_t i) {
            const size_t offset = (size_t)local[i].rank * bytes;
            for (size_t j = 0; j < local.size(); j++) {
                if (j == i) continue;
                int rc = copy(static_cast<char *>(slabs[j]) + offset, local[j].device, static_cast<char *>(slabs[i]) + offset,
                              local[i].device, bytes, local[i].stream);
                if (rc != GVK_OK) return rc;
            }
            return (int)GVK_OK;
        });
    }

    int all_to_all(const std::vector<Peer> &local, const std::vector<const void *> &send, const std::vector<void *> &recv,
                   size_t bytes) override {
        return run(local, [&](size_t i) {
            for (size_t j = 0; j < local.size(); j++) {
                int rc = copy(static_cast<char *>(recv[j]) + (size_t)local[i].rank * bytes, local[j].device,
                              static_cast<const char *>(send[i]) + (size_t)local[j].rank * bytes, local[i].device, bytes,
                              local[i].stream);
                if (rc != GVK_OK) return rc;
            }
            return (int)GVK_OK;
        });
    }
};

// ---- a transport the embedding program supplies ------------------------------------------------------------------------

class CallbackComm : public Comm {
public:
    gvx_transport t;
    explicit CallbackComm(const gvx_transport &transport) : t(transport) {}
    const char *name() const override { return "caller-supplied transport"; }
    int all_gather(const std::vector<Peer> &local, const std::vector<void *> &slabs, size_t bytes) override {
        for (size_t i = 0; i < local.size(); i++) {
            int rc = t.all_gather(t.user, slabs[i], bytes, local[i].stream);
            if (rc != GVK_OK) return gvk_fail(rc, "transport all_gather failed (%d)", rc);
        }
        return GVK_OK;
    }
    int all_to_all(const std::vector<Peer> &local, const std::vector<const void *> &send, const std::vector<void *> &recv,
                   size_t bytes) override {
        for (size_t i = 0; i < local.size(); i++) {
            int rc = t.all_to_all(t.user, send[i], recv[i], bytes, local[i].stream);
            if (rc != GVK_OK) return gvk_fail(rc, "transport all_to_all failed (%d)", rc);
        }
        return GVK_OK;
    }
};

}  // namespace

#if !defined(GVX_HOST_BUILD)
Comm *make_rccl_in_process(const std::vector<int> &devices, std::string *why) {
    if (std::set<int>(devices.begin(), devices.end()).size() != devices.size()) {
        *why = "a GPU is listed more than once (RCCL wants one rank per device)";
        return nullptr;
    }
    Rccl *lib = rccl();
    if (!lib->handle) {
        *why = lib->error;
        return nullptr;
    }
    RcclComm *comm = new RcclComm(lib, (int)devices.size());
    comm->comms.assign(devices.size(), nullptr);
    ncclResult_t r = lib->CommInitAll(comm->comms.data(), (int)devices.size(), devices.data());
    if (r != ncclSuccess) {
        *why = std::string("ncclCommInitAll: ") + lib->GetErrorString(r);
        comm->comms.clear();
        delete comm;
        return nullptr;
    }
    return comm;
}

Comm *make_rccl_rank(int rank, int world, int device, const void *unique_id, size_t unique_id_bytes, std::string *why) {
    Rccl *lib = rccl();
    if (!lib->handle) {
        *why = lib->error;
        return nullptr;
    }
    if (!unique_id || unique_id_bytes < sizeof(ncclUniqueId)) {
        *why = "the RCCL unique id (gvx_unique_id on rank 0, broadcast to all ranks) is missing";
        return nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    RcclComm *comm = new RcclComm(lib, world);
    comm->comms.assign(1, nullptr);
    hipSetDevice(device);
    ncclResult_t r = lib->CommInitRank(&comm->comms[0], world, id, rank);
    if (r != ncclSuccess) {
        *why = std::string("ncclCommInitRank: ") + lib->GetErrorString(r);
        comm->comms.clear();
        delete comm;
        return nullptr;
    }
    return comm;
}

#else
Comm *make_rccl_in_process(const std::vector<int> &, std::string *why) {
    *why = "host build: no RCCL";
    return nullptr;
}
Comm *make_rccl_rank(int, int, int, const void *, size_t, std::string *why) {
    *why = "host build: no RCCL";
    return nullptr;
}
#endif

Comm *make_copies(int world) { return new CopyComm(world); }

Comm *make_callbacks(const gvx_transport &transport, int rank, int world) {
    (void)rank, (void)world;
    return new CallbackComm(transport);
}

#if !defined(GVX_HOST_BUILD)
int rccl_unique_id(void *out, size_t capacity, std::string *why) {
    Rccl *lib = rccl();
    if (!lib->handle) {
        *why = lib->error;
        return GVK_EHIP;
    }
    if (capacity < sizeof(ncclUniqueId)) {
        *why = "buffer too small for an RCCL unique id (128 bytes)";
        return GVK_EINVAL;
    }
    ncclUniqueId id;
    ncclResult_t r = lib->GetUniqueId(&id);
    if (r != ncclSuccess) {
        *why = std::string("ncclGetUniqueId: ") + lib->GetErrorString(r);
        return GVK_EHIP;
    }
    memcpy(out, &id, sizeof(id));
    return GVK_OK;
}

#else
int rccl_unique_id(void *, size_t, std::string *why) {
    *why = "host build: no RCCL";
    return GVK_EHIP;
}
#endif

}  // namespace gvx
