// comm.cpp — the one collective of the render path between PROCESSES (one process per GPU): RCCL over xGMI.
//
// Every rank holds the packed u32 stripes it rendered (mi355rt_config.stripe_*).  Gather-to-one as grouped
// ncclSend / ncclRecv: each peer's stripes travel on its own direct xGMI link to the root (4.1 MB per peer for a
// 3840x2160 frame over 8 GPUs), the root then places the slots into the frame with one kernel.  A ring all-gather
// would move 7x the bytes per link for nothing: only the root needs the frame.
// librccl.so is loaded with dlopen when a communicator is first asked for, so that single-GPU users of libmi355rt.so
// do not carry the dependency.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>
#include "renderer.hpp"

namespace mi355rt {

namespace {
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi* rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.error = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + n; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.CommUserRank = (decltype(api.CommUserRank))sym("ncclCommUserRank");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return &api;
}
}  // namespace

bool comm_unique_id(uint8_t* id128, std::string& err)
{
    RcclApi* a = rccl();
    if (!a->error.empty()) { err = a->error; return false; }
    static_assert(sizeof(ncclUniqueId) == 128, "mi355rt.h promises 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = a->GetUniqueId(&id);
    if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + a->GetErrorString(r); return false; }
    std::memcpy(id128, &id, 128);
    return true;
}

#define RCCL_TRY(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) { last_error = std::string(#expr) + ": " + a->GetErrorString(r__); return false; } } while (0)

// Everything mi355rt_comm_init can fail on BEFORE it enters the collective ncclCommInitRank, checked locally and without
// communication: the ranks agree on this answer first (stripes.py), so that no rank waits inside RCCL's bootstrap for a
// peer that never got there.
bool Renderer::comm_available()
{
    if (!bind()) return false;
    RcclApi* a = rccl();
    if (!a->error.empty()) { last_error = a->error; return false; }
    if (comm_) { last_error = "communicator already initialised"; return false; }
    return true;
}

bool Renderer::comm_init(const uint8_t* id128)
{
    if (!bind()) return false;
    RcclApi* a = rccl();
    if (!a->error.empty()) { last_error = a->error; return false; }
    if (comm_) { last_error = "communicator already initialised"; return false; }
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclComm_t c = nullptr;
    RCCL_TRY(a->CommInitRank(&c, (int)cfg.stripe_world, id, (int)cfg.stripe_rank));
    comm_ = c;
    return true;
}

// ranks of the communicator as RCCL itself counts them (0: none) — what bench.py reports as `rccl_ranks`
uint32_t Renderer::comm_ranks()
{
    if (!comm_) return 0u;
    RcclApi* a = rccl();
    int n = 0, me = -1;
    if (!a->CommCount || a->CommCount((ncclComm_t)comm_, &n) != ncclSuccess || n < 0) return 0u;
    if (a->CommUserRank && a->CommUserRank((ncclComm_t)comm_, &me) == ncclSuccess && me != (int)cfg.stripe_rank) return 0u;   // not the communicator this handle was dealt into
    return (uint32_t)n;
}

void Renderer::comm_destroy()
{
    if (!comm_) return;
    RcclApi* a = rccl();
    if (a->CommDestroy && hipSetDevice(cfg.device) == hipSuccess) {
        (void)hipStreamSynchronize(stream_);
        (void)a->CommDestroy((ncclComm_t)comm_);
    }
    comm_ = nullptr;
}

// Collective.  On return the transfers are queued on the handle's stream; the root has the frame in its device
// LDR buffer (and in host_out, after a synchronisation, when host_out != null).
bool Renderer::comm_gather(uint32_t root, uint32_t* host_out, size_t n)
{
    if (!bind()) return false;
    RcclApi* a = rccl();
    if (!comm_) { last_error = "no communicator: call mi355rt_comm_init first"; return false; }
    if (root >= cfg.stripe_world) { last_error = "root out of range"; return false; }
    const bool is_root = cfg.stripe_rank == root;
    if (!gather_prepare(is_root)) return false;
    if (!tonemap_to_gather_slot()) return false;
    // Inside GroupStart .. GroupEnd nothing may return early: a group left open makes every later RCCL call of this thread
    // part of it.  The first failing Send / Recv is remembered, the group is closed regardless, and a communicator whose
    // GroupEnd fails is destroyed (its state is undefined) so that the caller gets a clean "no communicator" next time.
    RCCL_TRY(a->GroupStart());
    ncclResult_t first = ncclSuccess; const char* what = "";
    if (is_root) {
        for (uint32_t r = 0; r < cfg.stripe_world && first == ncclSuccess; ++r) {
            if (r == root) continue;
            const size_t count = (size_t)rows_of_rank(r) * cfg.width;
            if (count) { first = a->Recv(gather_slot(r), count, ncclUint32, (int)r, (ncclComm_t)comm_, stream_); what = "ncclRecv"; }
        }
    } else {
        const size_t count = owned_rows.size() * (size_t)cfg.width;
        if (count) { first = a->Send(gather_slot(cfg.stripe_rank), count, ncclUint32, (int)root, (ncclComm_t)comm_, stream_); what = "ncclSend"; }
    }
    const ncclResult_t end = a->GroupEnd();
    if (first != ncclSuccess || end != ncclSuccess) {
        last_error = first != ncclSuccess ? std::string(what) + ": " + a->GetErrorString(first) : std::string("ncclGroupEnd: ") + a->GetErrorString(end);
        if (end != ncclSuccess) comm_destroy();
        return false;
    }
    if (is_root) return finish_gather(host_out, n);
    return true;
}

}  // namespace mi355rt
