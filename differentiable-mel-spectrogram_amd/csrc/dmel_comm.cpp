// dmel_comm.cpp -- the one exchange step of the path: all-reduce of the scalar gradient d lambd over RCCL.
//
// The reference has no distributed code; with the batch sharded over GPUs the only cross-GPU
// datum is lambd.grad (4 bytes).  torch.distributed can do it, but at a ~35 us step its Python/Work
// bookkeeping costs more host time than the step has (measured: async_op=True 46 us/iter host, stream/event
// juggling from Python 64 us); this file issues the same ncclAllReduce natively: on the communicator's own
// stream, ordered against the caller's stream with two events, ~5 us of host time, overlapping the next step.
// RCCL is resolved with dlopen at first use (the library must stay loadable on machines without it), and the
// copy torch already loaded is preferred so that the process holds ONE RCCL.
#include "../../include/dmel.h"
#include "dmel_kernels.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

extern "C" const char* dmel_last_error(void);
namespace dmel { dmel_status set_error(dmel_status st, const std::string& msg); }

namespace {

struct NcclId { char internal[128]; };
typedef void* NcclComm;
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(NcclComm*, int, NcclId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_comm_destroy)(NcclComm);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
    std::string why;
};

Rccl& rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.handle) break; }   // torch's copy first
    if (!r.handle) for (const char* n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.handle) break; }
    if (!r.handle) { r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return r; }
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
        r.why = "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclAllReduce/ncclCommDestroy";
        r.handle = nullptr;
    }
    return r;
}

std::string nccl_msg(int rc)
{
    Rccl& r = rccl();
    return std::string("RCCL error ") + std::to_string(rc) + (r.error_string ? std::string(": ") + r.error_string(rc) : "");
}

constexpr int kRing = 64;

}  // namespace

struct dmel_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t side = nullptr;
    hipEvent_t ready[kRing];
    hipEvent_t done[kRing];
    int next = 0;
};

#define COMM_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return dmel::set_error(DMEL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

dmel_status dmel_comm_unique_id(uint8_t id[DMEL_COMM_ID_BYTES])
{
    if (!id) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "id is NULL");
    Rccl& r = rccl();
    if (!r.handle) return dmel::set_error(DMEL_ERR_UNSUPPORTED, r.why);
    NcclId nid;
    const int rc = r.get_unique_id(&nid);
    if (rc != 0) return dmel::set_error(DMEL_ERR_HIP, nccl_msg(rc));
    static_assert(sizeof(NcclId) == DMEL_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id, &nid, sizeof(nid));
    return DMEL_OK;
}

dmel_status dmel_comm_create(const uint8_t id[DMEL_COMM_ID_BYTES], int32_t rank, int32_t world, dmel_comm** comm)
{
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_comm_create: bad arguments");
    *comm = nullptr;
    Rccl& r = rccl();
    if (!r.handle) return dmel::set_error(DMEL_ERR_UNSUPPORTED, r.why);
    dmel_comm* c = new (std::nothrow) dmel_comm();
    if (!c) return dmel::set_error(DMEL_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->rank = rank; c->world = world;
    NcclId nid;
    std::memcpy(&nid, id, sizeof(nid));
    const int rc = r.comm_init_rank(&c->comm, world, nid, rank);
    if (rc != 0) { delete c; return dmel::set_error(DMEL_ERR_HIP, nccl_msg(rc)); }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    for (int i = 0; i < kRing && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) { r.comm_destroy(c->comm); delete c; return dmel::set_error(DMEL_ERR_HIP, std::string("stream/event creation: ") + hipGetErrorString(e)); }
    *comm = c;
    return DMEL_OK;
}

dmel_status dmel_comm_destroy(dmel_comm* c)
{
    if (!c) return DMEL_OK;
    (void)hipStreamSynchronize(c->side);
    for (int i = 0; i < kRing; ++i) { (void)hipEventDestroy(c->ready[i]); (void)hipEventDestroy(c->done[i]); }
    (void)hipStreamDestroy(c->side);
    if (c->comm) rccl().comm_destroy(c->comm);
    delete c;
    return DMEL_OK;
}

dmel_status dmel_comm_allreduce_async(dmel_comm* c, float* buf, int32_t count, void* stream, int32_t* ticket)
{
    if (!c || !buf || count < 1 || !ticket) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_comm_allreduce_async: bad arguments");
    const int i = c->next;
    c->next = (c->next + 1) % kRing;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    COMM_HIP(hipEventRecord(c->ready[i], s));                 // everything queued on the caller's stream so far ...
    COMM_HIP(hipStreamWaitEvent(c->side, c->ready[i], 0));    // ... happens before the collective
    const int rc = rccl().all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, c->side);
    if (rc != 0) return dmel::set_error(DMEL_ERR_HIP, nccl_msg(rc));
    COMM_HIP(hipEventRecord(c->done[i], c->side));
    *ticket = i;
    return DMEL_OK;
}

dmel_status dmel_comm_allreduce(dmel_comm* c, float* buf, int32_t count, void* stream)
{
    if (!c || !buf || count < 1) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_comm_allreduce: bad arguments");
    const int rc = rccl().all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) return dmel::set_error(DMEL_ERR_HIP, nccl_msg(rc));
    return DMEL_OK;
}

dmel_status dmel_comm_wait(dmel_comm* c, int32_t ticket, void* stream)
{
    if (!c || ticket < 0 || ticket >= kRing) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_comm_wait: bad ticket");
    COMM_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->done[ticket], 0));
    return DMEL_OK;
}

}  // extern "C"


// ---- peer-to-peer mailbox (include/dmel.h; the kernel side is mailbox_exchange in dmel_aux.hip) ---------------------------------
struct dmel_mailbox {
    int rank = 0, world = 1, device = 0;
    unsigned long long* inbox = nullptr;        // [2][world] granules on this device
    unsigned* step = nullptr;                   // device word
    unsigned long long* host_error = nullptr;   // pinned
    void* peer[DMEL_MAILBOX_MAX_WORLD] = {};    // mapped inboxes (peer[rank] == inbox)
    bool opened[DMEL_MAILBOX_MAX_WORLD] = {};
    bool connected = false;
    unsigned spin_limit = 0;                                  // 0: the wall-clock bound alone
    unsigned long long timeout_ticks = 120ull * 100000000ull; // 120 s of the 100 MHz s_memrealtime clock
    int attached = 0;                                         // plans that hold this mailbox (guarded by g_mb_mu)
    uint8_t handle[DMEL_MAILBOX_HANDLE_BYTES] = {};
};

namespace {
// mailboxes created by this process: a handle exported here cannot be opened here (hipIpcOpenMemHandle refuses), so ranks of one
// process find each other's inbox through this list
std::mutex g_mb_mu;
std::vector<dmel_mailbox*> g_mailboxes;
}

namespace dmel {
// what dmel_api.cpp needs to launch the dot kernel with the exchange in its tail
bool mailbox_args(const dmel_mailbox* mb, MailboxArgs* out)
{
    if (!mb || !mb->connected) return false;
    MailboxArgs a{};
    for (int r = 0; r < mb->world; ++r) a.peer_inbox[r] = static_cast<unsigned long long*>(mb->peer[r]);
    a.my_inbox = mb->inbox; a.step = mb->step; a.host_error = mb->host_error;
    a.rank = mb->rank; a.world = mb->world; a.spin_limit = mb->spin_limit; a.timeout_ticks = mb->timeout_ticks;
    *out = a;
    return true;
}
int mailbox_device(const dmel_mailbox* mb) { return mb ? mb->device : -1; }
void mailbox_attach_count(dmel_mailbox* mb, int delta)
{
    if (!mb) return;
    std::lock_guard<std::mutex> lock(g_mb_mu);
    mb->attached += delta;
}
// non-zero: (step << 32) | missing rank of an exchange that timed out (the word stays: dmel_mailbox_error clears it)
unsigned long long mailbox_peek_error(const dmel_mailbox* mb)
{
    return (mb && mb->host_error) ? __atomic_load_n(mb->host_error, __ATOMIC_RELAXED) : 0ull;
}
}  // namespace dmel

extern "C" {

dmel_status dmel_mailbox_create(int32_t rank, int32_t world, dmel_mailbox** out, uint8_t handle[DMEL_MAILBOX_HANDLE_BYTES])
{
    if (!out || !handle || world < 1 || world > DMEL_MAILBOX_MAX_WORLD || rank < 0 || rank >= world)
        return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_create: bad arguments (world <= 16)");
    *out = nullptr;
    static_assert(sizeof(hipIpcMemHandle_t) == DMEL_MAILBOX_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    static_assert(DMEL_MAILBOX_MAX_WORLD == dmel::kMailboxMaxWorld, "header and kernel agree on the largest world");
    dmel_mailbox* mb = new (std::nothrow) dmel_mailbox();
    if (!mb) return dmel::set_error(DMEL_ERR_OUT_OF_MEMORY, "host allocation failed");
    mb->rank = rank; mb->world = world;
    hipError_t e = hipGetDevice(&mb->device);
    const size_t bytes = 4096;                                           // 2 x 16 granules; a page of its own
    // uncached (fine-grained) device memory: remote stores and local polls both go to memory, not to a cache of either device
    if (e == hipSuccess) {
        e = hipExtMallocWithFlags(reinterpret_cast<void**>(&mb->inbox), bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(reinterpret_cast<void**>(&mb->inbox), bytes); }
    }
    if (e == hipSuccess) e = hipMemset(mb->inbox, 0, bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&mb->step), 64);
    if (e == hipSuccess) e = hipMemset(mb->step, 0, 64);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&mb->host_error), 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) { std::memset(mb->host_error, 0, 64); e = hipDeviceSynchronize(); }
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    if (e == hipSuccess && world > 1) {
        e = hipIpcGetMemHandle(&h, mb->inbox);
        if (e != hipSuccess) {
            // not every allocation kind can be exported: fall back to ordinary device memory (the exchange uses system-scope
            // stores and loads either way)
            (void)hipGetLastError();
            (void)hipFree(mb->inbox); mb->inbox = nullptr;
            e = hipMalloc(reinterpret_cast<void**>(&mb->inbox), bytes);
            if (e == hipSuccess) e = hipMemset(mb->inbox, 0, bytes);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipIpcGetMemHandle(&h, mb->inbox);
        }
    }
    if (e != hipSuccess) {
        const std::string msg = std::string("dmel_mailbox_create: ") + hipGetErrorString(e);
        dmel_mailbox_destroy(mb);
        return dmel::set_error(DMEL_ERR_HIP, msg);
    }
    std::memcpy(handle, &h, sizeof(h));
    std::memcpy(mb->handle, &h, sizeof(h));
    {
        std::lock_guard<std::mutex> lock(g_mb_mu);
        g_mailboxes.push_back(mb);
    }
    *out = mb;
    return DMEL_OK;
}

dmel_status dmel_mailbox_connect(dmel_mailbox* mb, const uint8_t* handles)
{
    if (!mb || (!handles && mb->world > 1)) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_connect: NULL argument");
    if (mb->connected) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_connect: already connected");
    for (int r = 0; r < mb->world; ++r) {
        if (r == mb->rank) { mb->peer[r] = mb->inbox; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * DMEL_MAILBOX_HANDLE_BYTES, sizeof(h));
        {
            // a rank of this very process: its pointer is used directly (peer access between the two devices enabled first)
            dmel_mailbox* local = nullptr;
            {
                std::lock_guard<std::mutex> lock(g_mb_mu);
                for (dmel_mailbox* o : g_mailboxes)
                    if (o != mb && o->world == mb->world && o->rank == r && std::memcmp(o->handle, &h, sizeof(h)) == 0) { local = o; break; }
            }
            if (local) {
                if (local->device != mb->device) {
                    int cur = -1;
                    (void)hipGetDevice(&cur);
                    hipError_t pe = hipSetDevice(mb->device);
                    if (pe == hipSuccess) pe = hipDeviceEnablePeerAccess(local->device, 0);
                    if (pe == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); pe = hipSuccess; }
                    if (cur >= 0) (void)hipSetDevice(cur);
                    if (pe != hipSuccess) {
                        (void)hipGetLastError();
                        return dmel::set_error(DMEL_ERR_HIP, "dmel_mailbox_connect: peer access to the device of rank " + std::to_string(r) + ": " + hipGetErrorString(pe));
                    }
                }
                mb->peer[r] = local->inbox;
                continue;
            }
        }
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return dmel::set_error(DMEL_ERR_HIP, "dmel_mailbox_connect: hipIpcOpenMemHandle(rank " + std::to_string(r) + "): " + hipGetErrorString(e));
        }
        mb->peer[r] = p; mb->opened[r] = true;
    }
    mb->connected = true;
    return DMEL_OK;
}

dmel_status dmel_mailbox_destroy(dmel_mailbox* mb)
{
    if (!mb) return DMEL_OK;
    {
        std::lock_guard<std::mutex> lock(g_mb_mu);
        if (mb->attached > 0)
            return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_destroy: still attached to " + std::to_string(mb->attached) +
                                   " plan(s) (dmel_plan_attach_mailbox(plan, NULL) or release them first): their next backward would use freed memory");
        for (size_t i = 0; i < g_mailboxes.size(); ++i)
            if (g_mailboxes[i] == mb) { g_mailboxes.erase(g_mailboxes.begin() + (long)i); break; }
    }
    // queued kernels may still poll the inbox: the mailbox's OWN device is drained, whichever device is current
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != mb->device && hipSetDevice(mb->device) == hipSuccess;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < mb->world; ++r) if (mb->opened[r]) (void)hipIpcCloseMemHandle(mb->peer[r]);
    (void)hipFree(mb->inbox); (void)hipFree(mb->step);
    if (mb->host_error) (void)hipHostFree(mb->host_error);
    if (switched) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    delete mb;
    return DMEL_OK;
}

dmel_status dmel_mailbox_allreduce(dmel_mailbox* mb, float* buf, void* stream)
{
    dmel::MailboxArgs a;
    if (!buf || !dmel::mailbox_args(mb, &a)) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_allreduce: NULL buffer or a mailbox that is not connected");
    COMM_HIP(dmel::launch_mailbox_allreduce(buf, a, reinterpret_cast<hipStream_t>(stream)));
    return DMEL_OK;
}

dmel_status dmel_mailbox_error(dmel_mailbox* mb, int32_t* failed, uint32_t* step, int32_t* missing_rank)
{
    if (!mb || !failed) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_error: NULL argument");
    const unsigned long long w = __atomic_exchange_n(mb->host_error, 0ull, __ATOMIC_RELAXED);
    *failed = w != 0 ? 1 : 0;
    if (step) *step = (uint32_t)(w >> 32);
    if (missing_rank) *missing_rank = (int32_t)(uint32_t)w;
    return DMEL_OK;
}

dmel_status dmel_mailbox_set_spin_limit(dmel_mailbox* mb, uint32_t polls)
{
    if (!mb) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_set_spin_limit: mailbox is NULL");
    mb->spin_limit = polls;
    return DMEL_OK;
}

dmel_status dmel_mailbox_set_timeout_ms(dmel_mailbox* mb, uint64_t milliseconds)
{
    if (!mb) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_set_timeout_ms: mailbox is NULL");
    if (milliseconds > (1ull << 40)) return dmel::set_error(DMEL_ERR_INVALID_ARGUMENT, "dmel_mailbox_set_timeout_ms: out of range");
    mb->timeout_ticks = milliseconds * 100000ull;             // 100 MHz
    return DMEL_OK;
}

}  // extern "C"
