// libmorl_hip.so, third translation unit: the collectives of the weight-sharded Envelope step behind the C ABI
// (include/morl_hip.h, "multi-GPU").  One process per GPU; the communicator is RCCL over xGMI.  RCCL is bound at RUN TIME
// (dlopen, preferring an instance the process has already loaded -- PyTorch ships its own librccl.so.1 and two instances in
// one process would each own half of the node's peer mappings): libmorl_hip.so itself has no link-time dependency on it,
// single-GPU users never load it.  Nothing here synchronises the host; every call enqueues on the caller's stream.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>

#include "morl_hip.h"
#include "morl_host.h"

using morl_host::fail;

namespace {

// the handful of RCCL entry points used, with their rccl.h signatures (ncclResult_t == int, ncclComm_t == opaque pointer)
struct UniqueId { char internal[MORL_COMM_ID_BYTES]; };
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(const void*, int*) = nullptr;     // optional (reporting only)
    bool ok = false;
    char why[256] = {0};                               // the dlopen / dlsym reason when !ok, captured where it happened
};
constexpr int kNcclFloat32 = 7, kNcclSum = 0;     // ncclDataType_t / ncclRedOp_t values of rccl.h

Rccl& rccl() {
    static Rccl r = [] {
        Rccl q;
        // MORL_RCCL_LIB: this RCCL build and no other (a site's own build; the fake one of tests/test_distributed.py, which fails on
        // a chosen rank to exercise the ranks' agreement on a common fall-back)
        if (const char* own = getenv("MORL_RCCL_LIB")) {
            q.handle = dlopen(own, RTLD_NOW | RTLD_LOCAL);
            if (!q.handle) {
                const char* e = dlerror();
                snprintf(q.why, sizeof(q.why), "MORL_RCCL_LIB=%s: %s", own, e ? e : "dlopen failed");
                return q;
            }
        }
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {                              // an instance that is already mapped first
            if (q.handle) break;
            q.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        }
        if (!q.handle) {
            const char* more[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : more) {
                q.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (q.handle) break;
            }
        }
        if (!q.handle) {
            const char* e = dlerror();                             // ONE call: dlerror() clears the message it returns
            snprintf(q.why, sizeof(q.why), "%s", e ? e : "dlopen failed");
            return q;
        }
        q.GetUniqueId = (decltype(q.GetUniqueId))dlsym(q.handle, "ncclGetUniqueId");
        q.CommInitRank = (decltype(q.CommInitRank))dlsym(q.handle, "ncclCommInitRank");
        q.CommDestroy = (decltype(q.CommDestroy))dlsym(q.handle, "ncclCommDestroy");
        q.AllGather = (decltype(q.AllGather))dlsym(q.handle, "ncclAllGather");
        q.AllReduce = (decltype(q.AllReduce))dlsym(q.handle, "ncclAllReduce");
        q.GetErrorString = (decltype(q.GetErrorString))dlsym(q.handle, "ncclGetErrorString");
        q.CommCount = (decltype(q.CommCount))dlsym(q.handle, "ncclCommCount");
        q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllGather && q.AllReduce;
        if (!q.ok) snprintf(q.why, sizeof(q.why), "librccl is loaded but lacks one of ncclGetUniqueId / ncclCommInitRank / "
                                                   "ncclCommDestroy / ncclAllGather / ncclAllReduce");
        return q;
    }();
    return r;
}

int rccl_fail(const char* what, int code) {
    Rccl& r = rccl();
    return fail(MORL_ERR_HIP, "%s failed: %s", what, r.GetErrorString ? r.GetErrorString(code) : "RCCL error");
}

}  // namespace

// A communicator = (rank, world) + a TRANSPORT: two functions that enqueue the all-gather / the in-place sum in stream order.
// RCCL is one instance (rccl_allgather / rccl_allreduce below), the loopback of one rank another (a copy / nothing), and
// morl_comm_init_custom takes the caller's (torch.distributed over gloo in the CPU tests, over its own RCCL communicator with
// MORL_COMM=torch): the rank step of morl_envelope_step_sharded / _batch_sharded is the same code over any of them.
struct morl_comm {
    void* nccl = nullptr;
    morl_allgather_fn allgather = nullptr;
    morl_allreduce_fn allreduce = nullptr;
    void* user = nullptr;
    bool custom = false;                 // the caller's call-backs: their return codes are not ours
    int rank = 0, world = 1;
    int nccl_count = -1;                 // ncclCommCount of an RCCL communicator: the rank count RCCL itself computes with
    hipStream_t side = nullptr;          // the all-gather runs here, beside the training forward on the caller's stream
    hipEvent_t ready = nullptr, done = nullptr;
};

namespace {
int rccl_allgather(void* user, const float* send, float* recv, int64_t count_per_rank, void* stream) {
    morl_comm* c = (morl_comm*)user;
    const int rc = rccl().AllGather(send, recv, (size_t)count_per_rank, kNcclFloat32, c->nccl, (hipStream_t)stream);
    return rc ? rccl_fail("ncclAllGather", rc) : MORL_OK;
}
int rccl_allreduce(void* user, float* buf, int64_t count, void* stream) {
    morl_comm* c = (morl_comm*)user;
    const int rc = rccl().AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->nccl, (hipStream_t)stream);
    return rc ? rccl_fail("ncclAllReduce", rc) : MORL_OK;
}
int loop_allgather(void*, const float* send, float* recv, int64_t count_per_rank, void* stream) {
    if (send != recv) HIP_TRY(hipMemcpyAsync(recv, send, (size_t)count_per_rank * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MORL_OK;
}
int loop_allreduce(void*, float*, int64_t, void*) { return MORL_OK; }     // the sum over one rank

int finish_comm(morl_comm* c, morl_comm** out) {
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        morl_comm_destroy(c);
        return fail(MORL_ERR_HIP, "stream / event creation failed");
    }
    *out = c;
    return MORL_OK;
}
}  // namespace

extern "C" int morl_comm_unique_id(void* id_out) {
    if (!id_out) return fail(MORL_ERR_ARG, "id_out is NULL");
    Rccl& r = rccl();
    if (!r.ok) return fail(MORL_ERR_STATE, "RCCL (librccl.so.1) could not be loaded: %s", r.why);
    UniqueId id;
    const int rc = r.GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, &id, MORL_COMM_ID_BYTES);
    return MORL_OK;
}

extern "C" int morl_comm_init(morl_comm** out, const void* unique_id, int rank, int world) {
    if (!out || !unique_id) return fail(MORL_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(MORL_ERR_ARG, "rank %d / world %d", rank, world);
    // an all-zero id with world == 1: a loopback communicator that never touches RCCL (the all-gather of one rank is a copy,
    // its all-reduce nothing) -- single-rank runs on machines without RCCL, and the emulated build of the CPU tests
    bool loopback = world == 1;
    for (int k = 0; k < MORL_COMM_ID_BYTES && loopback; ++k) loopback = ((const char*)unique_id)[k] == 0;
    Rccl* r = nullptr;
    if (!loopback) {
        r = &rccl();
        if (!r->ok) return fail(MORL_ERR_STATE, "RCCL (librccl.so.1) could not be loaded: %s", r->why);
    }
    morl_comm* c = new (std::nothrow) morl_comm();
    if (!c) return fail(MORL_ERR_ALLOC, "out of host memory");
    c->rank = rank; c->world = world;
    if (!loopback) {
        UniqueId id;
        std::memcpy(&id, unique_id, MORL_COMM_ID_BYTES);
        const int rc = r->CommInitRank(&c->nccl, world, id, rank);    // blocks until all `world` ranks have joined
        if (rc) { delete c; return rccl_fail("ncclCommInitRank", rc); }
        c->allgather = rccl_allgather; c->allreduce = rccl_allreduce; c->user = c;
        // RCCL's own view of the job, for the log of a multi-GPU run (the rank count the library computes with, not the launcher's)
        int n = -1;
        if (r->CommCount) (void)r->CommCount(c->nccl, &n);
        c->nccl_count = n;
        fprintf(stderr, "[morl_comm] rank %d: RCCL communicator of %d rank(s) (ncclCommCount), asked for %d\n", rank, n, world);
        fflush(stderr);
        if (n > 0 && n != world) {
            (void)r->CommDestroy(c->nccl);
            delete c;
            return fail(MORL_ERR_STATE, "RCCL reports a communicator of %d rank(s), %d were asked for", n, world);
        }
    } else {
        c->allgather = loop_allgather; c->allreduce = loop_allreduce;
    }
    return finish_comm(c, out);
}

// A communicator over the CALLER's transport: `allgather(user, send, recv, count_per_rank, stream)` must make recv
// [world][count_per_rank] complete and `allreduce(user, buf, count, stream)` must leave the sum over the ranks in buf, both in
// stream order on the `stream` they are handed (a host transport -- gloo in the CPU tests -- simply performs them: the emulated
// build executes launches synchronously) and return 0, anything else fails the step with MORL_ERR_STATE.  The rank step then
// runs exactly as over RCCL.
extern "C" int morl_comm_init_custom(morl_comm** out, int rank, int world, morl_allgather_fn allgather,
                                     morl_allreduce_fn allreduce, void* user) {
    if (!out || !allgather || !allreduce) return fail(MORL_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(MORL_ERR_ARG, "rank %d / world %d", rank, world);
    morl_comm* c = new (std::nothrow) morl_comm();
    if (!c) return fail(MORL_ERR_ALLOC, "out of host memory");
    c->rank = rank; c->world = world;
    c->allgather = allgather; c->allreduce = allreduce; c->user = user; c->custom = true;
    return finish_comm(c, out);
}

static void ipc_release(morl_comm* c);
extern "C" int morl_comm_destroy(morl_comm* c) {
    if (!c) return MORL_OK;
    ipc_release(c);
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    delete c;
    return MORL_OK;
}

extern "C" int morl_comm_size(const morl_comm* c, int* rank, int* world) {
    if (!c) return fail(MORL_ERR_ARG, "comm is NULL");
    if (rank) *rank = c->rank;
    // (an RCCL communicator answers with RCCL's own count -- ncclCommCount --, so that a multi-GPU benchmark line can state how many
    // ranks the library's collectives really ran over: bench.py config.rccl_ranks)
    if (world) *world = (c->nccl && c->nccl_count > 0) ? c->nccl_count : c->world;
    return MORL_OK;
}

// all-gather of the ranks' next-state slabs: issued on the communicator's side stream once everything enqueued on `stream`
// so far (the slabs launch) has finished; morl_comm_wait makes `stream` wait for it.  What the caller enqueues on `stream`
// in between (the training forward of its own rows) overlaps the exchange.
extern "C" int morl_allgather_q_begin(morl_comm* c, const float* send, float* recv, int64_t count_per_rank, void* stream) {
    if (!c || !send || !recv || count_per_rank < 1) return fail(MORL_ERR_ARG, "allgather: bad argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipEventRecord(c->ready, s));
    HIP_TRY(hipStreamWaitEvent(c->side, c->ready, 0));
    const int rc = c->allgather(c->user, send, recv, count_per_rank, (void*)c->side);
    if (rc) return c->custom ? fail(MORL_ERR_STATE, "the transport's all-gather call-back failed (%d)", rc) : rc;
    HIP_TRY(hipEventRecord(c->done, c->side));
    return MORL_OK;
}

extern "C" int morl_comm_wait(morl_comm* c, void* stream) {
    if (!c) return fail(MORL_ERR_ARG, "comm is NULL");
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->done, 0));
    return MORL_OK;
}

// in-place sum of the flat [gradient | loss | priorities] buffer over the ranks, on the caller's stream (everything after it
// depends on it)
extern "C" int morl_allreduce_grads(morl_comm* c, float* buf, int64_t count, void* stream) {
    if (!c || !buf || count < 1) return fail(MORL_ERR_ARG, "allreduce: bad argument");
    const int rc = c->allreduce(c->user, buf, count, stream);
    if (rc) return c->custom ? fail(MORL_ERR_STATE, "the transport's all-reduce call-back failed (%d)", rc) : rc;
    return MORL_OK;
}

// ================================================================================================================================
// Single-hop transport over peer-mapped memory: the messages of the sharded step are 0.15 - 2.4 MB, far below
// the size at which a ring is bandwidth-bound on the seven point-to-point xGMI links of a GPU -- they are latency-bound, and a
// ring / tree pays its latency once per hop.  Here every rank owns ONE region of device memory that all its peers map (hipIpc);
// a collective is two or three small kernels that write (or read) the peers' regions directly, every link busy at once:
//
//   all-gather   push   : my slabs -> slot [me] of EVERY peer's gather area, then one flag per peer          (1 hop)
//                collect: wait for the world's flags, gather area -> the caller's buffer (local copy)
//   all-reduce   push   : chunk j of my buffer -> slot [me] of peer j's inbox, then one flag per peer         (reduce-scatter, 1 hop)
//                reduce : wait; my chunk = sum over the inbox slots IN RANK ORDER (one rank computes a chunk, so every replica
//                         receives the same bits) -> my outbox and my buffer; one flag per peer
//                pull   : wait; peer j's outbox -> chunk j of my buffer (remote reads)                         (all-gather, 1 hop)
//
// Flags carry the epoch (call count) of the collective, one 64-byte line per (phase, source); a kernel's last workgroup to finish
// its writes (device-scope ticket behind a system-scope fence) stores them with release semantics, waiters poll with acquire loads
// -- and give up after IPC_TIMEOUT (wall clock): a lost peer costs a wrong step and an error code (morl_comm_check), never a hung
// GPU.  The region is fine-grained (uncached) device memory so that a peer's writes are visible to a running kernel.
// The reference has no counterpart (common/morl_algorithm.py:42: one device).
// ================================================================================================================================
namespace {

constexpr int IPC_MAX_WORLD = 8;
constexpr int IPC_THREADS = 256;
constexpr long long IPC_TIMEOUT_TICKS = 300000000ll;       // wall_clock64() ticks at 100 MHz: 3 s (MORL_IPC_TIMEOUT_MS overrides: ranks
                                                            // that SHARE a device, as in the tests, wait on each other's time slices)
enum { IPC_PH_AG = 0, IPC_PH_RS = 1, IPC_PH_AR = 2 };

struct IpcHeader {                                  // at the start of every rank's region
    unsigned int flag[3][IPC_MAX_WORLD][16];        // [phase][source rank]: the last epoch that source signalled (64-byte lines)
    unsigned int error;                             // 1 + phase of a wait that ran out
    unsigned int pad[15];
};
struct IpcPeers { float* base[IPC_MAX_WORLD]; };    // every rank's region as mapped HERE (base[me] = my own)
struct IpcGeom {
    long long inbox_off, outbox_off, gather_off;    // float offsets from the region base
    long long chunk_cap, ag_cap;                    // floats per inbox slot / per gather slot
    long long timeout_ticks;                        // bound of every wait
    unsigned int* host_err;                         // mirror of IpcHeader::error in mapped host memory: the host reads it WITHOUT
                                                    // synchronising (morl_comm_poll)
    int rank, world;
};

__device__ __forceinline__ IpcHeader* ipc_hdr(float* base) { return reinterpret_cast<IpcHeader*>(base); }

// every workgroup calls this after its last write: the last one to arrive tells every peer that this rank's part of `phase` is
// complete for `epoch` (and re-arms the ticket)
__device__ __forceinline__ void ipc_finish_and_signal(const IpcPeers& P, const IpcGeom& g, int phase, unsigned epoch,
                                                      unsigned* ticket, unsigned n_blocks) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(ticket, 1u);
        if (prev + 1u == n_blocks) {
            *ticket = 0u;
            __threadfence_system();
            for (int j = 0; j < g.world; ++j)
                __hip_atomic_store(&ipc_hdr(P.base[j])->flag[phase][g.rank][0], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// the workgroup waits until every rank has signalled `epoch` for `phase` in MY header (bounded)
// A wait that runs out is NOT silent: the error word (device copy: the sharded steps' clip + Adam launch reads it and leaves the
// optimiser state alone -- morl_host::comm_error_word; host mirror: morl_comm_poll, no synchronisation) says which phase.
__device__ __forceinline__ void ipc_wait(IpcHeader* mine, int world, int phase, unsigned epoch, long long timeout_ticks,
                                         unsigned int* host_err) {
    if ((int)threadIdx.x < world) {
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(&mine->flag[phase][threadIdx.x][0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(63);
            if (wall_clock64() - t0 > timeout_ticks) {
                __hip_atomic_store(&mine->error, 1u + (unsigned)phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (host_err != nullptr) __hip_atomic_store(host_err, 1u + (unsigned)phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}

// grid-strided copy of `len` floats, 16 bytes at a time when both ends allow it (slots and chunks are multiples of 4 floats; the
// caller's buffers usually are 16-byte aligned)
__device__ __forceinline__ void ipc_copy(float* __restrict__ dst, const float* __restrict__ src, long long len) {
    const long long t = (long long)blockIdx.x * IPC_THREADS + threadIdx.x, stride = (long long)gridDim.x * IPC_THREADS;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        const long long n4 = len >> 2;
        for (long long e = t; e < n4; e += stride) reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[e];
        for (long long e = (n4 << 2) + t; e < len; e += stride) dst[e] = src[e];
    } else {
        for (long long e = t; e < len; e += stride) dst[e] = src[e];
    }
}

// push: blockIdx.y = destination rank j; floats [src_start(j), +len(j)) of `src` -> P.base[j] + dst_off + rank * slot_cap.
// all-gather: every destination gets the same `count` floats; reduce-scatter: destination j gets chunk j.
__global__ __launch_bounds__(IPC_THREADS) void ipc_push_kernel(IpcPeers P, IpcGeom g, const float* __restrict__ src, long long count,
                                                               long long chunk, int scatter, long long dst_off, long long slot_cap,
                                                               int phase, unsigned epoch, unsigned* ticket) {
    const int j = (int)blockIdx.y;
    const long long start = scatter ? (j * chunk < count ? j * chunk : count) : 0;
    const long long len = scatter ? (count - start < chunk ? count - start : chunk) : count;
    float* dst = P.base[j] + dst_off + (long long)g.rank * slot_cap;
    ipc_copy(dst, src + start, len);
    ipc_finish_and_signal(P, g, phase, epoch, ticket, gridDim.x * gridDim.y);
}

__global__ __launch_bounds__(IPC_THREADS) void ipc_collect_kernel(IpcPeers P, IpcGeom g, float* __restrict__ recv, long long count,
                                                                  unsigned epoch) {
    float* mine = P.base[g.rank];
    ipc_wait(ipc_hdr(mine), g.world, IPC_PH_AG, epoch, g.timeout_ticks, g.host_err);
    // the gather area is double-buffered by the parity of the epoch: a fast peer's push of call N + 1 must not land in the slots this
    // rank is still copying out for call N (peers are at most one call apart: nobody passes collect N + 1 before this rank has
    // pushed N + 1, which its stream orders behind this kernel)
    const float* gather = mine + g.gather_off + (long long)(epoch & 1u) * g.world * g.ag_cap;
    for (int r = 0; r < g.world; ++r) ipc_copy(recv + (long long)r * count, gather + (long long)r * g.ag_cap, count);
}

__global__ __launch_bounds__(IPC_THREADS) void ipc_reduce_kernel(IpcPeers P, IpcGeom g, float* __restrict__ buf, long long count,
                                                                 long long chunk, unsigned epoch, unsigned* ticket) {
    float* mine = P.base[g.rank];
    ipc_wait(ipc_hdr(mine), g.world, IPC_PH_RS, epoch, g.timeout_ticks, g.host_err);
    const long long start = g.rank * chunk < count ? g.rank * chunk : count;
    const long long len = count - start < chunk ? count - start : chunk;
    const float* inbox = mine + g.inbox_off;
    float* outbox = mine + g.outbox_off;
    for (long long e = (long long)blockIdx.x * IPC_THREADS + threadIdx.x; e < len; e += (long long)gridDim.x * IPC_THREADS) {
        float s = inbox[e];                                   // rank order: the same bits whoever asks
        for (int r = 1; r < g.world; ++r) s += inbox[(long long)r * g.chunk_cap + e];
        outbox[e] = s;
        buf[start + e] = s;
    }
    ipc_finish_and_signal(P, g, IPC_PH_AR, epoch, ticket, gridDim.x);
}

__global__ __launch_bounds__(IPC_THREADS) void ipc_pull_kernel(IpcPeers P, IpcGeom g, float* __restrict__ buf, long long count,
                                                               long long chunk, unsigned epoch) {
    ipc_wait(ipc_hdr(P.base[g.rank]), g.world, IPC_PH_AR, epoch, g.timeout_ticks, g.host_err);
    const int j = (int)blockIdx.y;
    if (j == g.rank) return;
    const long long start = j * chunk < count ? j * chunk : count;
    const long long len = count - start < chunk ? count - start : chunk;
    ipc_copy(buf + start, P.base[j] + g.outbox_off, len);
}

struct IpcState {
    IpcPeers peers{};
    IpcGeom geom{};
    void* region = nullptr;
    size_t region_bytes = 0;
    unsigned int* host_err = nullptr;     // host address of the mapped error word (IpcGeom::host_err is its device address)
    bool opened[IPC_MAX_WORLD] = {};
    unsigned* tickets = nullptr;          // [2] device counters of the last-workgroup pattern
    unsigned epoch_ag = 0, epoch_ar = 0;
    long long max_ar = 0, max_ag = 0;
    bool connected = false;
};

long long round4(long long v) { return (v + 3) / 4 * 4; }

}  // namespace

// (kept outside morl_comm's definition above: a side table keyed by the communicator, so the struct shared with the other
// transports stays what it was)
#include <map>
#include <mutex>
namespace {
std::mutex g_ipc_mu;
std::map<const morl_comm*, IpcState*> g_ipc;
IpcState* ipc_of(const morl_comm* c) {
    std::lock_guard<std::mutex> lk(g_ipc_mu);
    auto it = g_ipc.find(c);
    return it == g_ipc.end() ? nullptr : it->second;
}

int ipc_allgather(void* user, const float* send, float* recv, int64_t count, void* stream) {
    morl_comm* c = (morl_comm*)user;
    IpcState* st = ipc_of(c);
    if (!st || !st->connected) return fail(MORL_ERR_STATE, "ipc communicator is not connected");
    if (count > st->max_ag) return fail(MORL_ERR_STATE, "all-gather of %lld floats per rank, the region holds %lld", (long long)count, st->max_ag);
    const unsigned epoch = ++st->epoch_ag;
    const int gx = (int)std::max<long long>(1, std::min<long long>(128, (count + IPC_THREADS * 4 - 1) / (IPC_THREADS * 4)));
    hipLaunchKernelGGL(ipc_push_kernel, dim3(gx, st->geom.world), dim3(IPC_THREADS), 0, (hipStream_t)stream, st->peers, st->geom, send,
                       (long long)count, 0ll, 0, st->geom.gather_off + (long long)(epoch & 1u) * st->geom.world * st->geom.ag_cap, st->geom.ag_cap,
                       (int)IPC_PH_AG, epoch, st->tickets + 0);
    LAUNCH_CHECK("ipc_push(all-gather)");
    const int gc = (int)std::max<long long>(1, std::min<long long>(256, (count + IPC_THREADS * 4 - 1) / (IPC_THREADS * 4)));
    hipLaunchKernelGGL(ipc_collect_kernel, dim3(gc), dim3(IPC_THREADS), 0, (hipStream_t)stream, st->peers, st->geom, recv, (long long)count,
                       epoch);
    LAUNCH_CHECK("ipc_collect");
    return MORL_OK;
}

int ipc_allreduce(void* user, float* buf, int64_t count, void* stream) {
    morl_comm* c = (morl_comm*)user;
    IpcState* st = ipc_of(c);
    if (!st || !st->connected) return fail(MORL_ERR_STATE, "ipc communicator is not connected");
    if (count > st->max_ar) return fail(MORL_ERR_STATE, "all-reduce of %lld floats, the region holds %lld", (long long)count, st->max_ar);
    const int world = st->geom.world;
    const long long chunk = round4(((long long)count + world - 1) / world);
    const unsigned epoch = ++st->epoch_ar;
    const int gx = (int)std::max<long long>(1, std::min<long long>(32, (chunk + IPC_THREADS * 4 - 1) / (IPC_THREADS * 4)));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ipc_push_kernel, dim3(gx, world), dim3(IPC_THREADS), 0, s, st->peers, st->geom, (const float*)buf, (long long)count,
                       chunk, 1, st->geom.inbox_off, st->geom.chunk_cap, (int)IPC_PH_RS, epoch, st->tickets + 0);
    LAUNCH_CHECK("ipc_push(reduce-scatter)");
    hipLaunchKernelGGL(ipc_reduce_kernel, dim3(gx), dim3(IPC_THREADS), 0, s, st->peers, st->geom, buf, (long long)count, chunk, epoch,
                       st->tickets + 1);
    LAUNCH_CHECK("ipc_reduce");
    hipLaunchKernelGGL(ipc_pull_kernel, dim3(gx, world), dim3(IPC_THREADS), 0, s, st->peers, st->geom, buf, (long long)count, chunk, epoch);
    LAUNCH_CHECK("ipc_pull");
    return MORL_OK;
}
}  // namespace

extern "C" int morl_comm_ipc_create(morl_comm** out, int rank, int world, int64_t max_allreduce_floats, int64_t max_allgather_floats,
                                    void* handle_out) {
    if (!out || !handle_out) return fail(MORL_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (world < 1 || world > IPC_MAX_WORLD || rank < 0 || rank >= world)
        return fail(MORL_ERR_ARG, "rank %d / world %d (at most %d ranks)", rank, world, IPC_MAX_WORLD);
    if (max_allreduce_floats < 1 || max_allgather_floats < 0) return fail(MORL_ERR_ARG, "bad buffer sizes");
    morl_comm* c = new (std::nothrow) morl_comm();
    IpcState* st = new (std::nothrow) IpcState();
    if (!c || !st) { delete c; delete st; return fail(MORL_ERR_ALLOC, "out of host memory"); }
    c->rank = rank; c->world = world;
    st->max_ar = max_allreduce_floats; st->max_ag = max_allgather_floats;
    IpcGeom& g = st->geom;
    g.rank = rank; g.world = world;
    g.timeout_ticks = IPC_TIMEOUT_TICKS;
    if (const char* e = getenv("MORL_IPC_TIMEOUT_MS")) g.timeout_ticks = std::max(1ll, atoll(e)) * 100000ll;
    g.chunk_cap = round4((max_allreduce_floats + world - 1) / world);
    g.ag_cap = round4(std::max<int64_t>(max_allgather_floats, 4));
    g.inbox_off = (long long)(sizeof(IpcHeader) / sizeof(float));
    g.outbox_off = g.inbox_off + (long long)world * g.chunk_cap;
    g.gather_off = g.outbox_off + g.chunk_cap;
    st->region_bytes = (size_t)(g.gather_off + 2ll * world * g.ag_cap) * sizeof(float);       // (two gather areas: epoch parity)
    // fine-grained (uncached) device memory: a peer's writes must become visible to a kernel that is already running.  Ordinary
    // (coarse-grained) memory gives no such guarantee, so there is no fall-back to it: without this allocation the transport is
    // unavailable and the caller takes another one (distributed.make_comm)
    if (hipExtMallocWithFlags(&st->region, st->region_bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        delete c; delete st;
        return fail(MORL_ERR_ALLOC, "hipExtMallocWithFlags(%zu bytes, uncached) failed: no fine-grained device memory for the shared region", st->region_bytes);
    }
    {
        void* h_err = nullptr;
        if (hipHostMalloc(&h_err, 64, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(st->region); delete c; delete st;
            return fail(MORL_ERR_ALLOC, "hipHostMalloc(mapped) of the error word failed");
        }
        std::memset(h_err, 0, 64);
        st->host_err = (unsigned int*)h_err;
        void* d_err = nullptr;
        if (hipHostGetDevicePointer(&d_err, h_err, 0) != hipSuccess) {
            (void)hipHostFree(h_err); (void)hipFree(st->region); delete c; delete st;
            return fail(MORL_ERR_HIP, "hipHostGetDevicePointer failed");
        }
        g.host_err = (unsigned int*)d_err;
    }
    if (hipMemset(st->region, 0, st->region_bytes) != hipSuccess || hipMalloc((void**)&st->tickets, 2 * sizeof(unsigned)) != hipSuccess ||
        hipMemset(st->tickets, 0, 2 * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(st->region); (void)hipHostFree(st->host_err); delete c; delete st;
        return fail(MORL_ERR_HIP, "setting up the shared region failed");
    }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, st->region) != hipSuccess) {
        (void)hipFree(st->region); (void)hipFree(st->tickets); (void)hipHostFree(st->host_err); delete c; delete st;
        return fail(MORL_ERR_HIP, "hipIpcGetMemHandle failed (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    }
    static_assert(sizeof(hipIpcMemHandle_t) <= MORL_COMM_IPC_HANDLE_BYTES, "handle size");
    std::memset(handle_out, 0, MORL_COMM_IPC_HANDLE_BYTES);
    std::memcpy(handle_out, &h, sizeof(h));
    st->peers.base[rank] = (float*)st->region;
    c->allgather = ipc_allgather; c->allreduce = ipc_allreduce; c->user = c;
    { std::lock_guard<std::mutex> lk(g_ipc_mu); g_ipc[c] = st; }
    const int rc = finish_comm(c, out);
    if (rc) { std::lock_guard<std::mutex> lk(g_ipc_mu); g_ipc.erase(c); }
    return rc;
}

extern "C" int morl_comm_ipc_connect(morl_comm* c, const void* all_handles) {
    IpcState* st = c ? ipc_of(c) : nullptr;
    if (!st || !all_handles) return fail(MORL_ERR_ARG, "not an ipc communicator / NULL handles");
    for (int j = 0; j < st->geom.world; ++j) {
        if (j == st->geom.rank || st->opened[j]) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char*)all_handles + (size_t)j * MORL_COMM_IPC_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(MORL_ERR_HIP, "hipIpcOpenMemHandle(rank %d) failed: %s", j, hipGetErrorString(e));
        st->peers.base[j] = (float*)p;
        st->opened[j] = true;
    }
    st->connected = true;
    return MORL_OK;
}

const unsigned int* morl_host::comm_error_word(const morl_comm* c) {
    IpcState* st = c ? ipc_of(c) : nullptr;
    return (st && st->region) ? &reinterpret_cast<const IpcHeader*>(st->region)->error : nullptr;
}

// the same verdict as morl_comm_check WITHOUT synchronising: reads the host mirror of the error word, i.e. what the collectives that
// have EXECUTED so far found (cheap enough to call before every step; the training loops of distributed.py do)
extern "C" int morl_comm_poll(morl_comm* c) {
    if (!c) return fail(MORL_ERR_ARG, "comm is NULL");
    IpcState* st = ipc_of(c);
    if (!st || !st->host_err) return MORL_OK;
    const unsigned e = __atomic_load_n(st->host_err, __ATOMIC_RELAXED);
    if (e) return fail(MORL_ERR_STATE, "a peer did not arrive within the time limit (phase %u of the single-hop collectives); the step that "
                                       "waited for it was NOT applied on this rank", e - 1);
    return MORL_OK;
}

// 0 if no bounded wait of this rank's collectives has run out so far; synchronises the device
extern "C" int morl_comm_check(morl_comm* c) {
    IpcState* st = c ? ipc_of(c) : nullptr;
    if (!c) return fail(MORL_ERR_ARG, "comm is NULL");
    if (!st) return MORL_OK;                       // the other transports report their failures at the call
    IpcHeader h;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&h, st->region, sizeof(h), hipMemcpyDeviceToHost));
    if (h.error) return fail(MORL_ERR_STATE, "a peer did not arrive within the time limit (phase %u of the single-hop collectives)", h.error - 1);
    return MORL_OK;
}

static void ipc_release(morl_comm* c) {
    IpcState* st = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ipc_mu);
        auto it = g_ipc.find(c);
        if (it == g_ipc.end()) return;
        st = it->second;
        g_ipc.erase(it);
    }
    (void)hipDeviceSynchronize();
    for (int j = 0; j < st->geom.world; ++j)
        if (st->opened[j]) (void)hipIpcCloseMemHandle(st->peers.base[j]);
    if (st->region) (void)hipFree(st->region);
    if (st->tickets) (void)hipFree(st->tickets);
    if (st->host_err) (void)hipHostFree(st->host_err);
    delete st;
}
