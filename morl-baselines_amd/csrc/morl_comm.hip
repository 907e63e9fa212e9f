// libmorl_hip.so, third translation unit: the collectives of the weight-sharded Envelope step behind the C ABI
// (include/morl_hip.h, "multi-GPU").  One process per GPU; the communicator is RCCL over xGMI.  RCCL is bound at RUN TIME
// (dlopen, preferring an instance the process has already loaded -- PyTorch ships its own librccl.so.1 and two instances in
// one process would each own half of the node's peer mappings): libmorl_hip.so itself has no link-time dependency on it,
// single-GPU users never load it.  Nothing here synchronises the host; every call enqueues on the caller's stream.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
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
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {                              // an instance that is already mapped first
            q.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (q.handle) break;
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
        fprintf(stderr, "[morl_comm] rank %d: RCCL communicator of %d rank(s) (ncclCommCount), asked for %d\n", rank, n, world);
        fflush(stderr);
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

extern "C" int morl_comm_destroy(morl_comm* c) {
    if (!c) return MORL_OK;
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
    if (world) *world = c->world;
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
