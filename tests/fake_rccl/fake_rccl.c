/* TEST INFRASTRUCTURE: a stand-in for librccl that never touches a device (tests/test_distributed.py loads it through MORL_RCCL_LIB).
 * ncclGetUniqueId / ncclCommInitRank succeed or fail as the environment says -- FAKE_RCCL_FAIL_ID=1: no unique id on rank 0;
 * FAKE_RCCL_FAIL_INIT_RANK=k: ncclCommInitRank fails on rank k, at once, and succeeds at once everywhere else -- so that the ranks'
 * agreement on a common outcome (NativeComm / make_comm of morl-baselines_amd/distributed.py) can be exercised on every ordering
 * without GPUs.  The collectives themselves are not implemented (a communicator that came up is only asked for its size). */
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct { int rank, world; } fake_comm;

int ncclGetUniqueId(ncclUniqueId* id) {
    const char* f = getenv("FAKE_RCCL_FAIL_ID");
    if (f && atoi(f)) return 2;                      /* ncclSystemError */
    memset(id->internal, 0x5a, sizeof(id->internal));
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    const char* f = getenv("FAKE_RCCL_FAIL_INIT_RANK");
    if (id.internal[0] != 0x5a) return 4;            /* ncclInvalidArgument: not the id rank 0 drew */
    if (f && atoi(f) == rank) return 1;              /* ncclUnhandledCudaError */
    fake_comm* c = (fake_comm*)malloc(sizeof(fake_comm));
    c->rank = rank; c->world = nranks;
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) { free(comm); return 0; }
int ncclCommCount(const void* comm, int* count) { *count = ((const fake_comm*)comm)->world; return 0; }
int ncclAllGather(const void* s, void* r, size_t n, int t, void* c, void* st) { (void)s; (void)r; (void)n; (void)t; (void)c; (void)st; return 3; }
int ncclAllReduce(const void* s, void* r, size_t n, int t, int op, void* c, void* st) { (void)s; (void)r; (void)n; (void)t; (void)op; (void)c; (void)st; return 3; }
const char* ncclGetErrorString(int code) { return code == 1 ? "fake: device error" : code == 2 ? "fake: system error" : "fake: error"; }
