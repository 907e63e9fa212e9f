// Host-side plumbing shared by the translation units of libmorl_hip.so: error reporting, launch checks, small helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "morl_hip.h"

namespace morl_host {

// one message buffer per host thread, shared by every translation unit (defined in morl_hip.hip)
extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// entries one sum-tree update launch holds (ST_MAX_B of replay_kernels.h; asserted equal in morl_hip.hip)
constexpr int TREE_UPDATE_MAX = 1024;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline int vec_ok(const void* p, int ld) { return (((uintptr_t)p & 15u) == 0 && (ld & 3) == 0) ? 1 : 0; }

inline int stream_grid(long long n, int threads, int cap = 2048) {
    long long b = (n + threads - 1) / threads;
    return (int)std::max(1ll, std::min<long long>(b, cap));
}

inline int dmalloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(MORL_ERR_ALLOC, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return MORL_OK;
}

// device address of the word a communicator's bounded waits set when one runs out (single-hop transport), or NULL for a transport
// that reports its failures at the call (defined in morl_comm.hip; the sharded steps hand it to their clip + Adam launch)
const unsigned int* comm_error_word(const morl_comm* c);

}  // namespace morl_host

#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return morl_host::fail(MORL_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCH_CHECK(name)                                                                                         \
    do {                                                                                                           \
        hipError_t e_ = hipGetLastError();                                                                         \
        if (e_ != hipSuccess) return morl_host::fail(MORL_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_));  \
    } while (0)
