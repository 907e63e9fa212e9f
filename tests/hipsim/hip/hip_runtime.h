// hipsim -- TEST INFRASTRUCTURE ONLY.
//
// A tiny wave-level SIMT emulator that lets the *unmodified* kernel sources under
// morl-baselines_amd/csrc/ be compiled for the x86 host (clang++ -x c++ -I tests/hipsim ...) so
// that their index arithmetic, LDS tiling, MFMA fragment maps and cross-lane reductions can be
// checked against the oracle in the GPU-less build container (pytest -m "not gpu").
//
// It is NOT a product path: the shipped library is built by hipcc for gfx950 only, nothing in
// morl-baselines_amd/ loads the emulated build by default, and bench.py / smoke() never do.
//
// Model: one OS thread; every work-item of a workgroup is a fiber (ucontext).  Blocks run one
// after another.  __syncthreads() and the wave collectives (__shfl*, __ballot, __all, __any,
// MFMA) are rendezvous points: a fiber yields until all work-items of its block / wave arrived.
// Wave size is 64.  The f32 MFMA fragment maps are the gfx950 ones
// (guide: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31).
#pragma once
#define HIPSIM_EMULATED 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

static inline long long clock64() { return 0; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // only used on wave-uniform values
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline const char* hipGetErrorString(hipError_t) { return "hipsim error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
namespace hipsim { bool shm_free(void* p); }
static inline hipError_t hipFree(void* p) { if (!hipsim::shm_free(p)) std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
enum { hipHostMallocMapped = 2 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventDisableSystemFence = 0x20000000 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(void** e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, void*, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }

// ---- "peer-mapped device memory" of the emulated build: POSIX shared memory, so that the single-hop collectives of
// morl_comm.hip (hipIpc between the ranks of a job) run between the PROCESSES of a gloo CPU test exactly as they do between GPUs
#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
enum { hipDeviceMallocFinegrained = 1, hipDeviceMallocUncached = 3, hipIpcMemLazyEnablePeerAccess = 1 };
struct hipIpcMemHandle_t { char reserved[64]; };
namespace hipsim {
struct ShmRec { void* ptr; size_t bytes; char name[48]; bool owner; };
inline ShmRec* shm_table() { static ShmRec t[64]; return t; }
inline ShmRec* shm_find(const void* p) { for (int k = 0; k < 64; ++k) if (shm_table()[k].ptr == p && p) return &shm_table()[k]; return nullptr; }
inline ShmRec* shm_slot() { for (int k = 0; k < 64; ++k) if (!shm_table()[k].ptr) return &shm_table()[k]; return nullptr; }
}
static inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) {
    static int counter = 0;
    hipsim::ShmRec* r = hipsim::shm_slot();
    if (!r) return hipErrorOutOfMemory;
    std::snprintf(r->name, sizeof(r->name), "/hipsim_%d_%d", (int)getpid(), counter++);
    const int fd = shm_open(r->name, O_CREAT | O_RDWR | O_EXCL, 0600);
    if (fd < 0) return hipErrorOutOfMemory;
    if (ftruncate(fd, (off_t)n) != 0) { close(fd); shm_unlink(r->name); return hipErrorOutOfMemory; }
    void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { shm_unlink(r->name); return hipErrorOutOfMemory; }
    r->ptr = m; r->bytes = n; r->owner = true;
    *p = m;
    return hipSuccess;
}
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
    hipsim::ShmRec* r = hipsim::shm_find(p);
    if (!r) return hipErrorInvalidValue;
    std::memset(h, 0, sizeof(*h));
    std::snprintf(h->reserved, sizeof(h->reserved), "%zu %s", r->bytes, r->name);
    return hipSuccess;
}
static inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
    size_t n = 0; char name[48] = {0};
    if (std::sscanf(h.reserved, "%zu %47s", &n, name) != 2) return hipErrorInvalidValue;
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return hipErrorInvalidValue;
    void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return hipErrorInvalidValue;
    hipsim::ShmRec* r = hipsim::shm_slot();
    if (r) { r->ptr = m; r->bytes = n; r->owner = false; std::snprintf(r->name, sizeof(r->name), "%s", name); }
    *p = m;
    return hipSuccess;
}
static inline hipError_t hipIpcCloseMemHandle(void* p) {
    hipsim::ShmRec* r = hipsim::shm_find(p);
    if (!r) return hipErrorInvalidValue;
    munmap(r->ptr, r->bytes);
    r->ptr = nullptr;
    return hipSuccess;
}
namespace hipsim {
inline bool shm_free(void* p) {
    ShmRec* r = shm_find(p);
    if (!r) return false;
    munmap(r->ptr, r->bytes);
    if (r->owner) shm_unlink(r->name);
    r->ptr = nullptr;
    return true;
}
}
static inline long long wall_clock64() {          // 100 MHz, like the device's constant-rate counter
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 100000000ll + ts.tv_nsec / 10;
}
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

struct hipDeviceProp_t { int multiProcessorCount; };
// (HIPSIM_CUS: a smaller chip, so that tilings chosen from the CU count -- the 64-row chain tiles, which want a tile per CU -- are reached
// with row counts the emulator can afford)
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    const char* e = getenv("HIPSIM_CUS");
    p->multiProcessorCount = (e && atoi(e) > 0) ? atoi(e) : 256;
    return hipSuccess;
}

namespace hipsim {
struct LaneCtx {
    uint3 tid, bid;
    int lane, wave;
};
struct CollIn { float f[24]; double d; unsigned long long u; int i; };
struct CollOut { float f[16]; double d; unsigned long long u; int i; };
extern LaneCtx* cur;
extern dim3 cur_block_dim, cur_grid_dim;
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
void block_barrier();
// generic wave rendezvous: every lane deposits `in`, the last arriver runs `fn(in[64], out[64], nlanes)`
CollOut wave_collective(const CollIn& in, void (*fn)(const CollIn*, CollOut*, int));
}  // namespace hipsim

#define threadIdx (hipsim::cur->tid)
#define blockIdx (hipsim::cur->bid)
#define blockDim (hipsim::cur_block_dim)
#define gridDim (hipsim::cur_grid_dim)
#define warpSize 64

static inline void __syncthreads() { hipsim::block_barrier(); }

template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
    hipsim::run_grid(grid, block, [&]() { kernel(args...); });
}

// ---- cross-lane ------------------------------------------------------------------------------
namespace hipsim {
static void fn_shfl_xor_f(const CollIn* in, CollOut* out, int n) {
    for (int l = 0; l < n; ++l) { int s = l ^ in[l].i; out[l].f[0] = (s < n) ? in[s].f[0] : in[l].f[0]; }
}
static void fn_shfl_xor_i(const CollIn* in, CollOut* out, int n) {
    for (int l = 0; l < n; ++l) { int s = l ^ in[l].i; out[l].u = (s < n) ? in[s].u : in[l].u; }
}
static void fn_shfl_xor_d(const CollIn* in, CollOut* out, int n) {
    for (int l = 0; l < n; ++l) { int s = l ^ in[l].i; out[l].d = (s < n) ? in[s].d : in[l].d; }
}
static void fn_shfl_idx_f(const CollIn* in, CollOut* out, int n) {
    for (int l = 0; l < n; ++l) { int s = in[l].i & 63; out[l].f[0] = (s < n) ? in[s].f[0] : in[l].f[0]; }
}
static void fn_shfl_idx_i(const CollIn* in, CollOut* out, int n) {
    for (int l = 0; l < n; ++l) { int s = in[l].i & 63; out[l].u = (s < n) ? in[s].u : in[l].u; }
}
static void fn_shfl_idx_d(const CollIn* in, CollOut* out, int n) {
    for (int l = 0; l < n; ++l) { int s = in[l].i & 63; out[l].d = (s < n) ? in[s].d : in[l].d; }
}
static void fn_ballot(const CollIn* in, CollOut* out, int n) {
    unsigned long long m = 0;
    for (int l = 0; l < n; ++l) if (in[l].i) m |= 1ull << l;
    for (int l = 0; l < n; ++l) out[l].u = m;
}
static void fn_mfma_32x32x2(const CollIn* in, CollOut* out, int n) {
    if (n != 64) { std::fprintf(stderr, "hipsim: MFMA needs a full wave (got %d lanes)\n", n); std::abort(); }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            float acc = in[l].f[2 + r];
            for (int k = 0; k < 2; ++k) acc = std::fmaf(in[row + 32 * k].f[0], in[col + 32 * k].f[1], acc);
            out[l].f[r] = acc;
        }
}
}  // namespace hipsim

static inline float __shfl_xor(float v, int m, int = 64) { hipsim::CollIn in{}; in.f[0] = v; in.i = m; return hipsim::wave_collective(in, hipsim::fn_shfl_xor_f).f[0]; }
static inline int __shfl_xor(int v, int m, int = 64) { hipsim::CollIn in{}; in.u = (unsigned)v; in.i = m; return (int)hipsim::wave_collective(in, hipsim::fn_shfl_xor_i).u; }
static inline unsigned __shfl_xor(unsigned v, int m, int = 64) { hipsim::CollIn in{}; in.u = v; in.i = m; return (unsigned)hipsim::wave_collective(in, hipsim::fn_shfl_xor_i).u; }
static inline double __shfl_xor(double v, int m, int = 64) { hipsim::CollIn in{}; in.d = v; in.i = m; return hipsim::wave_collective(in, hipsim::fn_shfl_xor_d).d; }
static inline float __shfl(float v, int s, int = 64) { hipsim::CollIn in{}; in.f[0] = v; in.i = s; return hipsim::wave_collective(in, hipsim::fn_shfl_idx_f).f[0]; }
static inline int __shfl(int v, int s, int = 64) { hipsim::CollIn in{}; in.u = (unsigned)v; in.i = s; return (int)hipsim::wave_collective(in, hipsim::fn_shfl_idx_i).u; }
static inline double __shfl(double v, int s, int = 64) { hipsim::CollIn in{}; in.d = v; in.i = s; return hipsim::wave_collective(in, hipsim::fn_shfl_idx_d).d; }
// wave-private LDS hand-offs: the fibres of a wave meet at a collective, fences are meaningless on the host
static inline void __builtin_amdgcn_wave_barrier() { hipsim::CollIn in{}; in.u = 0; in.i = 0; (void)hipsim::wave_collective(in, hipsim::fn_shfl_idx_i); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline unsigned long long __ballot(int p) { hipsim::CollIn in{}; in.i = p != 0; return hipsim::wave_collective(in, hipsim::fn_ballot).u; }
static inline int __all(int p) { return __ballot(!p) == 0ull; }
static inline int __any(int p) { return __ballot(p) != 0ull; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

typedef float hipsim_f32x16 __attribute__((ext_vector_type(16)));
static inline hipsim_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipsim_f32x16 c, int, int, int) {
    hipsim::CollIn in{};
    in.f[0] = a; in.f[1] = b;
    for (int r = 0; r < 16; ++r) in.f[2 + r] = c[r];
    hipsim::CollOut o = hipsim::wave_collective(in, hipsim::fn_mfma_32x32x2);
    hipsim_f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = o.f[r];
    return d;
}

namespace hipsim {
static void fn_mfma_16x16x4(const CollIn* in, CollOut* out, int n) {
    if (n != 64) { std::fprintf(stderr, "hipsim: MFMA needs a full wave (got %d lanes)\n", n); std::abort(); }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = (l >> 4) * 4 + r, col = l & 15;
            float acc = in[l].f[2 + r];
            for (int k = 0; k < 4; ++k) acc = std::fmaf(in[row + 16 * k].f[0], in[col + 16 * k].f[1], acc);
            out[l].f[r] = acc;
        }
}
}  // namespace hipsim
typedef float hipsim_f32x4 __attribute__((ext_vector_type(4)));
static inline hipsim_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipsim_f32x4 c, int, int, int) {
    hipsim::CollIn in{};
    in.f[0] = a; in.f[1] = b;
    for (int r = 0; r < 4; ++r) in.f[2 + r] = c[r];
    hipsim::CollOut o = hipsim::wave_collective(in, hipsim::fn_mfma_16x16x4);
    hipsim_f32x4 d;
    for (int r = 0; r < 4; ++r) d[r] = o.f[r];
    return d;
}

// 16 blocks of 4x4x1: block b = lane / 4; lane 4b + i supplies A_b[i] and B_b[i]; register r of lane 4b + j is D_b[r][j].
// cbsz = 4: the A operand of block `abid` is broadcast to all sixteen blocks (cbsz = 0: every block its own)
namespace hipsim {
static void fn_mfma_4x4x1(const CollIn* in, CollOut* out, int n) {
    if (n != 64) { std::fprintf(stderr, "hipsim: MFMA needs a full wave (got %d lanes)\n", n); std::abort(); }
    const int cbsz = in[0].i >> 8, abid = in[0].i & 255;
    if (cbsz != 0 && cbsz != 4) { std::fprintf(stderr, "hipsim: mfma 4x4x1 cbsz %d not emulated\n", cbsz); std::abort(); }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int a_lane = (cbsz == 4 ? 4 * abid : (l & ~3)) + r;
            out[l].f[r] = std::fmaf(in[a_lane].f[0], in[l].f[1], in[l].f[2 + r]);
        }
}
}  // namespace hipsim
static inline hipsim_f32x4 __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, hipsim_f32x4 c, int cbsz, int abid, int) {
    hipsim::CollIn in{};
    in.i = (cbsz << 8) | abid;
    in.f[0] = a; in.f[1] = b;
    for (int r = 0; r < 4; ++r) in.f[2 + r] = c[r];
    hipsim::CollOut o = hipsim::wave_collective(in, hipsim::fn_mfma_4x4x1);
    hipsim_f32x4 d;
    for (int r = 0; r < 4; ++r) d[r] = o.f[r];
    return d;
}

// v_mfma_f32_16x16x32_bf16: A lane l = row l & 15, eight k of group l >> 4; B lane l = column l & 15, the same eight k;
// D lane l: column l & 15, rows 4 (l >> 4) + r.  Products of bf16 are exact in fp32; the sum is formed in double and rounded once
// (the hardware's internal order is not specified; results agree with it to fp32 rounding)
namespace hipsim {
// CollIn::f[0..3] = C, the 32 bytes from f[4] on = the lane's eight A and eight B values (bf16)
static void fn_mfma_bf16_16x16x32(const CollIn* in, CollOut* out, int n) {
    if (n != 64) { std::fprintf(stderr, "hipsim: MFMA needs a full wave (got %d lanes)\n", n); std::abort(); }
    auto val = [&](int lane, int which, int e) {
        unsigned short h;
        std::memcpy(&h, reinterpret_cast<const char*>(&in[lane].f[4]) + 16 * which + 2 * e, 2);
        unsigned u = (unsigned)h << 16; float x; std::memcpy(&x, &u, 4); return (double)x;
    };
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            double acc = in[l].f[r];
            for (int g = 0; g < 4; ++g)
                for (int e = 0; e < 8; ++e) acc += val(row + 16 * g, 0, e) * val(col + 16 * g, 1, e);
            out[l].f[r] = (float)acc;
        }
}
}  // namespace hipsim
typedef __bf16 hipsim_bf16x8 __attribute__((ext_vector_type(8)));
static inline hipsim_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(hipsim_bf16x8 a, hipsim_bf16x8 b, hipsim_f32x4 c, int, int, int) {
    hipsim::CollIn in{};
    for (int r = 0; r < 4; ++r) in.f[r] = c[r];
    std::memcpy(reinterpret_cast<char*>(&in.f[4]), &a, 16);
    std::memcpy(reinterpret_cast<char*>(&in.f[4]) + 16, &b, 16);
    hipsim::CollOut o = hipsim::wave_collective(in, hipsim::fn_mfma_bf16_16x16x32);
    hipsim_f32x4 d;
    for (int r = 0; r < 4; ++r) d[r] = o.f[r];
    return d;
}
// v_mfma_f32_32x32x16_bf16: A lane l = row l & 31, eight k of group l >> 5; B lane l = column l & 31, the same eight k; D lane l:
// column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5).  CollIn::f[0..15] = C, then the lane's eight A and eight B values.
namespace hipsim {
static void fn_mfma_bf16_32x32x16(const CollIn* in, CollOut* out, int n) {
    if (n != 64) { std::fprintf(stderr, "hipsim: MFMA needs a full wave (got %d lanes)\n", n); std::abort(); }
    auto val = [&](int lane, int which, int e) {
        unsigned short h;
        std::memcpy(&h, reinterpret_cast<const char*>(&in[lane].f[16]) + 16 * which + 2 * e, 2);
        unsigned u = (unsigned)h << 16; float x; std::memcpy(&x, &u, 4); return (double)x;
    };
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            double acc = in[l].f[r];
            for (int g = 0; g < 2; ++g)
                for (int e = 0; e < 8; ++e) acc += val(row + 32 * g, 0, e) * val(col + 32 * g, 1, e);
            out[l].f[r] = (float)acc;
        }
}
}  // namespace hipsim
static inline hipsim_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipsim_bf16x8 a, hipsim_bf16x8 b, hipsim_f32x16 c, int, int, int) {
    hipsim::CollIn in{};
    for (int r = 0; r < 16; ++r) in.f[r] = c[r];
    std::memcpy(reinterpret_cast<char*>(&in.f[16]), &a, 16);
    std::memcpy(reinterpret_cast<char*>(&in.f[16]) + 16, &b, 16);
    hipsim::CollOut o = hipsim::wave_collective(in, hipsim::fn_mfma_bf16_32x32x16);
    hipsim_f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = o.f[r];
    return d;
}
static inline void __builtin_amdgcn_s_barrier() { hipsim::block_barrier(); }
// v_bfe_i32: the sign-extended bit field [offset, offset + width) of src
static inline int __builtin_amdgcn_sbfe(int src, unsigned offset, unsigned width) {
    offset &= 31u; width &= 31u;
    if (width == 0) return 0;
    const unsigned u = (unsigned)src >> offset;
    const unsigned m = width >= 32 ? 0xffffffffu : ((1u << width) - 1u);
    const unsigned f = u & m;
    return (f >> (width - 1)) & 1u ? (int)(f | ~m) : (int)f;
}
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)

// LDS-DMA: lane l copies `size` bytes from its own global address to (first lane's LDS pointer) + l*size
namespace hipsim {
static void fn_first_ptr(const CollIn* in, CollOut* out, int n) { for (int l = 0; l < n; ++l) out[l].u = in[0].u; }
}
static inline void __builtin_amdgcn_global_load_lds(const __attribute__((address_space(1))) void* src,
                                                    __attribute__((address_space(3))) void* dst, unsigned size, int offset,
                                                    unsigned) {
    hipsim::CollIn in{};
    in.u = (unsigned long long)(uintptr_t)dst;
    char* base = (char*)(uintptr_t)hipsim::wave_collective(in, hipsim::fn_first_ptr).u;
    std::memcpy(base + offset + (size_t)hipsim::cur->lane * size, (const char*)(uintptr_t)src + offset, size);
}

// buffer resources: raw (stride 0) buffer loads with the hardware's per-dword range check (out of range -> 0)
struct hipsim_rsrc { const char* base; long long num_records; };
typedef hipsim_rsrc __amdgpu_buffer_rsrc_t;
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num_records, int) {
    return hipsim_rsrc{(const char*)p, (long long)num_records};
}
// buffer_load ... lds: lane l copies `size` bytes from base + voffset + soffset + offset (range-checked: zeros beyond the buffer) to
// (first lane's LDS pointer) + offset + l * size
static inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(__amdgpu_buffer_rsrc_t r, __attribute__((address_space(3))) void* dst,
                                                            unsigned size, int voffset, int soffset, int offset, int) {
    hipsim::CollIn in{};
    in.u = (unsigned long long)(uintptr_t)dst;
    char* base = (char*)(uintptr_t)hipsim::wave_collective(in, hipsim::fn_first_ptr).u;
    const long long off = (long long)(unsigned)voffset + soffset + offset;
    char* d = base + offset + (size_t)hipsim::cur->lane * size;
    for (unsigned b = 0; b < size; b += 4) {
        unsigned x = 0u;
        if (off + b + 4 <= r.num_records) std::memcpy(&x, r.base + off + b, 4);
        std::memcpy(d + b, &x, 4);
    }
}
typedef unsigned int hipsim_v2u __attribute__((vector_size(8)));
static inline hipsim_v2u __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    hipsim_v2u v = {0u, 0u};
    const long long off = (long long)(unsigned)voffset + soffset;
    for (int d = 0; d < 2; ++d)
        if (off + 4 * d + 4 <= r.num_records) { unsigned x; std::memcpy(&x, r.base + off + 4 * d, 4); v[d] = x; }
    return v;
}


static inline unsigned short __builtin_amdgcn_raw_buffer_load_b16(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    unsigned short x = 0;
    const long long off = (long long)(unsigned)voffset + soffset;
    if (off + 2 <= r.num_records) std::memcpy(&x, r.base + off, 2);
    return x;
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    unsigned x = 0u;
    const long long off = (long long)(unsigned)voffset + soffset;
    if (off + 4 <= r.num_records) std::memcpy(&x, r.base + off, 4);
    return x;
}

typedef unsigned int hipsim_v4u __attribute__((ext_vector_type(4)));
static inline hipsim_v4u __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    hipsim_v4u v = {0u, 0u, 0u, 0u};
    const long long off = (long long)(unsigned)voffset + soffset;
    for (int d = 0; d < 4; ++d)
        if (off + 4 * d + 4 <= r.num_records) { unsigned x; std::memcpy(&x, r.base + off + 4 * d, 4); v[d] = x; }
    return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b128(hipsim_v4u v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    const long long off = (long long)(unsigned)voffset + soffset;
    for (int d = 0; d < 4; ++d)
        if (off + 4 * d + 4 <= r.num_records) { unsigned x = v[d]; std::memcpy((char*)r.base + off + 4 * d, &x, 4); }
}
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }   // hardware identity registers: one CU on the host
// (work-items of a process run one at a time: plain read-modify-write)
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
// scheduling hints have no meaning on the host
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_sleep(n) do { if ((n) >= 32) usleep(50); } while (0)   // long sleeps = spin-waits on another process
#define __builtin_amdgcn_s_setprio(n) ((void)0)

// ---- math that hipcc provides as builtins -------------------------------------------------------
// Compile the emulated build with -ffp-contract=off so these stay separately rounded.
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fsqrt_rn(float a) { return std::sqrt(a); }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
static inline double __dsub_rn(double a, double b) { volatile double r = a - b; return r; }
