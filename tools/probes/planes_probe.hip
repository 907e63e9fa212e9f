// Development probe (not part of the product), round 6: the two facts the weight-gradient launch on pre-split operands rests on.
//   1. ds_read_b64_tr_b16: which lane receives which 16-bit element (dumped for a linear LDS image and for the granule image of
//      dw_planes.h), and whether the image's reads are free of bank conflicts (cycles per read against a plain ds_read_b64).
//   2. how fast the chip reads N MB that a kernel wrote just before (the bf16 planes the chains write, read back by the weight
//      gradients): straight after the write, with another 110 MB written in between, and after a 1 GB flush -- HBM against the
//      256 MB memory-side cache.
//   hipcc --offload-arch=gfx950 -O3 planes_probe.hip -o planes_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void tr_dump(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int addr_shorts = lane * 4;                   // mode 0: lane l reads the 8 bytes at 8 l
    if (mode == 1) addr_shorts = (lane & 15) * 4 + (lane >> 4) * 256;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr_shorts));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)r[e];
}

// cycles per transposing read for three address patterns: 0 = every lane its own 8 bytes, linear (conflict-free by construction:
// 512 contiguous bytes), 1 = the granule image (lane (i, q'): granule G(i, q'), half i & 1), 2 = lower halves of 64 granules (the
// pattern a 16-byte-granule image with one feature tile per half is forced into: 2-way)
__global__ void tr_time(long long* ticks, int* sink, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, qp = lane >> 4;
    int byte_addr;
    if (pattern == 0) byte_addr = lane * 8;
    else if (pattern == 1) {
        const int t = i >> 2, c = i & 3, qsel = c >> 1, h = c & 1;
        const int G = t | ((qp & 1) << 2) | (qsel << 3);              // + 1 KB per qp >> 1 (bank-neutral)
        byte_addr = (qp >> 1) * 1024 + G * 16 + h * 8;
    } else byte_addr = lane * 16;
    v4s acc = {0, 0, 0, 0};
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((char*)lds + byte_addr + u * 2048));
            acc += r;
        }
    }
    const long long t1 = clock64();
    if (lane == 0) ticks[pattern] = t1 - t0;
    sink[lane] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void fill(u32x4* dst, long long n16, unsigned seed) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = u32x4{seed, (unsigned)i, seed, 1u};
}
__global__ __launch_bounds__(256) void drain(const u32x4* src, long long n16, unsigned* out) {
    const long long stride = (long long)gridDim.x * 256;
    u32x4 a = {0, 0, 0, 0};
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) v[d] = src[i + d * stride];
#pragma unroll
        for (int d = 0; d < 8; ++d) a ^= v[d];
    }
    for (; i < n16; i += stride) a ^= src[i];
    out[blockIdx.x * 256 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
}

int main() {
    unsigned short* d; CK(hipMalloc(&d, 64 * 4 * 2));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(tr_dump, dim3(1), dim3(64), 0, 0, d, mode);
        unsigned short h[256]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("tr16_b64 dump, mode %d (LDS short index returned per lane, 4 elements):\n", mode);
        for (int l = 0; l < 64; ++l) { printf(" l%02d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : ""); }
    }
    long long* ticks; int* sink; CK(hipMalloc(&ticks, 64)); CK(hipMalloc(&sink, 256));
    for (int rep = 0; rep < 2; ++rep)
        for (int p = 0; p < 3; ++p) hipLaunchKernelGGL(tr_time, dim3(1), dim3(64), 0, 0, ticks, sink, p);
    long long ht[3]; CK(hipMemcpy(ht, ticks, sizeof(ht), hipMemcpyDeviceToHost));
    for (int p = 0; p < 3; ++p) printf("tr read pattern %d: %.2f cycles per read (one wave, 4096 reads)\n", p, ht[p] / 4096.0);

    // ---- write -> read bandwidth ----
    const long long MB = 1 << 20;
    u32x4 *buf, *other, *flush; unsigned* out;
    CK(hipMalloc(&buf, 512 * MB)); CK(hipMalloc(&other, 128 * MB)); CK(hipMalloc(&flush, 1024 * MB)); CK(hipMalloc(&out, 2048 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 2048;
    for (long long mb : {32, 64, 106, 160, 211, 256, 384, 512}) {
        const long long n16 = mb * MB / 16;
        for (int scen = 0; scen < 3; ++scen) {
            float best = 1e9f, sum = 0.f; const int reps = 5;
            for (int r = 0; r < reps; ++r) {
                if (scen == 2) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, flush, 1024 * MB / 16, 7u);
                hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, buf, n16, (unsigned)r);
                if (scen == 1) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, other, 110 * MB / 16, 3u);
                if (scen == 2) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, flush, 1024 * MB / 16, 9u);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(drain, dim3(grid), dim3(256), 0, 0, buf, n16, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best; sum += ms;
            }
            const char* what[3] = {"read right after its write", "110 MB written in between", "after a 1 GB flush"};
            printf("read %4lld MB, %-26s: best %.1f us = %.2f TB/s, mean %.1f us\n", mb, what[scen], best * 1e3, mb * MB / (best * 1e-3) / 1e12,
                   sum / reps * 1e3);
        }
    }
    // and the write side alone
    for (long long mb : {106, 211}) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, buf, mb * MB / 16, (unsigned)r);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("write %4lld MB: best %.1f us = %.2f TB/s\n", mb, best * 1e3, mb * MB / (best * 1e-3) / 1e12);
    }
    return 0;
}
