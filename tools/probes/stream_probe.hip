// Development micro-benchmark (not part of the product): how fast ONE workgroup (256 threads, one CU) streams a 1 MB weight set
// L2 -> registers with buffer_load_dwordx4 at a given number of loads in flight per lane, first touch inside a launch against a
// re-read inside the same launch, alone on the chip and beside 255 other workgroups; and the same stream straight into LDS.
//   hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// bytes: per pass; every lane loads 16 B per instruction, a wave 1 KB contiguous, the four waves interleave 1 KB blocks
template <int DEPTH>
__global__ __launch_bounds__(256) void stream(const float* src, int bytes, int passes, float* out, long long* ticks) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const int tid = threadIdx.x;
    const int n = bytes / (256 * 16);          // loads per lane per pass
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < passes; ++p) {
        const long long t0 = wall_clock64();
        float4 ring[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) ring[d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, d * 4096, 0));
        for (int i = 0; i < n; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const float4 v = ring[d];
                ring[d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (i + DEPTH + d) * 4096, 0));   // (beyond the buffer: zeros)
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        __syncthreads();
        const long long t1 = wall_clock64();
        if (tid == 0) ticks[blockIdx.x * 8 + p] = t1 - t0;
    }
    out[blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int DEPTH>
int run(const float* src, int bytes, int grid, float* out, long long* ticks, const char* what) {
    CK(hipMemset(ticks, 0, 256 * 8 * 8));
    hipLaunchKernelGGL((stream<DEPTH>), dim3(grid), dim3(256), 0, 0, src, bytes, 3, out, ticks);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((stream<DEPTH>), dim3(grid), dim3(256), 0, 0, src, bytes, 3, out, ticks);
    CK(hipDeviceSynchronize());
    long long h[256 * 8];
    CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
    double p[3] = {0, 0, 0};
    for (int b = 0; b < grid; ++b) for (int k = 0; k < 3; ++k) p[k] += (double)h[b * 8 + k] / grid;
    printf("%-28s depth %2d grid %3d: pass 1 %.2f us (%.1f GB/s per CU), pass 2 %.2f us (%.1f GB/s), pass 3 %.2f us\n", what, DEPTH, grid,
           p[0] * 0.01, bytes / (p[0] * 10.0), p[1] * 0.01, bytes / (p[1] * 10.0), p[2] * 0.01);
    return 0;
}

int main() {
    const int bytes = 1 << 20;
    float *src, *out, *big; long long* ticks;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&ticks, 256 * 8 * 8));
    CK(hipMalloc(&big, 512 << 20));
    CK(hipMemset(src, 0, bytes));
    for (int grid : {1, 32, 256}) {
        // (a 512 MB fill in front of each launch: the weight set is in neither L2 nor the memory-side cache's hot part)
        CK(hipMemset(big, 1, 512 << 20)); CK(hipDeviceSynchronize());
        if (run<4>(src, bytes, grid, out, ticks, "after a 512 MB fill") || run<8>(src, bytes, grid, out, ticks, "back to back")
            || run<16>(src, bytes, grid, out, ticks, "back to back") || run<24>(src, bytes, grid, out, ticks, "back to back")
            || run<48>(src, bytes, grid, out, ticks, "back to back")) return 1;
    }
    return 0;
}
