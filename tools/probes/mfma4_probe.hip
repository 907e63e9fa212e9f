// Development micro-benchmark (not part of the product): issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 with 1 / 2 / 4
// independent accumulators, with and without the block broadcast (cbsz / abid), one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma4_probe.hip -o mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC, int BC>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
    f32x4 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    float a0 = 0.001f * threadIdx.x, b0 = 0.002f * (threadIdx.x % 17);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (BC) acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc[u % NACC], 4, 5, 0);
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc[u % NACC], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int BC>
int run(float* d, long long* c) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<NACC, BC>), dim3(256), dim3(256), 0, 0, d, c, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<NACC, BC>), dim3(256), dim3(256), 0, 0, d, c, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    long long cy = 0; CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
    const double n = 16.0 * iters;
    printf("accumulators %d broadcast %d: %.2f ns per MFMA per wave (%.1f shader-clock ticks of s_memtime per instruction), %.1f TFLOP/s chip\n",
           NACC, BC, ms * 1e6 / n, (double)cy / n, 256.0 * 4 * n * 512 / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    float* d; long long* c;
    CK(hipMalloc(&d, 256 * 256 * 4)); CK(hipMalloc(&c, 8));
    if (run<1, 0>(d, c) || run<2, 0>(d, c) || run<4, 0>(d, c) || run<1, 1>(d, c) || run<2, 1>(d, c) || run<4, 1>(d, c)) return 1;
    return 0;
}
