// Development probe (not part of the product): issue rate of v_mfma_f32_16x16x32_bf16 against the number of accumulators the
// stream alternates between (1 = every MFMA depends on its predecessor), one wave per SIMD and two.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_dep_probe.hip -o tools/probes/mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void k(long long* out, float seed, int iters) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, 0, 0, 0};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}
template <int NACC> int run(long long* d) {
    for (int waves : {1, 2, 3, 4}) {
        long long h = 0;
        hipLaunchKernelGGL((k<NACC>), dim3(1), dim3(256 * waves), 0, 0, d, 1.5f, 500);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("%d accumulator(s), %d wave(s) per SIMD: %6.2f cycles per MFMA per wave", NACC, waves, (double)h / (24.0 * 500));
        // the whole chip: one such workgroup per CU, wall time
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(256 * waves), 0, 0, d, 1.5f, 2000);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(256 * waves), 0, 0, d, 1.5f, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 4 * waves * 24 * 2000 * 16384.0;
        printf("   chip-wide: %.3f ms, %.0f TFLOP/s\n", ms, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
int main() {
    long long* d; CK(hipMalloc(&d, 64));
    run<1>(d); run<4>(d);
    return 0;
}
