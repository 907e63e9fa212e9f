// Development probe (not part of the product): does a wave's OWN vector-ALU work run under its MFMAs?  One wave per SIMD issues
// v_mfma_f32_16x16x32_bf16 rotating over NACC accumulators with NVALU independent vector instructions behind each MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_same_wave_probe.hip -o tools/probes/mfma_same_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int NVALU>
__global__ void k(long long* out, float seed, int iters) {
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, 0, 0, 0};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    float v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = seed * (i + 1) + threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % NACC], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NVALU; ++u) {
                const int r = (i * NVALU + u) % 12;      // twelve independent chains: a chain's next instruction is >= 12 instructions away
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(seed));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}
template <int NACC, int NVALU> int run(long long* d) {
    long long h = 0;
    hipLaunchKernelGGL((k<NACC, NVALU>), dim3(1), dim3(256), 0, 0, d, 1.5f, 500);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("%d accumulator(s), %2d vector instruction(s) per MFMA, one wave per SIMD: %6.2f cycles per MFMA\n", NACC, NVALU, (double)h / (24.0 * 500));
    fflush(stdout);
    return 0;
}
int main() {
    long long* d; CK(hipMalloc(&d, 64));
    run<1, 0>(d); run<2, 0>(d); run<3, 0>(d); run<4, 0>(d);
    run<2, 1>(d); run<2, 2>(d); run<2, 3>(d); run<2, 4>(d); run<2, 6>(d); run<2, 8>(d);
    run<4, 1>(d); run<4, 2>(d); run<4, 3>(d); run<4, 4>(d); run<4, 6>(d); run<4, 8>(d);
    return 0;
}
