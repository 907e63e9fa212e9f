// Development micro-benchmark (not part of the product): what does the fp32 MFMA inner loop of the chain kernel
// sustain on gfx950 at one wave per SIMD?   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// variant 0: MFMA only (operands in registers)
// variant 1: operands re-read from LDS every k-pair exactly like mlp_chain_kernel (2 x ds_read2_b32 + 4 MFMA)
// variant 2: like 1 but the LDS reads of iteration n+1 are issued before the MFMAs of iteration n (software pipelined)
// variant 3: like 1, unrolled x4 by the compiler
template <int VARIANT>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int lds_bytes_dummy) {
    __shared__ float sA[256 * 65];
    __shared__ float sB[32 * 256 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, i = lane & 31;
    for (int e = tid; e < 256 * 65; e += 256) sA[e] = 0.001f * (e % 97);
    for (int e = tid; e < 32 * 256 * 2; e += 256) sB[e] = 0.002f * (e % 89);
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float* pa = sA + h * 65 + i;
    const float* pb = sB + h * 256 + wave * 64 + i;
    if (VARIANT == 0) {
        float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
        for (int it = 0; it < iters; ++it) {
#pragma unroll 1
            for (int kk = 0; kk < 32; kk += 2) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    } else if (VARIANT == 1 || VARIANT == 3) {
        for (int it = 0; it < iters; ++it) {
            const int k0 = (it & 7) * 32;
#pragma unroll(VARIANT == 3 ? 4 : 1)
            for (int kk = 0; kk < 32; kk += 2) {
                const float a0 = pa[(k0 + kk) * 65], a1 = pa[(k0 + kk) * 65 + 32];
                const float b0 = pb[kk * 256], b1 = pb[kk * 256 + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            const int k0 = (it & 7) * 32;
            float a0 = pa[k0 * 65], a1 = pa[k0 * 65 + 32], b0 = pb[0], b1 = pb[32];
#pragma unroll 1
            for (int kk = 0; kk < 32; kk += 2) {
                const int kn = (kk + 2) & 31;
                const float na0 = pa[(k0 + kn) * 65], na1 = pa[(k0 + kn) * 65 + 32];
                const float nb0 = pb[kn * 256], nb1 = pb[kn * 256 + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
int run(const char* name, int blocks, int iters, float* d) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d, iters, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d, iters, 0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    double flop = (double)blocks * 4 /*waves*/ * iters * 16 * 4 * 4096.0;
    printf("%-28s blocks=%4d iters=%5d  %.3f ms  %.1f TFLOP/s  (%.1f cycles per 4-MFMA group @2.4GHz)\n", name, blocks, iters, ms,
           flop / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 16.0));
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 2048 * 256 * 4));
    for (int blocks : {256, 512, 1024}) {
        run<0>("mfma only", blocks, 200, d);
        run<1>("lds reads, rolled", blocks, 200, d);
        run<3>("lds reads, unroll 4", blocks, 200, d);
        run<2>("lds reads, sw pipelined", blocks, 200, d);
    }
    return 0;
}
