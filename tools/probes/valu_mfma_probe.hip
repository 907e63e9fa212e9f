// Development probe (not part of the product): vector-ALU issue rate of a wave whose SIMD also carries a wave of back-to-back MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/valu_mfma_probe.hip -o tools/probes/valu_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// waves [0, 4): MFMA (if mfma_waves), waves [4, 4 + 4 * valu_per_simd): 16 independent VALU streams; all loop `iters` times
template <int MIX>
__global__ void k(long long* out, float seed, int iters, int mfma_on) {
    const int wave = threadIdx.x >> 6;
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed + i + threadIdx.x;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) {
        if (mfma_on == 2) {
            // round 6: the same pipe time as 32 x 32 x 16 instructions (half as many, twice as long each)
            f32x16 big[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { big[0][i] = 0.f; big[1][i] = 0.f; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i & 1], 0, 0, 0);
            }
            acc[0][0] += big[0][0] + big[1][1];
        } else if (mfma_on)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
            }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MIX == 0) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 8) & 15]));
                if (MIX == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 8) & 15]));
                if (MIX == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(float2*)&r[i & 14]) : "v"(*(float2*)&r[(i + 8) & 14]));
                if (MIX == 4) { if (i & 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[i])); else asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(r[i])); }
                if (MIX == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(float2*)&r[i & 14]) : "v"(*(float2*)&r[(i + 8) & 14]));
                if (MIX == 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(r[(i + 8) & 15]));
                if (MIX == 7) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(r[(i + 8) & 15]));
                if (MIX == 1) { if (i & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 8) & 15]));
                                else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(float2*)&r[i & 14]) : "v"(*(float2*)&r[(i + 8) & 14])); }
            }
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int i = 0; i < 16; ++i) s += r[i];
    s += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (s == 12345.678f) out[63] = 1;
}
template <int MIX> int one(const char* name, long long* d) {
    for (int mfma_on = 0; mfma_on < 2; ++mfma_on) {
        long long h[16] = {0};
        hipLaunchKernelGGL((k<MIX>), dim3(1), dim3(512), 0, 0, d, 1.5f, 1000, mfma_on);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-22s beside an MFMA wave %s: %.2f cycles/MFMA, %.2f cycles per VALU instruction\n", name, mfma_on ? "ON " : "off", (double)h[0] / 16000.0, (double)h[4] / 16000.0);
    }
    return 0;
}
int main() {
    long long* d; CK(hipMalloc(&d, 64 * 8));
    one<0>("v_sub_f32", d); one<2>("v_cvt_pk_bf16_f32", d); one<3>("v_pk_add_f32", d); one<4>("v_and / v_lshlrev", d); one<5>("v_pk_mul_f32", d); one<6>("v_fma_f32", d); one<7>("v_perm_b32", d);
    for (int vw = 1; vw <= 3; ++vw) {
        long long h[16] = {0};
        hipLaunchKernelGGL((k<0>), dim3(1), dim3(256 + 256 * vw), 0, 0, d, 1.5f, 1000, 2);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("32x32x16 mfma wave ON, %d VALU wave(s) per SIMD: mfma wave %.2f cycles per 32x32x16 MFMA, VALU waves %.2f cycles/instruction each (v_sub_f32)\n",
               vw, (double)h[0] / 8000.0, (double)h[4] / 16000.0);
    }
    for (int mfma_on = 0; mfma_on < 2; ++mfma_on)
        for (int vw = 1; vw <= 3; ++vw) {
            long long h[16] = {0};
            hipLaunchKernelGGL((k<0>), dim3(1), dim3(256 + 256 * vw), 0, 0, d, 1.5f, 1000, mfma_on);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            printf("mfma wave %s, %d VALU wave(s) per SIMD: mfma wave %.2f cycles/MFMA, VALU waves %.2f cycles/instruction each (v_sub_f32)\n",
                   mfma_on ? "ON " : "off", vw, (double)h[0] / 16000.0, (double)h[4] / 16000.0);
            hipLaunchKernelGGL((k<1>), dim3(1), dim3(256 + 256 * vw), 0, 0, d, 1.5f, 1000, mfma_on);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            printf("                                         mfma wave %.2f cycles/MFMA, VALU waves %.2f cycles/instruction each (cvt_pk / pk_add mix)\n",
                   (double)h[0] / 16000.0, (double)h[4] / 16000.0);
        }
    return 0;
}
