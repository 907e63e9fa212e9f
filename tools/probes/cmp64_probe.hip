// Development probe (not part of the product), round 6: issue rate of v_cmp_le_f64 / v_cmp_eq_f64 -- the instruction the Pareto
// prune (pareto_kernels.h) is made of -- per SIMD at 1, 2 and 4 waves per SIMD, beside v_cmp_le_f32 for scale.  The rate at 4 waves
// per SIMD x 1 024 SIMDs x the clock is the peak bench_front.py prices the kernel's executed compares against.
//   hipcc --offload-arch=gfx950 -O3 cmp64_probe.hip -o cmp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void k(long long* out, double seed, int iters) {
    double a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    float fa = (float)a, fb = (float)b;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_cmp_le_f64 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
            if (KIND == 1) asm volatile("v_cmp_eq_f64 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
            if (KIND == 2) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(fa), "v"(fb) : "vcc");
            // (rotating scalar destinations: back-to-back compares of one wave do not queue behind each other's VCC write)
            if (KIND == 3) {
                if ((i & 3) == 0) asm volatile("v_cmp_le_f64 s[40:41], %0, %1" :: "v"(a), "v"(b) : "s40", "s41");
                if ((i & 3) == 1) asm volatile("v_cmp_le_f64 s[42:43], %0, %1" :: "v"(a), "v"(b) : "s42", "s43");
                if ((i & 3) == 2) asm volatile("v_cmp_le_f64 s[44:45], %0, %1" :: "v"(a), "v"(b) : "s44", "s45");
                if ((i & 3) == 3) asm volatile("v_cmp_le_f64 s[46:47], %0, %1" :: "v"(a), "v"(b) : "s46", "s47");
            }
        }
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int KIND> int one(const char* name, long long* d) {
    for (int wps : {1, 2, 4}) {
        long long h[16] = {0};
        hipLaunchKernelGGL((k<KIND>), dim3(1), dim3(256 * wps), 0, 0, d, 1.5, 2000);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        const double per_wave = (double)h[0] / 32000.0;
        printf("%-28s %d wave(s)/SIMD: %.2f cycles per instruction per wave = %.2f cycles per instruction per SIMD = %.1f lane-compares/clk/SIMD\n",
               name, wps, per_wave, per_wave / wps, 64.0 * wps / per_wave);
        fflush(stdout);
    }
    return 0;
}
int main() {
    long long* d; CK(hipMalloc(&d, 64 * 8));
    one<0>("v_cmp_le_f64", d); one<1>("v_cmp_eq_f64", d); one<2>("v_cmp_le_f32", d); one<3>("v_cmp_le_f64, 4 destinations", d);
    return 0;
}
