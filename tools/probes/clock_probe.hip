// Development probe: the shader clock the chip sustains under dense fp32 MFMA load (the DVFS give-back of MI355X_MICROARCH.md).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/clock_probe.hip -o tools/probes/clock_probe
// Each workgroup (4 waves) issues `iters` x 16 v_mfma_f32_32x32x2_f32 on four accumulators, operands from registers (random or
// zero data); wave 0 stamps s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) before and after.
// clock = d(s_memtime) / d(s_memrealtime) * 100 MHz; the HIP-event duration cross-checks the 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void burn(const float* __restrict__ data, float* __restrict__ out, long long* __restrict__ stamps, int iters) {
    const int tid = threadIdx.x;
    float a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = data[(blockIdx.x * 256 + tid) * 8 + k]; b[k] = data[(blockIdx.x * 256 + tid) * 8 + 4 + k]; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 1) & 3], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 3], b[k], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 2) & 3], b[(k + 3) & 3], acc[3], 0, 0, 0);
        }
    }
    if (tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        stamps[blockIdx.x * 4 + 2] = r0;
        stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    const int max_blocks = 1024;
    std::vector<float> h((size_t)max_blocks * 256 * 8);
    float *d_rand, *d_zero, *out; long long* stamps;
    CK(hipMalloc(&d_rand, h.size() * 4)); CK(hipMalloc(&d_zero, h.size() * 4)); CK(hipMalloc(&out, (size_t)max_blocks * 256 * 4));
    CK(hipMalloc(&stamps, (size_t)max_blocks * 4 * 8));
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(d_rand, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_zero, 0, h.size() * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int blocks; bool zero; int iters; } cases[] = {
        {"1 workgroup, random data", 1, false, 40000}, {"512 workgroups (2 per CU), random data", 512, false, 40000},
        {"512 workgroups, zero data", 512, true, 40000}, {"256 workgroups (1 per CU), random data", 256, false, 40000},
        {"512 workgroups, random data, 10x longer", 512, false, 400000}};
    for (const Case& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {     // second run reported (clocks settled)
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(burn, dim3(c.blocks), dim3(256), 0, 0, c.zero ? d_zero : d_rand, out, stamps, c.iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> st((size_t)c.blocks * 4);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double clk = 0, dur = 0;
        for (int b = 0; b < c.blocks; ++b) {
            const double dc = (double)(st[b * 4 + 1] - st[b * 4 + 0]), dr = (double)(st[b * 4 + 3] - st[b * 4 + 2]);
            clk += dc / dr * 0.1;      // GHz if s_memrealtime ticks at 100 MHz
            dur += dr * 10e-9;
        }
        clk /= c.blocks; dur /= c.blocks;
        const double flop = (double)c.blocks * 4 * c.iters * 16 * 4096.0;
        printf("%-48s event %.3f ms, in-kernel %.3f ms (100 MHz assumed), shader clock %.3f GHz, %.1f TFLOP/s, MFMA cycles/instr/SIMD %.1f\n", c.name, ms, dur * 1e3, clk,
               flop / (ms * 1e-3) / 1e12, clk * 1e9 * dur / ((double)c.iters * 16 * (c.blocks > 256 ? 2 : 1)));
    }
    return 0;
}
