// Development probe (not part of the product): where does the time of one latency-bound hidden layer
// (G nets x M rows x 256 x 256, gemm_wave4_batched_kernel) go?  Variants drop one phase at a time.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../morl-baselines_amd/csrc -I../../include wave_gemm_probe.hip -o wave_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ac_kernels.h"
using namespace morl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// V: 1 no global loads, 2 no MFMA, 3 nothing (launch floor), 4 no split-K reduction / single store per wave,
//    5 direct (non-LDS) operand loads, 6 loads only (no MFMA, no reduce)
template <int V>
__global__ __launch_bounds__(256) void probe_kernel(GemmBatched b) {
    __shared__ float s_red[3][16][64];
    __shared__ __attribute__((aligned(16))) float s_panel[4][32 * WG4_PITCH];
    GemmProblem g = b.p;
    const int z = (int)blockIdx.z, tile = (int)blockIdx.x;
    g.A += (long long)(z / b.a_div) * b.sA; g.B += (long long)z * b.sB; g.C += (long long)z * b.sC; g.bias += (long long)z * b.sBias;
    if (V == 3) { if (g.M < 0) g.C[0] = 1.f; return; }
    const int tile_m = tile / g.tiles_n, tile_n = tile % g.tiles_n;
    const int lane = lane_id(), wave = wave_id(), h = lane >> 5, i = lane & 31;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int kper = ((g.K + 3) / 4 + 3) / 4 * 4, kbeg = wave * kper, kend = min(g.K, kbeg + kper);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += WG4_CHUNK) {
        float a[32], bb[32];
        if (V == 1) {
            for (int s = 0; s < 32; ++s) { a[s] = 0.001f * (lane + s); bb[s] = 0.002f * (lane - s); }
        } else if (V == 7) {
            float4 ta[8];
            panel_fetch(ta, g.A, g.lda, m0, g.M, k0, kend, lane);
            wave_load64<false>(bb, g.B, g.ldb, n0 + i, g.N, k0, kend, h, 1);     // B read as [k][n]: coalesced along n
            panel_transpose(a, ta, s_panel[wave], lane);
        } else if (V == 5) {
            wave_load64<true>(a, g.A, g.lda, m0 + i, g.M, k0, kend, h, 1);
            wave_load64<true>(bb, g.B, g.ldb, n0 + i, g.N, k0, kend, h, 1);
        } else {
            float4 ta[8], tb[8];
            panel_fetch(ta, g.A, g.lda, m0, g.M, k0, kend, lane);
            panel_fetch(tb, g.B, g.ldb, n0, g.N, k0, kend, lane);
            panel_transpose(a, ta, s_panel[wave], lane);
            panel_transpose(bb, tb, s_panel[wave], lane);
        }
        if (V == 2 || V == 6) { for (int s = 0; s < 32; ++s) acc[s & 15] += a[s] * bb[s]; }
        else for (int s = 0; s < 32; ++s) acc = mfma32(a[s], bb[s], acc);
    }
    if (V != 4 && V != 6) {
        if (wave > 0) for (int r = 0; r < 16; ++r) s_red[wave - 1][r][lane] = acc[r];
        __syncthreads();
        if (wave != 0) return;
        for (int w = 0; w < 3; ++w) for (int r = 0; r < 16; ++r) acc[r] += s_red[w][r][lane];
    } else if (wave != 0) {
        float t = 0.f; for (int r = 0; r < 16; ++r) t += acc[r];
        if (t == 12345.678f) g.C[0] = t;
        return;
    }
    const int col = n0 + i;
    const float bias = g.bias[col];
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        g.C[(size_t)row * g.ldc + col] = fmaxf(acc[r] + bias, 0.f);
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 2, M = argc > 2 ? atoi(argv[2]) : 128, N = 256, K = 256, iters = 500;
    float *X, *Y, *W, *Bv;
    CK(hipMalloc(&X, (size_t)G * M * K * 4)); CK(hipMalloc(&Y, (size_t)G * M * N * 4));
    CK(hipMalloc(&W, (size_t)G * N * K * 4)); CK(hipMalloc(&Bv, (size_t)G * N * 4));
    std::vector<float> hx((size_t)G * M * K, 0.01f), hw((size_t)G * N * K, 0.003f), hb((size_t)G * N, 0.1f);
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Bv, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Y, hx.data(), (size_t)G * M * N * 4, hipMemcpyHostToDevice));
    GemmBatched b{};
    b.p.lda = K; b.p.ldb = K; b.p.ldc = N; b.p.M = M; b.p.N = N; b.p.K = K; b.p.a_vec = b.p.b_vec = 1;
    b.p.tiles_m = (M + 31) / 32; b.p.tiles_n = (N + 31) / 32; b.p.B = W; b.p.bias = Bv;
    b.sA = (long long)M * K; b.sB = (long long)N * K; b.sC = (long long)M * N; b.sBias = N; b.a_div = 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(b.p.tiles_m * b.p.tiles_n, 1, G);
    const char* names[] = {"product kernel", "no global loads", "no MFMA (fma instead)", "empty kernel", "no split-K reduce",
                           "direct row-walk loads", "loads + LDS transpose only", "A panel + B n-contiguous"};
    for (int v = 0; v <= 7; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < iters; ++it) {
                b.p.A = (it & 1) ? Y : X; b.p.C = (it & 1) ? X : Y;           // dependent chain, like consecutive layers
                switch (v) {
                    case 0: hipLaunchKernelGGL((gemm_wave4_batched_kernel<true, true, EPI_BIAS_RELU>), grid, dim3(256), 0, 0, b); break;
                    case 1: hipLaunchKernelGGL(probe_kernel<1>, grid, dim3(256), 0, 0, b); break;
                    case 2: hipLaunchKernelGGL(probe_kernel<2>, grid, dim3(256), 0, 0, b); break;
                    case 3: hipLaunchKernelGGL(probe_kernel<3>, grid, dim3(256), 0, 0, b); break;
                    case 4: hipLaunchKernelGGL(probe_kernel<4>, grid, dim3(256), 0, 0, b); break;
                    case 5: hipLaunchKernelGGL(probe_kernel<5>, grid, dim3(256), 0, 0, b); break;
                    case 6: hipLaunchKernelGGL(probe_kernel<6>, grid, dim3(256), 0, 0, b); break;
                    case 7: hipLaunchKernelGGL(probe_kernel<7>, grid, dim3(256), 0, 0, b); break;
                }
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("G=%d M=%d  %-28s %7.2f us / launch\n", G, M, names[v], ms * 1e3 / iters);
        }
    }
    return 0;
}
