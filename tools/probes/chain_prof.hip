// Development probe (not part of the product): phase breakdown of mlp_chain_kernel on the flagship forward chain.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I morl-baselines_amd/csrc tools/probes/chain_prof.hip -o tools/probes/chain_prof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mlp_chain.h"
using namespace morl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(CH_THREADS) void chain_prof_kernel(ChainArgs p) { mlp_chain_body<true, false>(p); }
__global__ __launch_bounds__(CH_THREADS) void chain_prof_dma_kernel(ChainArgs p) { mlp_chain_body<true, true>(p); }

int main() {
    const int B = 256, W = 64, D = 32, R = 3, A = 6, rows = B * W;
    const int dims[6] = {35, 256, 256, 256, 256, 18};
    float *obs, *wv, *wt, *bias, *q, *hbuf; long long* prof;
    CK(hipMalloc(&obs, B * D * 4)); CK(hipMalloc(&wv, W * R * 4)); CK(hipMalloc(&wt, 4 * 256 * 256 * 4 + 256 * 20 * 4));
    CK(hipMalloc(&bias, 5 * 256 * 4)); CK(hipMalloc(&q, (size_t)rows * 20 * 4)); CK(hipMalloc(&hbuf, (size_t)4 * rows * 256 * 4));
    CK(hipMalloc(&prof, 256 * 8 * 8));
    CK(hipMemset(obs, 0, B * D * 4)); CK(hipMemset(wv, 0, W * R * 4)); CK(hipMemset(wt, 0, 4 * 256 * 256 * 4 + 256 * 20 * 4)); CK(hipMemset(bias, 0, 5 * 256 * 4));
    for (int save = 0; save < 2; ++save) {
        ChainArgs a{};
        a.n_steps = 5; a.rows = rows; a.in_mode = 0; a.obs = obs; a.weights = wv; a.B = B; a.W = W; a.D = D; a.R = R; a.row_order = 0;
        a.prof = prof;
        size_t off = 0;
        for (int l = 0; l < 5; ++l) {
            ChainStep& st = a.step[l];
            st.Bmat = wt + off; st.ldb = (dims[l + 1] + 3) / 4 * 4; st.K = dims[l]; st.N = dims[l + 1]; st.bias = bias + l * 256; st.relu = l < 4;
            off += (size_t)dims[l] * st.ldb;
            if (l == 4) { st.out = q; st.ldout = 20; }
            else if (save) { st.out = hbuf + (size_t)l * rows * 256; st.ldout = 256; }
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int dma = 0; dma < 2; ++dma) {
        if (dma) hipLaunchKernelGGL(chain_prof_dma_kernel, dim3(rows / CH_TM), dim3(CH_THREADS), 0, 0, a);
        else hipLaunchKernelGGL(chain_prof_kernel, dim3(rows / CH_TM), dim3(CH_THREADS), 0, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(256 * 8);
        CK(hipMemcpy(h.data(), prof, 256 * 8 * 8, hipMemcpyDeviceToHost));
        double s[4] = {0, 0, 0, 0};
        for (int b = 0; b < 256; ++b) for (int k = 0; k < 4; ++k) s[k] += h[b * 8 + k] / 256.0;
        printf("save=%d dma=%d  phase cycles (avg over blocks, thread 0): input %.0f  mfma-loop %.0f  stage(load issue+store+barrier) %.0f  epilogue %.0f  total %.0f\n",
               save, dma, s[0], s[1], s[2], s[3], s[0] + s[1] + s[2] + s[3]);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) {
            if (dma) hipLaunchKernelGGL(mlp_chain_dma_kernel, dim3(rows / CH_TM), dim3(CH_THREADS), 0, 0, a);
            else hipLaunchKernelGGL(mlp_chain_kernel, dim3(rows / CH_TM), dim3(CH_THREADS), 0, 0, a);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("save=%d dma=%d  plain kernel %.1f us/launch  %.1f TFLOP/s\n", save, dma, ms / 20 * 1e3, (double)rows * 420352.0 / (ms / 20 * 1e-3) / 1e12);
      }
    }
    return 0;
}
