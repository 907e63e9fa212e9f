// Development probe (not part of the product): dw_tiles_kernel on the flagship backward shapes -- timing, per-phase cycle sums.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I morl-baselines_amd/csrc tools/probes/dw_probe.hip -o tools/probes/dw_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "dw_tiles.h"
using namespace morl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(DW2_THREADS, 2) void dw_prof_kernel(Dw2Args a) { dw_tiles_body<true>(a); }

int main() {
    const int rows = 16384, L = 5;
    const int dims[6] = {35, 256, 256, 256, 256, 18};
    const int ldh[5] = {36, 256, 256, 256, 256}, ldg[5] = {256, 256, 256, 256, 20};
    float *G[5], *H[5], *slabs;
    long long P = 0; long long offW[5], offB[5];
    for (int l = 0; l < L; ++l) { offW[l] = P; P += (long long)dims[l + 1] * dims[l]; offB[l] = P; P += dims[l + 1]; }
    CK(hipMalloc(&slabs, (size_t)64 * P * 4));
    std::vector<float> tmp((size_t)rows * 256);
    for (size_t e = 0; e < tmp.size(); ++e) tmp[e] = (float)((int)(e * 2654435761u % 2000) - 1000) / 1000.f;
    for (int l = 0; l < L; ++l) {
        CK(hipMalloc(&G[l], (size_t)rows * ldg[l] * 4)); CK(hipMalloc(&H[l], (size_t)rows * ldh[l] * 4));
        CK(hipMemcpy(G[l], tmp.data(), (size_t)rows * ldg[l] * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(H[l], tmp.data(), (size_t)rows * ldh[l] * 4, hipMemcpyHostToDevice));
    }
    static const int lay_bm[3] = {128, 128, 32}, lay_bn[3] = {128, 64, 128}, lay_cost[3] = {4, 2, 1};
    for (int target : {512, 256, 768, 1024}) {
        Dw2Args a{};
        a.n = L; a.rows = rows; a.slab_stride = P;
        double unit_tiles = 0;
        for (int l = 0; l < L; ++l) {
            Dw2Problem& q = a.p[l];
            q.G = G[l]; q.ldg = ldg[l]; q.H = H[l]; q.ldh = ldh[l];
            q.C = slabs + offW[l]; q.ldc = dims[l]; q.colsum = slabs + offB[l];
            q.M = dims[l + 1]; q.N = dims[l];
            q.layout = (q.M <= 32) ? 2 : (q.N <= 64) ? 1 : 0;
            q.tiles_m = (q.M + lay_bm[q.layout] - 1) / lay_bm[q.layout];
            q.tiles_n = (q.N + lay_bn[q.layout] - 1) / lay_bn[q.layout];
            q.gcols = q.ldg; q.hcols = q.ldh; q.c_vec2 = ((q.ldc & 1) == 0 && (offW[l] & 1) == 0 && (P & 1) == 0) ? 1 : 0;
            unit_tiles += (double)q.tiles_m * q.tiles_n * lay_cost[q.layout] / 4.0;
        }
        int base = (std::max(1, (int)std::ceil(unit_tiles * rows / (double)target)) + 31) / 32 * 32;
        int jobs = 0;
        for (int l = 0; l < L; ++l) {
            Dw2Problem& q = a.p[l];
            q.k_per_split = base * 4 / lay_cost[q.layout];
            q.splits = (rows + q.k_per_split - 1) / q.k_per_split;
            if (q.splits > 64) { printf("too many splits\n"); return 1; }
            q.job_start = jobs; jobs += q.splits * q.tiles_m * q.tiles_n;
        }
        a.jobs = jobs;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(dw_tiles_kernel, dim3(jobs), dim3(DW2_THREADS), 0, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 20);
        }
        printf("target %4d: jobs %4d base slice %4d  %7.1f us  %6.1f TFLOP/s (algorithmic 6.89 GF)\n", target, jobs, base, best * 1e3, 6.887e9 / (best * 1e-3) / 1e12);
        if (target == 512) {
            long long* prof; CK(hipMalloc(&prof, (size_t)jobs * 6 * 8)); CK(hipMemset(prof, 0, (size_t)jobs * 6 * 8));
            a.prof = prof;
            hipLaunchKernelGGL(dw_prof_kernel, dim3(jobs), dim3(DW2_THREADS), 0, 0, a);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(dw_prof_kernel, dim3(jobs), dim3(DW2_THREADS), 0, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("  instrumented launch: %.1f us\n", ms * 1e3); }
            std::vector<long long> hp((size_t)jobs * 6); CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
            for (int l = 0; l < L; ++l) {
                const Dw2Problem& q = a.p[l];
                const int nj = q.splits * q.tiles_m * q.tiles_n;
                double s[6] = {0, 0, 0, 0, 0, 0};
                for (int j = 0; j < nj; ++j) for (int k = 0; k < 6; ++k) s[k] += (double)hp[(size_t)(q.job_start + j) * 6 + k] / nj;
                const int chunks = (q.k_per_split + 31) / 32;
                printf("  problem %d layout %d: %d jobs, %d chunks; per chunk cycles: load-issue %.0f | mfma loop %.0f | wait+lds stores %.0f | barrier %.0f   (mfma ideal %d); whole job: prologue %.0f | loop %.0f | epilogue %.0f\n",
                       l, q.layout, nj, chunks, s[3] / chunks, s[0] / chunks, s[1] / chunks, s[2] / chunks, 16 * lay_cost[q.layout] * 64,
                       s[4], s[0] + s[1] + s[2] + s[3], s[5]);
            }
        }
    }
    return 0;
}
