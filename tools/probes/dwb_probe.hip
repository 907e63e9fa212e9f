// Development probe (not part of the product): dw_bf_kernel on the flagship backward shapes -- timing with / without the extra
// (PER) workgroup, per-phase cycle sums of one consumer and one producer wave per job, job start / end skew.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I morl-baselines_amd/csrc tools/probes/dwb_probe.hip -o tools/probes/dwb_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "dw_bf.h"
using namespace morl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(DWB_THREADS) void dw_bf_prof_kernel(DwbArgs a) { dw_bf_body<true>(a); }

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
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int target : {cus, cus - 1, 2 * cus}) {
        DwbArgs a{};
        a.n = L; a.rows = rows; a.slab_stride = P;
        double cost_rows = 0;
        for (int l = 0; l < L; ++l) {
            DwbProblem& q = a.p[l];
            q.G = G[l]; q.ldg = ldg[l]; q.H = H[l]; q.ldh = ldh[l];
            q.C = slabs + offW[l]; q.ldc = dims[l]; q.colsum = slabs + offB[l];
            q.M = dims[l + 1]; q.N = dims[l]; q.gcols = q.ldg; q.hcols = q.ldh;
            q.shape = (q.M <= 32) ? 2 : (q.N <= 64) ? 1 : 0;
            q.mgroups = (q.M + 16 * DWB_TG[q.shape] - 1) / (16 * DWB_TG[q.shape]);
            q.ngroups = (q.N + 16 * DWB_TH[q.shape] - 1) / (16 * DWB_TH[q.shape]);
            cost_rows += (double)q.mgroups * q.ngroups * rows;
        }
        const int base = (std::max(1, (int)std::ceil(cost_rows / target)) + 31) / 32 * 32;
        int jobs = 0;
        for (int l = 0; l < L; ++l) {
            DwbProblem& q = a.p[l];
            q.k_per_split = base; q.splits = (rows + base - 1) / base;
            if (q.splits > 64) { printf("too many splits\n"); return 1; }
            q.job_start = jobs; jobs += q.splits * q.mgroups * q.ngroups;
        }
        a.jobs = jobs;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int extra = 0; extra < 2; ++extra) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(dw_bf_kernel, dim3(jobs + extra), dim3(DWB_THREADS), 0, 0, a);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 20);
            }
            printf("target %4d: jobs %4d (+%d idle block) slice %4d rows (%d chunks)  %7.1f us\n", target, jobs, extra, base, base / 32, best * 1e3);
        }
        if (target == cus || target == cus - 1) {
            long long* prof; CK(hipMalloc(&prof, (size_t)jobs * DWB_PROF_SLOTS * 8)); CK(hipMemset(prof, 0, (size_t)jobs * DWB_PROF_SLOTS * 8));
            a.prof = prof;
            hipLaunchKernelGGL(dw_bf_prof_kernel, dim3(jobs), dim3(DWB_THREADS), 0, 0, a);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(dw_bf_prof_kernel, dim3(jobs), dim3(DWB_THREADS), 0, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("  instrumented launch: %.1f us\n", ms * 1e3); }
            std::vector<long long> hp((size_t)jobs * DWB_PROF_SLOTS); CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
            long long w0 = hp[48], w1 = 0;
            for (int j = 0; j < jobs; ++j) { w0 = std::min(w0, hp[(size_t)j * DWB_PROF_SLOTS + 48]); w1 = std::max(w1, hp[(size_t)j * DWB_PROF_SLOTS + 49]); }
            printf("  first job start -> last job end: %.1f us (100 MHz wall clock)\n", (double)(w1 - w0) / 100.0);
            for (int l = 0; l < L; ++l) {
                const DwbProblem& q = a.p[l];
                const int nj = q.splits * q.mgroups * q.ngroups;
                double s[DWB_PROF_SLOTS] = {0};
                double en_min = 1e18, en_max = 0;
                for (int j = 0; j < nj; ++j) {
                    const long long* r = &hp[(size_t)(q.job_start + j) * DWB_PROF_SLOTS];
                    for (int k = 0; k < DWB_PROF_SLOTS; ++k) s[k] += (double)r[k] / nj;
                    en_min = std::min(en_min, (double)(r[49] - w0)); en_max = std::max(en_max, (double)(r[49] - w0));
                }
                const int chunks = q.k_per_split / 32;
                printf("  problem %d shape %d: %d jobs x %d chunks; end %.1f..%.1f us; per wave: prologue | per chunk work, barrier wait | epilogue (cycles)\n",
                       l, q.shape, nj, chunks, en_min / 100, en_max / 100);
                for (int w = 0; w < 12; ++w)
                    printf("    wave %2d (%s): %6.0f | %5.0f %5.0f | %6.0f\n", w, w < 4 ? "consumer" : "producer", s[4 * w], s[4 * w + 1] / chunks,
                           s[4 * w + 2] / chunks, s[4 * w + 3]);
            }
            a.prof = nullptr;
            CK(hipFree(prof));
        }
    }
    return 0;
}
