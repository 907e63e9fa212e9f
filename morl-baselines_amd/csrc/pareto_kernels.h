// Non-dominated filtering (common/pareto.py:34-57, get_non_pareto_dominated_inds), float64 comparisons.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

// keep[i] = (#{j : c_i <= c_j in every objective} == #{j : c_j == c_i in every objective})
//           && (!remove_duplicates || no j < i with c_j == c_i)
// (the reference's second clause any(~res_g) is identically true because res_g[i][i] is false).
//
// All-pairs with LDS tiling: one work-item per candidate i, the j points stream through LDS in tiles of
// PARETO_TILE points and are read as broadcasts (every lane of a wave reads the same j).  A candidate
// that has met a weak dominator that is not its duplicate is settled; __ballot over the wave lets the
// whole wave skip the remaining tiles once all 64 of its candidates are settled.
// Integer/boolean result -> bit-exact.  Work N^2*R compares; bytes N*R*8 per workgroup (L2 hits).
constexpr int PARETO_TILE = 512;
constexpr int PARETO_THREADS = 256;

__global__ __launch_bounds__(PARETO_THREADS) void pareto_mask_kernel(const double* __restrict__ pts, int N, int R,
                                                                     int remove_duplicates,
                                                                     uint8_t* __restrict__ mask) {
    __shared__ double s_pts[PARETO_TILE * MORL_MAX_OBJ];
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool valid = i < N;
    double ci[MORL_MAX_OBJ];
#pragma unroll
    for (int r = 0; r < MORL_MAX_OBJ; ++r) ci[r] = (valid && r < R) ? pts[(size_t)i * R + r] : 0.0;
    bool dominated = !valid;      // found j with c_i <= c_j and c_i != c_j
    bool dup_before = false;      // found j < i with c_j == c_i
    for (int j0 = 0; j0 < N; j0 += PARETO_TILE) {
        const int nj = min(PARETO_TILE, N - j0);
        __syncthreads();
        for (int e = (int)threadIdx.x; e < nj * R; e += (int)blockDim.x) s_pts[e] = pts[(size_t)j0 * R + e];
        __syncthreads();
        // wave-uniform early exit: nothing left to learn for any candidate of this wave
        const bool settled = dominated;   // a dominated candidate is dropped regardless of duplicates
        if (__ballot(!settled) == 0ull) continue;
        if (!settled) {
            for (int j = 0; j < nj; ++j) {
                bool le = true, eq = true;
#pragma unroll
                for (int r = 0; r < MORL_MAX_OBJ; ++r)
                    if (r < R) {
                        const double cj = s_pts[j * R + r];
                        le = le && (ci[r] <= cj);
                        eq = eq && (ci[r] == cj);
                    }
                if (le && !eq) { dominated = true; break; }
                if (eq && (j0 + j) < i) dup_before = true;
            }
        }
    }
    if (valid) mask[i] = (uint8_t)((!dominated) && !(remove_duplicates && dup_before));
}

}  // namespace morl
