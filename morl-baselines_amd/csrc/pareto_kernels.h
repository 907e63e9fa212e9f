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
// whole wave skip the remaining tiles once all 64 of its candidates are settled.  The j axis is additionally split over
// blockIdx.y (see the kernel) so that mid-sized fronts still put many waves on every SIMD.
// Integer/boolean result -> bit-exact.  Work N^2*R compares; bytes N*R*8 per workgroup (L2 hits).
constexpr int PARETO_TILE = 512;
constexpr int PARETO_THREADS = 256;
constexpr int PARETO_UNROLL = 4;     // j points tested per loop trip, no branch between them

// R is a template parameter: the objective loop unrolls into straight v_cmp_{le,eq}_f64 pairs on registers / LDS
// broadcasts (the generic-R form with a predicated 8-trip loop and a data-dependent break per point measured ~600 cycles
// per pair test: 17 ms for 65 536 candidates).  The tile is padded to a multiple of PARETO_UNROLL with NaN points, which
// compare false both ways, so the unrolled body needs no bounds test; settled waves leave at the next group of 16 points.
template <int R>
__global__ __launch_bounds__(PARETO_THREADS) void pareto_mask_kernel(const double* __restrict__ pts, int N,
                                                                     int remove_duplicates, int chunk,
                                                                     uint8_t* __restrict__ mask) {
    // blockIdx.x: 256 candidates i; blockIdx.y: the j range [y * chunk, (y + 1) * chunk) they are tested against.  mask is
    // preset to 1; a candidate that meets a dominator (or, with remove_duplicates, an earlier duplicate) in ANY j range is
    // cleared with a plain byte store -- every writer stores the same 0, so the ranges need no atomics and no second pass.
    // Splitting j is what fills the chip: N / 64 waves alone are one wave per SIMD at N = 65 536, with every LDS read and
    // every v_cmp -> s_and dependency exposed.
    __shared__ double s_pts[PARETO_TILE * R];
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool valid = i < N;
    double ci[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ci[r] = valid ? pts[(size_t)i * R + r] : 0.0;
    // bit 1 of `remove_duplicates` (bench_front.py's roofline leg only): no early exit -- every one of the N^2 pair tests is
    // executed, so the launch's duration prices a KNOWN number of compares; the mask is the same
    const bool exits = (remove_duplicates & 2) == 0;
    remove_duplicates &= 1;
    bool dominated = !valid;      // found j with c_i <= c_j and c_i != c_j
    bool dup_before = false;      // found j < i with c_j == c_i
    const int jbeg = (int)blockIdx.y * chunk, jend = min(N, jbeg + chunk);
    for (int j0 = jbeg; j0 < jend; j0 += PARETO_TILE) {
        const int nj = min(PARETO_TILE, jend - j0);
        const int nj_pad = (nj + PARETO_UNROLL - 1) / PARETO_UNROLL * PARETO_UNROLL;
        __syncthreads();
        for (int e = (int)threadIdx.x; e < nj_pad * R; e += (int)blockDim.x)
            s_pts[e] = (e < nj * R) ? pts[(size_t)j0 * R + e] : __builtin_nan("");
        __syncthreads();
        // wave-uniform early exit: a dominated candidate is dropped regardless of duplicates
        if (exits && __ballot(!dominated) == 0ull) continue;
        for (int j = 0; j < nj_pad; j += PARETO_UNROLL) {
            bool dom = false, dup = false;
#pragma unroll
            for (int u = 0; u < PARETO_UNROLL; ++u) {
                bool le = true, eq = true;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double cj = s_pts[(j + u) * R + r];
                    le = le && (ci[r] <= cj);
                    eq = eq && (ci[r] == cj);
                }
                dom = dom || (le && !eq);
                dup = dup || (eq && (j0 + j + u) < i);
            }
            dominated = dominated || dom;
            dup_before = dup_before || dup;
            if (exits && (j & 15) == 12 && __ballot(!dominated) == 0ull) break;
        }
    }
    if (valid && (dominated || (remove_duplicates && dup_before))) mask[i] = 0;
}

}  // namespace morl
