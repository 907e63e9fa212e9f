// Front metrics on the device (gfx950, wave64): exact hypervolume and expected utility (EUM) of an archive that already
// lives in HBM after the Pareto prune (common/performance_indicators.py:15-25, 71-91).
//
// Hypervolume (maximisation, reference point r): the volume of { x : r <= x, x <= p for some point p }.  The reference
// delegates to pymoo's recursive slicing, a serial algorithm; the device form is a SLAB DECOMPOSITION instead: the
// first R-1 axes are cut at every point coordinate (clipped at r), which gives N^(R-1) boxes; inside a box the dominated
// set is a single interval of the last axis, [r_last, max{ p_last : p covers the box }].  So
//     HV = sum_boxes  vol_{R-1}(box) * max(0, max_{p >= box upper corner} p_last - r_last)
// -- every box is independent (one thread per box, points staged in LDS), the result is exact up to the rounding of the
// fp64 sum, and the sum order is fixed (thread-strided partials, butterfly, ordered block partials).  N^(R-1) * N point
// tests: 1e4 for the usual 100-point, 2-objective front, 1e8 at 4 objectives.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

constexpr int HV_MAX_N = 512;     // up to here the points and cut coordinates of a front are staged in LDS (60 KB at 8
                                  // objectives); larger fronts are read in place (L2-resident)
constexpr int HV_THREADS = 256;
constexpr int HV_MAX_BLOCKS = 1024;

// coords[d][0..N): coordinate d of every point, clipped below at ref[d], ascending (rank sort, ties by index).  One
// workgroup per axis d < R-1.
__global__ __launch_bounds__(HV_THREADS) void hv_sort_kernel(const double* __restrict__ pts, int N, int R,
                                                             const double* __restrict__ ref, double* __restrict__ coords) {
    __shared__ double s_v[HV_MAX_N];
    const int d = (int)blockIdx.x;
    const bool staged = N <= HV_MAX_N;
    const double rd = ref[d];
    if (staged) {
        for (int t = (int)threadIdx.x; t < N; t += (int)blockDim.x) s_v[t] = fmax(pts[(size_t)t * R + d], rd);
        __syncthreads();
    }
    for (int t = (int)threadIdx.x; t < N; t += (int)blockDim.x) {
        const double v = staged ? s_v[t] : fmax(pts[(size_t)t * R + d], rd);
        int rank = 0;
        for (int j = 0; j < N; ++j) {
            const double u = staged ? s_v[j] : fmax(pts[(size_t)j * R + d], rd);
            rank += (u < v || (u == v && j < t)) ? 1 : 0;
        }
        coords[(size_t)d * N + rank] = v;
    }
}

__global__ __launch_bounds__(HV_THREADS) void hv_boxes_kernel(const double* __restrict__ pts, int N, int R,
                                                              const double* __restrict__ ref,
                                                              const double* __restrict__ coords, long long n_boxes,
                                                              double* __restrict__ part) {
    __shared__ double s_pts[HV_MAX_N * MORL_MAX_OBJ];         // points [N][R]
    __shared__ double s_co[HV_MAX_N * (MORL_MAX_OBJ - 1)];    // cut coordinates [R-1][N]
    __shared__ double s_red[HV_THREADS / 64];
    const bool staged = N <= HV_MAX_N;
    if (staged) {
        for (int e = (int)threadIdx.x; e < N * R; e += (int)blockDim.x) s_pts[e] = pts[e];
        for (int e = (int)threadIdx.x; e < N * (R - 1); e += (int)blockDim.x) s_co[e] = coords[e];
        __syncthreads();
    }
    const double* __restrict__ P = staged ? s_pts : pts;      // larger fronts: read in place (every workgroup walks the
    const double* __restrict__ CO = staged ? s_co : coords;   // same arrays: L2 / L1 resident)
    const double ref_last = ref[R - 1];
    double acc = 0.0;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < n_boxes; b += (long long)gridDim.x * blockDim.x) {
        double upper[MORL_MAX_OBJ];
        double vol = 1.0;
        long long rem = b;
        for (int d = 0; d < R - 1; ++d) {
            const int i = (int)(rem % N);
            rem /= N;
            const double hi = CO[(size_t)d * N + i];
            const double lo = (i > 0) ? CO[(size_t)d * N + i - 1] : ref[d];
            upper[d] = hi;
            vol *= (hi - lo);
        }
        if (!(vol > 0.0)) continue;                  // an empty slab (duplicate coordinate, or a point not above ref)
        double top = ref_last;
        for (int j = 0; j < N; ++j) {
            bool covers = true;
            for (int d = 0; d < R - 1; ++d) covers = covers && (P[(size_t)j * R + d] >= upper[d]);
            if (covers) top = fmax(top, P[(size_t)j * R + R - 1]);
        }
        acc += vol * (top - ref_last);
    }
    acc = wave_sum(acc);
    if (lane_id() == 0) s_red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < HV_THREADS / 64; ++w) t += s_red[w];
        part[blockIdx.x] = t;
    }
}

// out = scale * sum of part[0..n) in index order (one wave; n <= HV_MAX_BLOCKS)
__global__ __launch_bounds__(64) void metric_finish_kernel(const double* __restrict__ part, int n, double scale,
                                                           double* __restrict__ out) {
    double t = 0.0;
    for (int e = lane_id(); e < n; e += kWave) t += part[e];
    t = wave_sum(t);
    if (lane_id() == 0) *out = t * scale;
}

// expected utility: mean over the weight vectors of max over the front of w . p (fp64, objective order).  One wave per
// weight vector, lanes stride the front; per-block partial sums, finished by metric_finish_kernel with scale 1 / M.
__global__ __launch_bounds__(HV_THREADS) void eum_kernel(const double* __restrict__ front, int N, int R,
                                                         const double* __restrict__ weights, int M,
                                                         double* __restrict__ part) {
    __shared__ double s_red[HV_THREADS / 64];
    const int m = (int)blockIdx.x * (HV_THREADS / 64) + wave_id();
    double best = -INFINITY;
    if (m < M) {
        for (int j = lane_id(); j < N; j += kWave) {
            double s = 0.0;
            for (int r = 0; r < R; ++r) s += weights[(size_t)m * R + r] * front[(size_t)j * R + r];
            best = fmax(best, s);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off));
    }
    if (lane_id() == 0) s_red[wave_id()] = (m < M) ? best : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < HV_THREADS / 64; ++w) t += s_red[w];
        part[blockIdx.x] = t;
    }
}

}  // namespace morl
