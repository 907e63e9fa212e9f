// Row-wise kernels of the probabilistic dynamics ensemble (common/model_based/probabilistic_ensemble.py), gfx950 wave64.
// The members' dense layers are the batched MFMA launches of ac_kernels.h (member = batch axis).
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

__device__ __forceinline__ float ens_softplus(float x) {       // F.softplus: beta 1, threshold 20
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float ens_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// input normalisation h = (x - mu) / sigma into the members' input rows (probabilistic_ensemble.py:92-95)
struct EnsNormArgs {
    const float* x;           // [E or 1][rows][in]
    long long x_gstride;      // 0: shared by the members
    const float* mu;          // [in] or NULL
    const float* sigma;
    float* dst;               // [E][cap][ld]
    long long dst_gstride;
    int in_dim, ld, rows, E;
};

__global__ __launch_bounds__(256) void ens_norm_kernel(EnsNormArgs a) {
    const long long total = (long long)a.E * a.rows * a.ld;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % a.ld);
        const long long gr = e / a.ld;
        const int row = (int)(gr % a.rows), g = (int)(gr / a.rows);
        float v = 0.f;
        if (c < a.in_dim) {
            v = a.x[(long long)g * a.x_gstride + (long long)row * a.in_dim + c];
            if (a.mu) v = (v - a.mu[c]) / a.sigma[c];
        }
        a.dst[(long long)g * a.dst_gstride + (long long)row * a.ld + c] = v;
    }
}

// bounded log-variance of the raw head output (probabilistic_ensemble.py:112-114)
__device__ __forceinline__ float ens_bound(float raw, float mx, float mn, float& s_max, float& s_min) {
    const float lv1 = mx - ens_softplus(mx - raw);
    s_max = ens_sigmoid(mx - raw);          // d lv1 / d raw        (and 1 - s_max = d lv1 / d max)
    const float lv = mn + ens_softplus(lv1 - mn);
    s_min = ens_sigmoid(lv1 - mn);          // d lv / d lv1         (and 1 - s_min = d lv / d min)
    return lv;
}

// Gaussian NLL (F.gaussian_nll_loss, full = False, eps 1e-6, reduction none -> mean over E * rows * out) + its
// derivative w.r.t. the head output (mean | raw logvar), + per-workgroup partial sums of the loss and of the
// max / min_logvar gradients.  One wave per (member, row); lanes stride the outputs.
struct EnsNllArgs {
    const float* head;        // [E][cap][ld]: mean (cols 0..O-1) | raw logvar (cols O..2O-1)
    float* dhead;             // [E][cap][ld]
    long long gstride;
    int ld;
    const float* y;           // [E][rows][O]
    const float* bounds;      // [2][O]: max_logvar | min_logvar
    double* part;             // [n_blocks][1 + 2 * O]  loss, d max[O], d min[O]
    int rows, O, E;
    float inv_count;          // 1 / (E * rows * O)
};

constexpr int ENS_MAX_OUT = 64;

__global__ __launch_bounds__(256) void ens_nll_kernel(EnsNllArgs a) {
    __shared__ double s_acc[4][1 + 2 * ENS_MAX_OUT];
    const int lane = lane_id(), wave = wave_id();
    const long long wid = (long long)blockIdx.x * 4 + wave;          // (member, row) pair of this wave
    double loss = 0.0;
    float dmx[(ENS_MAX_OUT + 63) / 64], dmn[(ENS_MAX_OUT + 63) / 64];
#pragma unroll
    for (int j = 0; j < (ENS_MAX_OUT + 63) / 64; ++j) dmx[j] = dmn[j] = 0.f;
    if (wid < (long long)a.E * a.rows) {
        const int g = (int)(wid / a.rows), row = (int)(wid % a.rows);
        const float* __restrict__ hd = a.head + (long long)g * a.gstride + (long long)row * a.ld;
        float* __restrict__ dh = a.dhead + (long long)g * a.gstride + (long long)row * a.ld;
        const float* __restrict__ y = a.y + ((long long)g * a.rows + row) * a.O;
#pragma unroll
        for (int j = 0; j < (ENS_MAX_OUT + 63) / 64; ++j) {
            const int o = lane + 64 * j;
            if (o < a.O) {
                const float mean = hd[o], raw = hd[a.O + o];
                float s_max, s_min;
                const float lv = ens_bound(raw, a.bounds[o], a.bounds[a.O + o], s_max, s_min);
                const float var = expf(lv);
                const float vc = fmaxf(var, 1e-6f);                  // gaussian_nll_loss clamps the variance
                const float d = mean - y[o];
                loss += 0.5 * ((double)logf(vc) + (double)(d * d / vc));
                const float dmean = a.inv_count * d / vc;
                // d/dvar of 0.5 (log var + d^2 / var) = 0.5 (1 / var - d^2 / var^2); var = exp(lv) -> times var (0 if clamped)
                const float dlv = (var > 1e-6f) ? a.inv_count * 0.5f * (1.f - d * d / vc) : 0.f;
                dh[o] = dmean;
                dh[a.O + o] = dlv * s_min * s_max;
                dmx[j] = dlv * s_min * (1.f - s_max);
                dmn[j] = dlv * (1.f - s_min);
            }
        }
        for (int c = 2 * a.O + lane; c < a.ld; c += 64) dh[c] = 0.f;
    }
    loss = wave_sum(loss);
    if (lane == 0) s_acc[wave][0] = loss;
#pragma unroll
    for (int j = 0; j < (ENS_MAX_OUT + 63) / 64; ++j) {
        const int o = lane + 64 * j;
        if (o < a.O) { s_acc[wave][1 + o] = (double)dmx[j]; s_acc[wave][1 + a.O + o] = (double)dmn[j]; }
    }
    __syncthreads();
    for (int e = (int)threadIdx.x; e < 1 + 2 * a.O; e += (int)blockDim.x)
        a.part[(long long)blockIdx.x * (1 + 2 * a.O) + e] = ((s_acc[0][e] + s_acc[1][e]) + s_acc[2][e]) + s_acc[3][e];
}

// finalise: loss = sum(part[.][0]) * inv_count + 0.01 * (sum max - sum min); gradients of the bounds (+-0.01 added),
// then their Adam step (no weight decay).  One workgroup.
__global__ __launch_bounds__(256) void ens_bounds_step_kernel(const double* __restrict__ part, int n_blocks, int O,
                                                              float inv_count, float* __restrict__ bounds,
                                                              float* __restrict__ m_, float* __restrict__ v_,
                                                              float neg_step_size, float bc2_sqrt, float one_minus_b1,
                                                              float b2, float one_minus_b2, float eps,
                                                              float* __restrict__ loss_out) {
    __shared__ double s_red[4];
    const int stride = 1 + 2 * O;
    // the loss uses the bounds as they were when the forward ran: evaluate it before they are stepped
    double l = 0.0;
    for (int b = (int)threadIdx.x; b < n_blocks; b += (int)blockDim.x) l += part[(long long)b * stride];
    l *= (double)inv_count;
    for (int e = (int)threadIdx.x; e < 2 * O; e += (int)blockDim.x) l += (e < O ? 0.01 : -0.01) * (double)bounds[e];
    l = wave_sum(l);
    if (lane_id() == 0) s_red[wave_id()] = l;
    __syncthreads();
    if (threadIdx.x == 0 && loss_out) *loss_out = (float)(((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]);
    // one wave per bound: lanes stride over the partial blocks (fixed order, so the sum is reproducible), butterfly after
    // (a wave's bounds in batches of eight: their partial sums are fetched together and lane k of the wave steps the k-th of them
    // -- the plain loop ran one load -> butterfly -> Adam chain per bound, ten in a row for 40 bounds: 9.8 us)
    const int n_waves = (int)blockDim.x / 64, lane = lane_id();
    constexpr int U = 8;
    for (int e0 = wave_id(); e0 < 2 * O; e0 += n_waves * U) {
        double g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + n_waves * u;
            g[u] = 0.0;
            if (e < 2 * O)
                for (int b = lane; b < n_blocks; b += 64) g[u] += part[(long long)b * stride + 1 + e];
        }
        double mine = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double t = wave_sum(g[u]);
            if (lane == u) mine = t;
        }
        const int e = e0 + n_waves * lane;
        if (lane < U && e < 2 * O) {
            const float grad = (float)mine + (e < O ? 0.01f : -0.01f);
            float m = m_[e], v = v_[e];
            m = fmaf(one_minus_b1, __fsub_rn(grad, m), m);
            v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(one_minus_b2, grad), grad));
            const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
            m_[e] = m;
            v_[e] = v;
            bounds[e] = __fadd_rn(bounds[e], __fmul_rn(neg_step_size, __fdiv_rn(m, denom)));
        }
    }
}

// torch Adam with L2 weight decay (grad += wd * p before the moments), per-layer coefficient looked up by offset
struct EnsAdamArgs {
    float* params;
    const float* grads;
    float* exp_avg;
    float* exp_avg_sq;
    long long Pm;             // parameters of one member
    long long layer_end[MORL_MAX_LAYERS];   // exclusive end offset (within a member) of layer l's W and b
    float wd[MORL_MAX_LAYERS];
    int n_layers;
    float neg_step_size, bc2_sqrt, one_minus_b1, b2, one_minus_b2, eps;
    long long total;          // E * Pm
};

__global__ __launch_bounds__(256) void ens_adam_kernel(EnsAdamArgs a) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < a.total; p += (long long)gridDim.x * blockDim.x) {
        const long long off = p % a.Pm;
        int l = 0;
        while (l + 1 < a.n_layers && off >= a.layer_end[l]) ++l;
        const float w = a.params[p];
        const float g = fmaf(a.wd[l], w, a.grads[p]);               // grad.add(param, alpha=weight_decay)
        float m = a.exp_avg[p], v = a.exp_avg_sq[p];
        m = fmaf(a.one_minus_b1, __fsub_rn(g, m), m);
        v = __fadd_rn(__fmul_rn(v, a.b2), __fmul_rn(__fmul_rn(a.one_minus_b2, g), g));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), a.bc2_sqrt), a.eps);
        a.params[p] = __fadd_rn(w, __fmul_rn(a.neg_step_size, __fdiv_rn(m, denom)));
        a.exp_avg[p] = m;
        a.exp_avg_sq[p] = v;
    }
}

// outputs of a forward: mean and bounded logvar, compacted to [E][rows][O]
struct EnsOutArgs {
    const float* head;
    long long gstride;
    int ld;
    const float* bounds;
    float* mean;
    float* logvar;            // or NULL
    int rows, O, E;
};

__global__ __launch_bounds__(256) void ens_out_kernel(EnsOutArgs a) {
    const long long total = (long long)a.E * a.rows * a.O;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(e % a.O);
        const long long gr = e / a.O;
        const int row = (int)(gr % a.rows), g = (int)(gr / a.rows);
        const float* __restrict__ hd = a.head + (long long)g * a.gstride + (long long)row * a.ld;
        a.mean[e] = hd[o];
        if (a.logvar) {
            float s1, s2;
            a.logvar[e] = ens_bound(hd[a.O + o], a.bounds[o], a.bounds[a.O + o], s1, s2);
        }
    }
}

// per-member MSE of the means against shared targets: mse[g] = mean_{rows, O} (mean - y)^2.  One workgroup per member.
__global__ __launch_bounds__(256) void ens_mse_kernel(const float* __restrict__ head, long long gstride, int ld,
                                                      const float* __restrict__ y, int rows, int O,
                                                      float* __restrict__ mse_out) {
    __shared__ double s_red[4];
    const int g = (int)blockIdx.x;
    double s = 0.0;
    for (long long e = threadIdx.x; e < (long long)rows * O; e += blockDim.x) {
        const int row = (int)(e / O), o = (int)(e % O);
        const float d = head[(long long)g * gstride + (long long)row * ld + o] - y[e];
        s += (double)d * (double)d;
    }
    s = wave_sum(s);
    if (lane_id() == 0) s_red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) mse_out[g] = (float)((((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]) / ((double)rows * O));
}

}  // namespace morl
