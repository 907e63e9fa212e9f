// Gradient finalisation, clip_grad_norm_ + Adam, Polyak -- all HBM-bound streaming kernels over the
// flat parameter buffer (P floats; 211 218 for the flagship Q-net).
#pragma once
#include "morl_device.h"

namespace morl {

constexpr int OPT_THREADS = 256;
constexpr int OPT_MAX_BLOCKS = 1024;  // sum-of-squares partials are re-reduced by every clip_adam wave (16 loads per lane)

// ----------------------------------------------------------------------------------------------
// grads[p] = sum_{s < splits} slabs[s][p]  (fixed order -> run-to-run deterministic), plus one
// sum-of-squares partial per block (fp64) and, in block 0, the final loss reduction:
//   loss = (1-lambda) * sum_b part[b][0] / (W*B*R) + lambda * sum_b part[b][1] / (W*B)   (envelope.py:307-313)
// Algorithmic bytes: 4*P*(splits + 1).
// ----------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(OPT_THREADS) void grad_reduce_kernel(const float* __restrict__ slabs, int splits,
                                                                  long long slab_stride, float* __restrict__ grads,
                                                                  long long P, double* __restrict__ sumsq_part,
                                                                  const double* __restrict__ loss_part, int n_loss,
                                                                  double inv_mse, double inv_aux, float lambda,
                                                                  float* __restrict__ loss_out) {
    __shared__ double s_red[OPT_THREADS / 64];
    double ss = 0.0;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P;
         p += (long long)gridDim.x * blockDim.x) {
        // 8 independent loads in flight per step (the adds stay in slab order -> same bits run to run)
        float g = slabs[p];
        int s = 1;
        for (; s + 8 <= splits; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(s + u) * slab_stride + p];
#pragma unroll
            for (int u = 0; u < 8; ++u) g += v[u];
        }
        for (; s < splits; ++s) g += slabs[(size_t)s * slab_stride + p];
        grads[p] = g;
        ss += (double)g * (double)g;
    }
    ss = wave_sum(ss);
    if (lane_id() == 0) s_red[wave_id()] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += s_red[w];
        sumsq_part[blockIdx.x] = t;
    }
    if (blockIdx.x == 0 && loss_out != nullptr && wave_id() == 0) {
        double a = 0.0, c = 0.0;
        for (int e = lane_id(); e < n_loss; e += kWave) { a += loss_part[2 * e]; c += loss_part[2 * e + 1]; }
        a = wave_sum(a);
        c = wave_sum(c);
        if (lane_id() == 0) {
            const float mse = (float)(a * inv_mse);
            float loss = mse;
            if (lambda > 0.f) loss = (1.0f - lambda) * mse + lambda * (float)(c * inv_aux);
            *loss_out = loss;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// clip_grad_norm_ (envelope.py:324-325) + torch _single_tensor_adam (envelope.py:326), fused:
//   total = sqrt(sum g^2); coef = min(1, max_norm / (total + 1e-6)); g *= coef           (clip, in place)
//   m = m + (g - m) * (1 - b1)                                                            (lerp_)
//   v = v * b2 + (1 - b2) * g * g                                                         (mul_, addcmul_)
//   p = p + (-lr / (1 - b1^t)) * (m / (sqrt(v) / sqrt(1 - b2^t) + eps))                   (addcdiv_)
// step_size and bias_correction2_sqrt are formed on the host in float64 exactly as torch does, then
// rounded to fp32 once.  Every block re-reduces the <= 256 sum-of-squares partials in the same order,
// so all blocks see the identical norm without a grid barrier.
// Algorithmic bytes: 4*P*7 (read p,g,m,v; write p,m,v) + 4*P (clipped g written back).
// ----------------------------------------------------------------------------------------------
// (`nblk` workgroups share the parameters: the kernel that carries the PER tree update as one extra workgroup -- morl_hip.hip:
// clip_adam_per_kernel -- has one more than take part here)
__device__ __forceinline__ void clip_adam_body(float* __restrict__ params, float* __restrict__ grads, float* __restrict__ exp_avg,
                                               float* __restrict__ exp_avg_sq, long long P, const double* __restrict__ sumsq_part,
                                               int n_part, float max_norm, float one_minus_b1, float b2, float one_minus_b2,
                                               float neg_step_size, float bc2_sqrt, float eps, int apply_step,
                                               float* __restrict__ grad_norm_out, int nblk,
                                               const unsigned int* __restrict__ skip_flag = nullptr) {
    // skip_flag (the sharded steps over the single-hop transport): non-zero = a bounded wait of the collectives that produced
    // `grads` ran out (a peer never arrived, morl_comm.hip) -- the sums are garbage, so the optimiser must not move; the host finds
    // the same word through morl_comm_poll / morl_comm_check and raises
    if (skip_flag != nullptr && *skip_flag != 0u) apply_step = 0;
    double t = 0.0;
    for (int e = lane_id(); e < n_part; e += kWave) t += sumsq_part[e];
    t = wave_sum(t);
    const float total = sqrtf((float)t);
    float coef = 1.0f;
    if (max_norm >= 0.f) coef = fminf(__fdiv_rn(max_norm, __fadd_rn(total, 1e-6f)), 1.0f);
    if (blockIdx.x == 0 && threadIdx.x == 0 && grad_norm_out) *grad_norm_out = total;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P;
         p += (long long)nblk * blockDim.x) {
        float g = grads[p];
        if (max_norm >= 0.f) { g = __fmul_rn(g, coef); grads[p] = g; }
        if (!apply_step) continue;
        float m = exp_avg[p], v = exp_avg_sq[p];
        m = fmaf(one_minus_b1, __fsub_rn(g, m), m);                       // lerp_ (weight < 0.5 branch)
        v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(one_minus_b2, g), g));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
        const float q = __fdiv_rn(m, denom);
        params[p] = __fadd_rn(params[p], __fmul_rn(neg_step_size, q));
        exp_avg[p] = m;
        exp_avg_sq[p] = v;
    }
}

static __global__ __launch_bounds__(OPT_THREADS) void clip_adam_kernel(float* __restrict__ params, float* __restrict__ grads,
                                                                float* __restrict__ exp_avg,
                                                                float* __restrict__ exp_avg_sq, long long P,
                                                                const double* __restrict__ sumsq_part, int n_part,
                                                                float max_norm, float one_minus_b1, float b2,
                                                                float one_minus_b2, float neg_step_size,
                                                                float bc2_sqrt, float eps, int apply_step,
                                                                float* __restrict__ grad_norm_out,
                                                                const unsigned int* __restrict__ skip_flag) {
    clip_adam_body(params, grads, exp_avg, exp_avg_sq, P, sumsq_part, n_part, max_norm, one_minus_b1, b2, one_minus_b2,
                   neg_step_size, bc2_sqrt, eps, apply_step, grad_norm_out, (int)gridDim.x, skip_flag);
}

// polyak_update (common/networks.py:120-139): tau == 1 -> copy, else t = t*(1-tau) + tau*p
static __global__ __launch_bounds__(OPT_THREADS) void polyak_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             long long n, float tau, float one_minus_tau) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (long long)gridDim.x * blockDim.x) {
        if (tau == 1.0f) dst[p] = src[p];
        else dst[p] = __fadd_rn(__fmul_rn(dst[p], one_minus_tau), __fmul_rn(tau, src[p]));
    }
}

}  // namespace morl
