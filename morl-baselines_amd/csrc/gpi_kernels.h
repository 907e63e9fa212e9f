// Row-wise kernels of GPI-PD / GPI-LS with discrete actions (multi_policy/gpi_pd/gpi_pd.py), gfx950 wave64.
// The dense layers are the batched MFMA launches of ac_kernels.h / gemm_wave.h; here: the feature product of the
// conditioned Q-net, the min-over-ensemble targets, the envelope (GPI) target over a weight set, the non-standard Huber
// loss with its derivative and the prioritised-replay errors, and the GPI action.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

// ---------------------------------------------------------------------------------------------------------------------
// QNet.forward front end (gpi_pd.py:71-76):  x = relu(Linear_s(obs)) * relu(Linear_w(w)).
// sf comes from a batched GEMM (EPI_BIAS_RELU); the weight embedding has only R inputs, so it is formed here:
// wf[row][j] = relu(sum_r Ww[j][r] * w[row][r] + bw[j]) in ascending r (the order of a K = R GEMM).
// ---------------------------------------------------------------------------------------------------------------------
struct EmbedArgs {
    const float* sf;          // [G][cap][ld]
    float* wf;                // [G][cap][ld]  out (kept for the backward)
    float* x;                 // [G][cap][ld]  out: sf * wf
    const float* w;           // [rows][R] (shared by the nets), row stride w_rstride (0: one vector for all rows)
    int w_rstride;
    const float* params;      // net g: params + g * pstride; Ww at +offWw ([H][R]), bw at +offBw
    long long pstride, offWw, offBw;
    long long gstride;
    int H, ld, R, rows, G;
};

__device__ __forceinline__ void gpi_embed_fwd_body(const EmbedArgs& a) {
    const long long total = (long long)a.G * a.rows * a.H;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e % a.H);
        const long long gr = e / a.H;
        const int row = (int)(gr % a.rows), g = (int)(gr / a.rows);
        const float* __restrict__ p = a.params + (long long)g * a.pstride;
        const float* __restrict__ wr = a.w + (long long)row * a.w_rstride;
        float acc = 0.f;
        for (int r = 0; r < a.R; ++r) acc = fmaf(p[a.offWw + (long long)j * a.R + r], wr[r], acc);
        const float wf = fmaxf(acc + p[a.offBw + j], 0.f);
        const long long o = (long long)g * a.gstride + (long long)row * a.ld + j;
        a.wf[o] = wf;
        a.x[o] = a.sf[o] * wf;
    }
}

__global__ __launch_bounds__(256) void gpi_embed_fwd_kernel(EmbedArgs a) { gpi_embed_fwd_body(a); }
// the two passes of a paired forward (target ensemble at s', online ensemble at s): blockIdx.y picks the pass
__global__ __launch_bounds__(256) void gpi_embed_fwd_pair_kernel(EmbedArgs a, EmbedArgs b) { gpi_embed_fwd_body(blockIdx.y ? b : a); }

// backward of the product: dsf = dx * wf * (sf > 0) (in place of dx), dwf = dx * sf * (wf > 0)
struct EmbedBwdArgs {
    float* dx;                // [G][cap][ld] in: dLoss/dx ; out: dLoss/d(z_s)
    float* dwf;               // [G][cap][ld] out: dLoss/d(z_w)
    const float* sf;
    const float* wf;
    long long gstride;
    int H, ld, rows, G;
};

__global__ __launch_bounds__(256) void gpi_embed_bwd_kernel(EmbedBwdArgs a) {
    const long long total = (long long)a.G * a.rows * a.H;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e % a.H);
        const long long gr = e / a.H;
        const int row = (int)(gr % a.rows), g = (int)(gr / a.rows);
        const long long o = (long long)g * a.gstride + (long long)row * a.ld + j;
        const float d = a.dx[o], s = a.sf[o], w = a.wf[o];
        a.dx[o] = (s > 0.f) ? d * w : 0.f;
        a.dwf[o] = (w > 0.f) ? d * s : 0.f;
    }
}

// gradients of the weight embedding (K = rows, N = R <= 8): dWw[j][r] = sum_rows dwf[row][j] * w[row][r], dbw[j] = sum.
// One workgroup of 16 waves per (64 features, net): lane = feature j, wave w sums rows w, w + 16, ...; partials combined
// in wave order (deterministic).
struct EmbedGradArgs {
    const float* dwf;
    const float* w;
    int w_rstride;
    float* grads;             // net g: grads + g * pstride
    long long pstride, offWw, offBw, gstride;
    int H, ld, R, rows, G;
};

__global__ __launch_bounds__(1024) void gpi_embed_grad_kernel(EmbedGradArgs a) {
    __shared__ float s_p[16][MORL_MAX_OBJ + 1][64];
    const int lane = lane_id(), wave = wave_id();
    const int j = (int)blockIdx.x * 64 + lane, g = (int)blockIdx.y;
    float acc[MORL_MAX_OBJ + 1];
#pragma unroll
    for (int r = 0; r <= MORL_MAX_OBJ; ++r) acc[r] = 0.f;
    if (j < a.H) {
        // eight rows per trip, their loads issued together, accumulated in the same row order (see ac_ln_grad_kernel)
        constexpr int U = 8;
        for (int row0 = wave; row0 < a.rows; row0 += 16 * U) {
            float dv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + 16 * u;
                dv[u] = a.dwf[(long long)g * a.gstride + (long long)(row < a.rows ? row : row0) * a.ld + j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + 16 * u;
                if (row >= a.rows) break;
                const float* __restrict__ wr = a.w + (long long)row * a.w_rstride;
                for (int r = 0; r < a.R; ++r) acc[r] = fmaf(dv[u], wr[r], acc[r]);
                acc[MORL_MAX_OBJ] += dv[u];
            }
        }
    }
#pragma unroll
    for (int r = 0; r <= MORL_MAX_OBJ; ++r) s_p[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave == 0 && j < a.H) {
        float* __restrict__ out = a.grads + (long long)g * a.pstride;
        for (int r = 0; r < a.R; ++r) {
            float t = 0.f;
            for (int w_ = 0; w_ < 16; ++w_) t += s_p[w_][r][lane];
            out[a.offWw + (long long)j * a.R + r] = t;
        }
        float t = 0.f;
        for (int w_ = 0; w_ < 16; ++w_) t += s_p[w_][MORL_MAX_OBJ][lane];
        out[a.offBw + j] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// TD target (gpi_pd.py:446-462): per (row, a) the ensemble member with the smallest scalarised value, then the greedy
// action under w, target = r + (1 - d) * gamma * Q_min(s', a*, :).   One thread per row.
// Envelope / GPI target (gpi_pd.py:662-690): same min over members per (row, k, a), max over a then arg-max over the K
// weights (first maximum in both), gathered.  Scalarisations: sum_r w[r] * q[r] in ascending r.
// ---------------------------------------------------------------------------------------------------------------------
struct GpiTargetArgs {
    const float* qt;          // [nn][cap][ldq]  target nets at (s', w_row)
    const float* qt_env;      // [nn][cap_env][ldq]  target nets at (s'_row, w_k), row index row * K + k; or NULL
    long long gstride, gstride_env;
    int ldq;
    const float* w;           // [rows][R] or one vector
    int w_rstride;
    const float* rewards;     // [rows][R] or NULL (raw max_next_q wanted: _reset_priorities)
    const float* dones;       // [rows]
    float* target;            // [rows][R] or NULL
    float* target_env;        // [rows][R] or NULL
    int rows, A, R, nn, K;
    float gamma;
};

__device__ __forceinline__ float gpi_dot(const float* __restrict__ q, const float* __restrict__ w, int R) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += w[r] * q[r];
    return s;
}

// One WAVE per batch row: lane = candidate (k, a) in k-major order (strided when there are more than 64); each lane takes
// the min over the ensemble of its candidate, then a butterfly arg-max keeps the largest value with the LOWEST candidate
// index -- which is "max over a (first), then arg-max over k (first)" of the reference.
__device__ __forceinline__ void gpi_best_candidate(const float* __restrict__ qbase, long long gstride, int ldq, int row0,
                                                  int K, int A, int R, int nn, const float* __restrict__ w, float& best,
                                                  int& best_i, const float*& best_q) {
    const int lane = lane_id();
    int best_n = 0;
    best = -INFINITY; best_i = 0x7fffffff;
    for (int cand = lane; cand < K * A; cand += kWave) {
        const int k = cand / A, ac = cand % A;
        float ms = 0.f;
        int mn = 0;
        for (int n = 0; n < nn; ++n) {
            const float s = gpi_dot(qbase + (long long)n * gstride + ((long long)row0 + k) * ldq + ac * R, w, R);
            if (n == 0 || s < ms) { ms = s; mn = n; }
        }
        if (ms > best) { best = ms; best_i = cand; best_n = mn; }       // ascending cand per lane: first maximum kept
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(best_i, off), on = __shfl_xor(best_n, off);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; best_n = on; }
    }
    best_q = qbase + (long long)best_n * gstride + ((long long)row0 + best_i / A) * ldq + (best_i % A) * R;
}

__global__ __launch_bounds__(256) void gpi_target_kernel(GpiTargetArgs a) {
    const int row = (int)blockIdx.x * 4 + wave_id();
    if (row >= a.rows) return;
    const int lane = lane_id();
    const float* __restrict__ w = a.w + (long long)row * a.w_rstride;
    const float nd = a.dones ? 1.f - a.dones[row] : 1.f;
    float best;
    int bi;
    const float* bq;
    if (a.target) {
        gpi_best_candidate(a.qt, a.gstride, a.ldq, row, 1, a.A, a.R, a.nn, w, best, bi, bq);
        if (lane < a.R) {
            const float v = bq[lane];
            a.target[(long long)row * a.R + lane] = a.rewards ? a.rewards[(long long)row * a.R + lane] + (nd * a.gamma) * v : v;
        }
    }
    if (a.target_env && a.qt_env) {
        gpi_best_candidate(a.qt_env, a.gstride_env, a.ldq, row * a.K, a.K, a.A, a.R, a.nn, w, best, bi, bq);
        if (lane < a.R) {
            const float v = bq[lane];
            a.target_env[(long long)row * a.R + lane] =
                a.rewards ? a.rewards[(long long)row * a.R + lane] + (nd * a.gamma) * v : v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// loss (gpi_pd.py:464-486): psi_n = Q_n(s, w)[action]; td = psi_n - target; huber(|td|, delta) = mean(where(x < delta,
// 0.5 x^2, delta x)) (common/networks.py:90-100: NOT the textbook Huber); critic_loss = (1 / nn) sum_n.
// dLoss/dpsi = (1 / (nn * rows * R)) * (|td| < delta ? td : delta * sign(td)), zero for the other actions.
// PER errors of the first n_per rows (:503-516): |sum_r w[r] * max_n |psi_n - target|[r]|, same with the envelope target.
// One workgroup; fp64 block sums in fixed order.
// ---------------------------------------------------------------------------------------------------------------------
struct GpiLossArgs {
    const float* q;           // [nn][cap][ldq]
    float* dq;                // [nn][cap][ldq]
    long long gstride;
    int ldq;
    const int32_t* actions;   // [rows]
    const float* target;      // [rows][R]
    const float* target_env;  // [rows][R] or NULL
    const float* w;           // [rows][R]
    float* loss_out;          // scalar or NULL
    float* td_prio;           // [n_per] or NULL
    float* gtd_prio;          // [n_per] or NULL
    int rows, A, R, nn, n_per;
    float delta;
};

__global__ __launch_bounds__(256) void gpi_loss_kernel(GpiLossArgs a) {
    __shared__ double s_red[4];
    const float c = 1.f / ((float)a.nn * (float)a.rows * (float)a.R);
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    for (int row = (int)threadIdx.x; row < a.rows; row += (int)blockDim.x) {
        const int act = a.actions[row];
        const float* __restrict__ w = a.w + (long long)row * a.R;
        float mtd[MORL_MAX_OBJ], mgtd[MORL_MAX_OBJ];
        for (int r = 0; r < a.R; ++r) mtd[r] = mgtd[r] = 0.f;
        for (int n = 0; n < a.nn; ++n) {
            const float* __restrict__ q = a.q + (long long)n * a.gstride + (long long)row * a.ldq;
            float* __restrict__ dq = a.dq + (long long)n * a.gstride + (long long)row * a.ldq;
            for (int e = 0; e < a.ldq; ++e) dq[e] = 0.f;
            for (int r = 0; r < a.R; ++r) {
                const float psi = q[act * a.R + r];
                const float td = psi - a.target[(long long)row * a.R + r];
                const float x = fabsf(td);
                part[n] += (x < a.delta) ? 0.5 * (double)x * (double)x : (double)a.delta * (double)x;
                dq[act * a.R + r] = c * ((x < a.delta) ? td : (td > 0.f ? a.delta : (td < 0.f ? -a.delta : 0.f)));
                mtd[r] = fmaxf(mtd[r], x);
                if (a.target_env) mgtd[r] = fmaxf(mgtd[r], fabsf(psi - a.target_env[(long long)row * a.R + r]));
            }
        }
        if (row < a.n_per) {
            if (a.td_prio) a.td_prio[row] = fabsf(gpi_dot(mtd, w, a.R));
            if (a.gtd_prio && a.target_env) a.gtd_prio[row] = fabsf(gpi_dot(mgtd, w, a.R));
        }
    }
    double total = 0.0;
    for (int n = 0; n < a.nn; ++n) {
        double v = wave_sum(part[n]);
        __syncthreads();
        if (lane_id() == 0) s_red[wave_id()] = v;
        __syncthreads();
        double t = 0.0;
        for (int w_ = 0; w_ < (int)(blockDim.x >> 6); ++w_) t += s_red[w_];
        total += t / ((double)a.rows * a.R);
    }
    if (threadIdx.x == 0 && a.loss_out) *a.loss_out = (float)(total / (double)a.nn);
}

// double-Q target of _reset_priorities without gpi_pd (gpi_pd.py:641-649): a* = argmax_a w . Q_online0(s', a); value from
// Q_target0(s', a*); out = r + (1 - d) * gamma * value
__global__ __launch_bounds__(256) void gpi_ddqn_target_kernel(const float* __restrict__ qo, const float* __restrict__ qt,
                                                              int ldq, const float* __restrict__ w,
                                                              const float* __restrict__ rewards,
                                                              const float* __restrict__ dones, int rows, int A, int R,
                                                              float gamma, float* __restrict__ out) {
    const int row = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (row >= rows) return;
    float best = 0.f;
    int ba = 0;
    for (int ac = 0; ac < A; ++ac) {
        const float s = gpi_dot(qo + (long long)row * ldq + ac * R, w, R);
        if (ac == 0 || s > best) { best = s; ba = ac; }
    }
    const float nd = 1.f - dones[row];
    for (int r = 0; r < R; ++r)
        out[(long long)row * R + r] = rewards[(long long)row * R + r] + (nd * gamma) * qt[(long long)row * ldq + ba * R + r];
}

// |w . (target - Q(s, w)[action])| of every row (gpi_pd.py:652-653, the priorities of _reset_priorities)
__global__ __launch_bounds__(256) void gpi_gtd_kernel(const float* __restrict__ q, int ldq,
                                                      const int32_t* __restrict__ actions,
                                                      const float* __restrict__ target, const float* __restrict__ w,
                                                      int w_rstride, int rows, int R, float* __restrict__ out) {
    const int row = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (row >= rows) return;
    const float* __restrict__ qa = q + (long long)row * ldq + actions[row] * R;
    const float* __restrict__ wr = w + (long long)row * w_rstride;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += wr[r] * (target[(long long)row * R + r] - qa[r]);
    out[row] = fabsf(s);
}

// ---------------------------------------------------------------------------------------------------------------------
// GPI action (gpi_pd.py:564-582): q[i][a][:] = Q_0(s, a, w_i) for the M support weights; scalarise with the CURRENT w,
// max over a (first), arg-max over i (first) -> action of the best policy.  With one "support" row holding the min over
// the ensemble this is also max_action (:610-617).  One wave.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gpi_action_kernel(const float* __restrict__ q, int ldq, int M, int A, int R,
                                                        const float* __restrict__ w, int32_t* __restrict__ action_out,
                                                        int32_t* __restrict__ policy_out) {
    const int lane = lane_id();
    float best = -INFINITY;
    int best_i = 0x7fffffff, best_a = 0;
    for (int i = lane; i < M; i += kWave) {
        float bq = 0.f;
        int ba = 0;
        for (int ac = 0; ac < A; ++ac) {
            const float s = gpi_dot(q + (long long)i * ldq + ac * R, w, R);
            if (ac == 0 || s > bq) { bq = s; ba = ac; }
        }
        if (bq > best) { best = bq; best_i = i; best_a = ba; }    // ascending i per lane: first maximum kept
    }
    // butterfly arg-max with lowest-index tie break (ballot-free: M is tiny)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(best_i, off), oa = __shfl_xor(best_a, off);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; best_a = oa; }
    }
    if (lane == 0) {
        *action_out = best_a;
        if (policy_out) *policy_out = best_i;
    }
}

// GPI actions of a BATCH of observations (the Dyna rollouts, gpi_pd.py:377-387): q rows (obs_i, w_k) at i * M + k from
// net 0; one wave per observation: arg-max over (k, a) of w . q, lowest index on ties -> a.  w_rstride = R gives every
// observation its own weight vector (evaluation episodes of many weights in lock-step), 0 shares one.
__global__ __launch_bounds__(256) void gpi_actions_kernel(const float* __restrict__ q, int ldq, int n, int M, int A, int R,
                                                          const float* __restrict__ w, int w_rstride,
                                                          int32_t* __restrict__ actions) {
    const int i = (int)blockIdx.x * 4 + wave_id();
    if (i >= n) return;
    float best;
    int bi;
    const float* bq;
    gpi_best_candidate(q, 0, ldq, i * M, M, A, R, 1, w + (long long)i * w_rstride, best, bi, bq);
    if (lane_id() == 0) actions[i] = bi % A;
}

// element-wise min over the ensemble (max_action's th.min(th.stack(...), dim=0)[0], gpi_pd.py:612)
// (out may alias net 0's rows of q: every element is read before it is written, by the same thread)
__global__ __launch_bounds__(256) void gpi_min_nets_kernel(const float* q, long long gstride, int nn, int n, float* out) {
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (e >= n) return;
    float m = q[e];
    for (int k = 1; k < nn; ++k) m = fminf(m, q[(long long)k * gstride + e]);
    out[e] = m;
}

// per-net clip_grad_norm_ (gpi_pd.py:498-499): one workgroup per net; fp64 sum of squares in fixed order, then scale
__global__ __launch_bounds__(1024) void gpi_clip_kernel(float* __restrict__ grads, long long P, float max_norm,
                                                        float* __restrict__ norm_out) {
    __shared__ double s_red[16];
    float* __restrict__ g = grads + (long long)blockIdx.x * P;
    double ss = 0.0;
    for (long long p = threadIdx.x; p < P; p += blockDim.x) ss += (double)g[p] * (double)g[p];
    ss = wave_sum(ss);
    if (lane_id() == 0) s_red[wave_id()] = ss;
    __syncthreads();
    double t = 0.0;
    for (int w_ = 0; w_ < (int)(blockDim.x >> 6); ++w_) t += s_red[w_];
    const float total = sqrtf((float)t);
    if (threadIdx.x == 0 && norm_out) norm_out[blockIdx.x] = total;
    const float coef = fminf(__fdiv_rn(max_norm, __fadd_rn(total, 1e-6f)), 1.0f);
    for (long long p = threadIdx.x; p < P; p += blockDim.x) g[p] = __fmul_rn(g[p], coef);
}

}  // namespace morl
