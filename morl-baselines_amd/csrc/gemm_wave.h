// Wave-level exact-fp32 MFMA GEMM for LATENCY-bound problems (gfx950, wave64).
//
// The actor-critic learners run 128-256 batch rows through [256, 256] layers: 17 MFLOP per layer.  The LDS-tiled engine
// of gemm_f32.h gives such a problem to two workgroups, each a >= 13.6 us dependent chain of 512 MFMAs per wave plus a
// barrier per 32-deep chunk -- the whole chip waits on 8 waves.  Here every wave owns ONE 32 x 32 tile of C and the
// full K range: 4x the waves, no LDS, no barriers, operands streamed global/L2 -> registers (the weights of these nets
// are L2 resident: 0.26 MB per layer), a 64-cycle v_mfma_f32_32x32x2_f32 per two k.  The k loop accumulates into one
// register tile in ascending k, exactly like gemm_tile<> -- results are bit-identical to the tiled engine.  Used for
// K <= 32 (first layers); deeper contractions use the split-K form at the end of this file.
//
// Operand addressing of lane (i = lane & 31, h = lane >> 5) for the MFMA of k-pair (k, k+1): A(m0+i, k+h), B(k+h, n0+i).
//   *_KC = 1 : operand is K-contiguous in memory (activations X[m][k], weights W[n][k]): one 16-byte load covers the
//              lane's elements of two MFMAs (k4+h and k4+2+h);
//   *_KC = 0 : operand is M/N-contiguous (dZ^T for dW, W for dX): 4-byte loads, 128 B coalesced per half-wave.
#pragma once
#include "gemm_f32.h"

namespace morl {

constexpr int WG_CHUNK = 16;   // k values per register stage

template <bool KC>
struct WaveStage {
    float v[WG_CHUNK / 2];     // this lane's operand elements for the 8 MFMAs of a stage
};

// load the lane's elements of k range [k0, k0 + 16): element s <-> k = k0 + 2 * s + h
template <bool KC>
__device__ __forceinline__ void wave_load(WaveStage<KC>& st, const float* __restrict__ P, int ld, int idx, int extent,
                                          int k0, int kend, int h, int vec) {
    if (KC) {
        const bool ok = idx < extent;
        const float* __restrict__ row = P + (size_t)(ok ? idx : 0) * ld;
#pragma unroll
        for (int q = 0; q < WG_CHUNK / 4; ++q) {
            const int k4 = k0 + 4 * q;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                if (vec && k4 + 3 < kend) {
                    t = *reinterpret_cast<const float4*>(row + k4);
                } else {
                    if (k4 < kend) t.x = row[k4];
                    if (k4 + 1 < kend) t.y = row[k4 + 1];
                    if (k4 + 2 < kend) t.z = row[k4 + 2];
                    if (k4 + 3 < kend) t.w = row[k4 + 3];
                }
            }
            st.v[2 * q] = h ? t.y : t.x;
            st.v[2 * q + 1] = h ? t.w : t.z;
        }
    } else {
        const bool ok = idx < extent;
#pragma unroll
        for (int s = 0; s < WG_CHUNK / 2; ++s) {
            const int k = k0 + 2 * s + h;
            st.v[s] = (ok && k < kend) ? P[(size_t)k * ld + idx] : 0.f;
        }
    }
}

template <bool A_KC, bool B_KC, int EPI>
__device__ __forceinline__ void gemm_wave_tile(const GemmProblem& g, int tile_m, int tile_n) {
    const int lane = lane_id();
    const int h = lane >> 5, i = lane & 31;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float colsum = 0.f;
    const bool do_colsum = (g.colsum != nullptr) && (tile_n == 0);

    WaveStage<A_KC> a0, a1;
    WaveStage<B_KC> b0, b1;
    const int K = g.K;
    wave_load<A_KC>(a0, g.A, g.lda, m0 + i, g.M, 0, K, h, g.a_vec);
    wave_load<B_KC>(b0, g.B, g.ldb, n0 + i, g.N, 0, K, h, g.b_vec);
    for (int k0 = 0; k0 < K; k0 += 2 * WG_CHUNK) {
        wave_load<A_KC>(a1, g.A, g.lda, m0 + i, g.M, k0 + WG_CHUNK, K, h, g.a_vec);
        wave_load<B_KC>(b1, g.B, g.ldb, n0 + i, g.N, k0 + WG_CHUNK, K, h, g.b_vec);
#pragma unroll
        for (int s = 0; s < WG_CHUNK / 2; ++s) {
            acc = mfma32(a0.v[s], b0.v[s], acc);
            if (do_colsum) { const float odd = __shfl_xor(a0.v[s], 32); colsum += a0.v[s]; colsum += odd; }
        }
        wave_load<A_KC>(a0, g.A, g.lda, m0 + i, g.M, k0 + 2 * WG_CHUNK, K, h, g.a_vec);
        wave_load<B_KC>(b0, g.B, g.ldb, n0 + i, g.N, k0 + 2 * WG_CHUNK, K, h, g.b_vec);
#pragma unroll
        for (int s = 0; s < WG_CHUNK / 2; ++s) {
            acc = mfma32(a1.v[s], b1.v[s], acc);
            if (do_colsum) { const float odd = __shfl_xor(a1.v[s], 32); colsum += a1.v[s]; colsum += odd; }
        }
    }

    const int col = n0 + i;
    float bias = 0.f;
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && col < g.N) bias = g.bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M && col < g.N) {
            float v = acc[r];
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) v += bias;
            if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
            if (EPI == EPI_RELU_MASK) v = (g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v : 0.f;
            g.C[(size_t)row * g.ldc + col] = v;
        }
    }
    // half-wave 0 added (even k, then its partner's odd k) in ascending k: the order of gemm_tile's column sums
    if (do_colsum && h == 0 && m0 + i < g.M) g.colsum[m0 + i] = colsum;
}

// ----------------------------------------------------------------------------------------------------------------
// Split-K form for K > 32: the FOUR waves of a workgroup share one 32 x 32 tile, wave w owning the contiguous k range
// [w * kper, (w + 1) * kper).  A wave's whole range (64 k at K = 256) is fetched with all its loads in flight at once --
// the single-wave form above exposes one memory round trip per 16 k (measured 42 us for a 128 x 256 x 256 layer: 16
// dependent trips on a chip whose other 1000 SIMDs idle); here it is ONE trip, 32 MFMAs and an LDS reduction of the four
// partial tiles in wave order (deterministic; the summation order differs from the single-chain engines by design).
// ----------------------------------------------------------------------------------------------------------------
constexpr int WG4_CHUNK = 64;   // k values a wave keeps in registers at a time

template <bool KC>
__device__ __forceinline__ void wave_load64(float (&v)[WG4_CHUNK / 2], const float* __restrict__ P, int ld, int idx,
                                            int extent, int k0, int kend, int h, int vec) {
    const bool ok = idx < extent;
    if (KC) {
        const float* __restrict__ row = P + (size_t)(ok ? idx : 0) * ld;
#pragma unroll
        for (int q = 0; q < WG4_CHUNK / 4; ++q) {
            const int k4 = k0 + 4 * q;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                if (vec && k4 + 3 < kend) {
                    t = *reinterpret_cast<const float4*>(row + k4);
                } else {
                    if (k4 < kend) t.x = row[k4];
                    if (k4 + 1 < kend) t.y = row[k4 + 1];
                    if (k4 + 2 < kend) t.z = row[k4 + 2];
                    if (k4 + 3 < kend) t.w = row[k4 + 3];
                }
            }
            v[2 * q] = h ? t.y : t.x;
            v[2 * q + 1] = h ? t.w : t.z;
        }
    } else {
#pragma unroll
        for (int s = 0; s < WG4_CHUNK / 2; ++s) {
            const int k = k0 + 2 * s + h;
            v[s] = (ok && k < kend) ? P[(size_t)k * ld + idx] : 0.f;
        }
    }
}

// K-contiguous operands (activations X[m][k], weights W[n][k]) through LDS: the direct form above has every lane walk its
// own row -- 32 rows at a >= 1 KB pitch per load instruction, 32 cache lines each, which measured +5 us per such operand
// on a 128 x 256 x 256 layer.  Here a wave fetches its 32 x 64 panel with fully coalesced 16-byte loads (four rows of
// 256 B per instruction), parks it in a wave-private LDS panel (pitch 68 floats: the 16 lanes of a ds_read_b128 phase hit
// disjoint banks) and reads its MFMA operands back row-wise.  Wave-private => a wave barrier, no workgroup barrier.
constexpr int WG4_PITCH = WG4_CHUNK + 4;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// issue the coalesced global loads of panel rows [idx0, idx0 + 32) x k [k0, k0 + 64): instruction q covers rows 4q .. 4q+3
__device__ __forceinline__ void panel_fetch(float4 (&t)[8], const float* __restrict__ P, int ld, int idx0, int extent,
                                            int k0, int kend, int lane) {
    const int kk = k0 + (lane & 15) * 4;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = idx0 + 4 * q + (lane >> 4);
        t[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < extent) {
            const float* __restrict__ src = P + (size_t)row * ld + kk;
            if (kk + 3 < kend) {
                t[q] = *reinterpret_cast<const float4*>(src);
            } else {
                if (kk < kend) t[q].x = src[0];
                if (kk + 1 < kend) t[q].y = src[1];
                if (kk + 2 < kend) t[q].z = src[2];
            }
        }
    }
}

__device__ __forceinline__ void panel_transpose(float (&v)[WG4_CHUNK / 2], const float4 (&t)[8], float* __restrict__ panel,
                                                int lane) {
    wave_lds_sync();                                           // earlier reads of the panel are done
#pragma unroll
    for (int q = 0; q < 8; ++q)
        *reinterpret_cast<float4*>(panel + (4 * q + (lane >> 4)) * WG4_PITCH + (lane & 15) * 4) = t[q];
    wave_lds_sync();
    const int h = lane >> 5, i = lane & 31;
#pragma unroll
    for (int q = 0; q < WG4_CHUNK / 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(panel + i * WG4_PITCH + 4 * q);
        v[2 * q] = h ? x.y : x.x;
        v[2 * q + 1] = h ? x.w : x.z;
    }
}

// The Adam step of ONE parameter as torch's single-tensor implementation evaluates it (torch/optim/adam.py: exp_avg.lerp_,
// exp_avg_sq.mul_().addcmul_(), denom = sqrt / bias_correction2_sqrt + eps, addcdiv_ with -lr / bias_correction1): every
// optimiser kernel of the actor-critic rows goes through this one function, so a fused and a separate step give the same bits.
struct AdamScalars {
    float neg_step_size, bc2_sqrt, one_minus_b1, b2, one_minus_b2, eps;
};
__device__ __forceinline__ float adam_element(const AdamScalars& c, float p, float gr, float& m, float& v) {
    m = fmaf(c.one_minus_b1, __fsub_rn(gr, m), m);
    v = __fadd_rn(__fmul_rn(v, c.b2), __fmul_rn(__fmul_rn(c.one_minus_b2, gr), gr));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), c.bc2_sqrt), c.eps);
    return __fadd_rn(p, __fmul_rn(c.neg_step_size, __fdiv_rn(m, denom)));
}
__device__ __forceinline__ AdamScalars adam_scalars(int t, double lr, double b1, double b2, float eps) {
    AdamScalars c;
    c.neg_step_size = (float)(-(lr / (1.0 - pow(b1, (double)t))));
    c.bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)t));
    c.one_minus_b1 = (float)(1.0 - b1); c.b2 = (float)b2; c.one_minus_b2 = (float)(1.0 - b2); c.eps = eps;
    return c;
}

// Adam in the epilogue of a weight-gradient tile (gemm_wave4_tile<..., ADAM = true>): the tile's sums ARE the finished gradient
// entries (the four waves split the batch rows inside the workgroup, no split over workgroups), so the step, the K-major shadow
// copy the forward chains stream and the Polyak average of the target net are applied where the gradient is produced -- the
// gradient itself never goes to memory.  Pointers are those of this net's flat parameter block.
struct AdamTile {
    float* params; float* exp_avg; float* exp_avg_sq;
    float* wt;              // K-major shadow copy of this net, or NULL
    float* target;          // Polyak target of this net, or NULL
    long long offW, offB;   // this layer's weight block ([M][N] dense, nn.Linear layout) and bias block
    const float* corr;      // {-lr / bias_correction1, sqrt(bias_correction2)} of this learner, left by an earlier kernel of the
                            // update (the two fp64 pow()s are ~2 us of one wave: not in this epilogue), or NULL: from `step`
    int step;               // Adam step number of this update (>= 1)
    double lr, b1, b2;
    float eps, tau;
};
__device__ __forceinline__ void adam_tile_apply(const AdamTile& a, const AdamScalars& c, long long e, long long e_t, float gr) {
    float m = a.exp_avg[e], v = a.exp_avg_sq[e];
    const float np_ = adam_element(c, a.params[e], gr, m, v);
    a.params[e] = np_;
    a.exp_avg[e] = m;
    a.exp_avg_sq[e] = v;
    if (a.wt != nullptr) a.wt[e_t] = np_;
    if (a.target != nullptr) {
        if (a.tau == 1.0f) a.target[e] = np_;
        else a.target[e] = __fadd_rn(__fmul_rn(a.target[e], 1.0f - a.tau), __fmul_rn(a.tau, np_));
    }
}

template <bool A_KC, bool B_KC, int EPI, bool ADAM = false>
__device__ __forceinline__ void gemm_wave4_tile(const GemmProblem& g, int tile_m, int tile_n, const AdamTile* ad = nullptr) {
    __shared__ float s_red[ADAM ? 4 : 3][16][64];
    __shared__ float s_col[ADAM ? 4 : 3][32];
    __shared__ __attribute__((aligned(16))) float s_panel[(A_KC || B_KC) ? 4 : 1][(A_KC || B_KC) ? 32 * WG4_PITCH : 4];
    const int lane = lane_id(), wave = wave_id();
    const int h = lane >> 5, i = lane & 31;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int kper = ((g.K + 3) / 4 + 3) / 4 * 4;            // per-wave k range, multiple of 4 (16-byte loads stay aligned)
    const int kbeg = wave * kper, kend = min(g.K, kbeg + kper);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float colsum = 0.f;
    const bool do_colsum = (g.colsum != nullptr) && (tile_n == 0);
    const bool a_lds = A_KC && g.a_vec, b_lds = B_KC && g.b_vec;          // wave-uniform
    for (int k0 = kbeg; k0 < kend; k0 += WG4_CHUNK) {
        float a[WG4_CHUNK / 2], b[WG4_CHUNK / 2];
        float4 ta[8], tb[8];
        // all global loads of the stage are issued before the first LDS round trip
        if (a_lds) panel_fetch(ta, g.A, g.lda, m0, g.M, k0, kend, lane);
        if (b_lds) panel_fetch(tb, g.B, g.ldb, n0, g.N, k0, kend, lane);
        if (!a_lds) wave_load64<A_KC>(a, g.A, g.lda, m0 + i, g.M, k0, kend, h, g.a_vec);
        if (!b_lds) wave_load64<B_KC>(b, g.B, g.ldb, n0 + i, g.N, k0, kend, h, g.b_vec);
        if (a_lds) panel_transpose(a, ta, s_panel[(A_KC || B_KC) ? wave : 0], lane);
        if (b_lds) panel_transpose(b, tb, s_panel[(A_KC || B_KC) ? wave : 0], lane);
#pragma unroll
        for (int s = 0; s < WG4_CHUNK / 2; ++s) {
            acc = mfma32(a[s], b[s], acc);
            if (do_colsum) { const float odd = __shfl_xor(a[s], 32); colsum += a[s]; colsum += odd; }
        }
    }
    if constexpr (ADAM) {
        // every wave leaves its partial tile in LDS and finishes a quarter of the entries: the sums in the wave order of the plain
        // epilogue below -- ((wave 0 + wave 1) + wave 2) + wave 3 --, then the optimiser step of those entries (three loads, the
        // step, up to five stores per entry: spread over the four waves instead of serial in one)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[r];
        if (do_colsum && h == 0) s_col[wave][i] = colsum;
        __syncthreads();
        AdamScalars c;
        if (ad->corr != nullptr) {
            c.neg_step_size = ad->corr[0]; c.bc2_sqrt = ad->corr[1];
            c.one_minus_b1 = (float)(1.0 - ad->b1); c.b2 = (float)ad->b2; c.one_minus_b2 = (float)(1.0 - ad->b2); c.eps = ad->eps;
        } else {
            c = adam_scalars(ad->step, ad->lr, ad->b1, ad->b2, ad->eps);
        }
        const int col = n0 + i;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr;
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float gr = ((s_red[0][r][lane] + s_red[1][r][lane]) + s_red[2][r][lane]) + s_red[3][r][lane];
            if (row < g.M && col < g.N)
                adam_tile_apply(*ad, c, ad->offW + (long long)row * g.N + col, ad->offW + (long long)col * g.M + row, gr);
        }
        if (do_colsum && wave == 3 && h == 0 && m0 + i < g.M) {
            const long long e = ad->offB + m0 + i;
            adam_tile_apply(*ad, c, e, e, ((s_col[0][i] + s_col[1][i]) + s_col[2][i]) + s_col[3][i]);
        }
        return;
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[wave - 1][r][lane] = acc[r];
        if (do_colsum && h == 0) s_col[wave - 1][i] = colsum;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += s_red[w][r][lane];
    const int col = n0 + i;
    float bias = 0.f;
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && col < g.N) bias = g.bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M && col < g.N) {
            float v = acc[r];
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) v += bias;
            if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
            if (EPI == EPI_RELU_MASK) v = (g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v : 0.f;
            g.C[(size_t)row * g.ldc + col] = v;
        }
    }
    if (do_colsum && h == 0 && m0 + i < g.M) g.colsum[m0 + i] = ((colsum + s_col[0][i]) + s_col[1][i]) + s_col[2][i];
}

}  // namespace morl
