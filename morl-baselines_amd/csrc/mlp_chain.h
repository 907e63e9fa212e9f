// Layer-fused MLP engine: a workgroup carries a 64-row tile through a whole chain of dense layers with the
// activations RESIDENT IN LDS -- they never round-trip through HBM between layers (gfx950, wave64).
//
//   step s:   out_s[64][N_s] = epilogue_s( act[64][K_s] @ Bmat_s[K_s][N_s] ),   act <- out_s
//
// Used three ways (same kernel, different step tables):
//   * no-grad forward      Q(s', w) slabs:  Bmat = W_l^T (pre-transposed copy), epilogue bias+ReLU, only Q leaves
//   * training forward     same, but every hidden activation is also saved to HBM for the backward pass
//   * backward (dX chain)  Bmat = W_l as stored ([out][in] is already K-major for g_l @ W_l), epilogue = ReLU mask
//                          from the saved activation; every g_l is written out for the weight-gradient GEMM
//
// Tiling: 256 threads = 4 waves; wave w owns output columns [64w, 64w+64) as 2x2 v_mfma_f32_32x32x2_f32 tiles
// (64 accumulator registers; exact fp32).  LDS: activations K-major sAct[k][m] (stride 65 -> the transposed
// epilogue stores and the MFMA operand reads are both bank-conflict free) = 66.6 KB, plus a double-buffered
// 32 x 256 weight chunk (64 KB) streamed from L2 with 16-byte loads / 16-byte LDS stores, prefetched into
// registers under the previous chunk's MFMAs.  One barrier per chunk.  1 workgroup per CU (133 KB LDS).
// Widths up to 256; all Bmat row strides must be multiples of 4 floats (host checks, else the per-layer GEMM
// path is used).
//
// Roofline: fp32 MFMA; per 64-row tile sum_s 2*64*K_s*N_s flop, weights re-read from L2 (per-XCD resident),
// algorithmic HBM bytes = inputs + outputs only.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

constexpr int CH_TM = 64;           // rows per workgroup
constexpr int CH_LDM = 65;          // sAct row stride (floats)
constexpr int CH_MAXW = 256;        // widest layer
constexpr int CH_BK = 32;           // K chunk
constexpr int CH_THREADS = 256;

struct ChainStep {
    const float* Bmat;   // [K][ldb] K-major operand (row k contiguous over n), zero beyond column N
    const float* bias;   // [N] or NULL
    const float* mask;   // [rows][ldmask] or NULL: result kept where mask > 0 (ReLU backward)
    float* out;          // [rows][ldout] or NULL: global copy of this step's output
    int K, N, ldb, ldmask, ldout;
    int relu;
};

struct ChainArgs {
    ChainStep step[MORL_MAX_LAYERS];
    int n_steps;
    int rows;
    // input assembly
    int in_mode;            // 0: cat(obs[b], weights[k]) with row -> (b, k) by row_order ; 1: dense matrix src[rows][ldsrc]
    const float* obs;       // [B][D]
    const float* weights;   // [W][R]  (row_order 2: [rows][R], paired with obs rows)
    int B, W, D, R, row_order;
    const float* src;       // in_mode 1
    int ldsrc, K0;
};

__global__ __launch_bounds__(CH_THREADS) void mlp_chain_kernel(ChainArgs p) {
    __shared__ __attribute__((aligned(16))) float sAct[CH_MAXW * CH_LDM];
    __shared__ __attribute__((aligned(16))) float sB[2][CH_BK * CH_MAXW];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int h = lane >> 5, i = lane & 31;
    const int row0 = (int)blockIdx.x * CH_TM;

    // ---- input tile -> sAct[k][m] ------------------------------------------------------------
    {
        const int K0 = (p.in_mode == 0) ? (p.D + p.R) : p.K0;
        const int K0pad = (K0 + 1) & ~1;
        for (int e = tid; e < K0pad * CH_TM; e += CH_THREADS) {
            const int k = e / CH_TM, m = e % CH_TM;
            const int row = row0 + m;
            float v = 0.f;
            if (row < p.rows && k < K0) {
                if (p.in_mode == 0) {
                    int b, w;
                    if (p.row_order == 0) { b = row / p.W; w = row % p.W; }
                    else if (p.row_order == 1) { w = row / p.B; b = row % p.B; }
                    else { b = row; w = row; }
                    v = (k < p.D) ? p.obs[(size_t)b * p.D + k] : p.weights[(size_t)w * p.R + (k - p.D)];
                } else {
                    v = p.src[(size_t)row * p.ldsrc + k];
                }
            }
            sAct[k * CH_LDM + m] = v;
        }
    }

    for (int s = 0; s < p.n_steps; ++s) {
        const ChainStep& st = p.step[s];
        const int K = st.K, N = st.N;
        const int Kpad = (K + 1) & ~1;
        const int n_tiles = (N + 31) >> 5;
        const int my_tiles = max(0, min(2, n_tiles - 2 * wave));   // 32-column tiles owned by this wave
        const int nbase = wave * 64;

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        // weight chunk staging: 32 rows x 256 cols = 2048 float4, 8 per thread
        float4 stage[8];
        auto load_chunk = [&](int k0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int f = tid + q * CH_THREADS;
                const int kr = k0 + (f >> 6), c = (f & 63) << 2;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kr < K && c < st.ldb) v = *reinterpret_cast<const float4*>(st.Bmat + (size_t)kr * st.ldb + c);
                if (c + 0 >= N) v.x = 0.f;
                if (c + 1 >= N) v.y = 0.f;
                if (c + 2 >= N) v.z = 0.f;
                if (c + 3 >= N) v.w = 0.f;
                stage[q] = v;
            }
        };
        auto store_chunk = [&](int buf) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int f = tid + q * CH_THREADS;
                *reinterpret_cast<float4*>(&sB[buf][(f >> 6) * CH_MAXW + ((f & 63) << 2)]) = stage[q];
            }
        };

        load_chunk(0);
        store_chunk(0);
        __syncthreads();   // also orders the previous step's sAct writes (or the input assembly) before the reads below
        int buf = 0;
        for (int k0 = 0; k0 < Kpad; k0 += CH_BK) {
            const bool more = (k0 + CH_BK) < Kpad;
            if (more) load_chunk(k0 + CH_BK);
            if (my_tiles > 0) {
                const int kc = min(CH_BK, Kpad - k0);
                const float* pa = sAct + (k0 + h) * CH_LDM + i;
                const float* pb = &sB[buf][h * CH_MAXW + nbase + i];
                if (my_tiles == 2) {
                    for (int kk = 0; kk < kc; kk += 2) {
                        const float a0 = pa[kk * CH_LDM], a1 = pa[kk * CH_LDM + 32];
                        const float b0 = pb[kk * CH_MAXW], b1 = pb[kk * CH_MAXW + 32];
                        acc[0][0] = mfma32(a0, b0, acc[0][0]);
                        acc[0][1] = mfma32(a0, b1, acc[0][1]);
                        acc[1][0] = mfma32(a1, b0, acc[1][0]);
                        acc[1][1] = mfma32(a1, b1, acc[1][1]);
                    }
                } else {
                    for (int kk = 0; kk < kc; kk += 2) {
                        const float a0 = pa[kk * CH_LDM], a1 = pa[kk * CH_LDM + 32];
                        const float b0 = pb[kk * CH_MAXW];
                        acc[0][0] = mfma32(a0, b0, acc[0][0]);
                        acc[1][0] = mfma32(a1, b0, acc[1][0]);
                    }
                }
            }
            if (more) store_chunk(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        // ---- epilogue: every wave is past its last read of sAct (barrier above) --------------------
        const bool feed_next = (s + 1 < p.n_steps);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            if (tn >= my_tiles) continue;
            const int col = nbase + tn * 32 + i;
            const bool col_ok = col < N;
            const float bias = (st.bias != nullptr && col_ok) ? st.bias[col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int row = row0 + m;
                    const bool ok = col_ok && row < p.rows;
                    float v = acc[tm][tn][r] + bias;
                    if (st.relu) v = fmaxf(v, 0.f);
                    if (st.mask != nullptr) v = (ok && st.mask[(size_t)row * st.ldmask + col] > 0.f) ? v : 0.f;
                    if (!ok) v = 0.f;
                    if (feed_next) sAct[col * CH_LDM + m] = v;
                    if (st.out != nullptr && ok) st.out[(size_t)row * st.ldout + col] = v;
                }
            }
        }
        // columns of the padded K range of the next step that no wave owns are already zero only if N is even or a
        // tile covers them: tiles span whole multiples of 32 >= N, so col = N (when N is odd) is inside a tile and was
        // written as 0 above.
    }
}

// W_l [N][K] (nn.Linear layout) -> Wt_l [K][ldn] with ldn = round_up(N, 4), zero padded: the K-major copy the
// forward chain streams.  All layers in one launch.
struct TransposeArgs {
    long long src_off[MORL_MAX_LAYERS];
    long long dst_off[MORL_MAX_LAYERS];
    long long elem_start[MORL_MAX_LAYERS + 1];   // prefix sums of K*ldn
    int K[MORL_MAX_LAYERS], N[MORL_MAX_LAYERS], ldn[MORL_MAX_LAYERS];
    int n;
};

__global__ __launch_bounds__(256) void transpose_params_kernel(const float* __restrict__ params,
                                                               float* __restrict__ wt, TransposeArgs t) {
    const long long total = t.elem_start[t.n];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < t.n && e >= t.elem_start[l + 1]) ++l;
        const long long loc = e - t.elem_start[l];
        const int k = (int)(loc / t.ldn[l]), n = (int)(loc % t.ldn[l]);
        wt[t.dst_off[l] + loc] = (n < t.N[l]) ? params[t.src_off[l] + (long long)n * t.K[l] + k] : 0.f;
    }
}

}  // namespace morl
