// Exact-fp32 MFMA GEMM tile engine for the Q-network layers (gfx950, wave64).
//
//   C[M][N] = epilogue( sum_k A(m,k) * B(k,n) )
//
// One 256-thread workgroup (4 waves as 2x2) owns a 128x128 tile of C; each wave owns 64x64 = 2x2
// v_mfma_f32_32x32x2_f32 tiles (64 accumulator registers).  Operands are staged global -> registers ->
// LDS in K-chunks of 32 and always stored K-MAJOR in LDS (sA[k][m], sB[k][n]) so that the MFMA operand
// fetch "lane (i, h) reads element (m0+i, k+h)" is a conflict-free ds_read_b32 over 32 consecutive
// floats per half-wave, whatever the global layout was:
//   *_KC = 1: operand is K-contiguous in global memory (activations X[m][k], weights W[n][k]);
//             16-B global loads along k, transposed 4-B LDS stores, row stride 129 (4*129 % 32 == 4 ->
//             the 8 k-quads x 4 rows of a half-wave hit 32 distinct banks).
//   *_KC = 0: operand is M/N-contiguous (dY^T for dW, W for dX); 16-B global loads and 16-B LDS stores,
//             row stride 132 (16-B aligned rows).
// The next chunk's global loads are issued before the current chunk's MFMAs (register prefetch).
// Roofline: fp32 MFMA, 2*M*N*K flop; algorithmic bytes 4*(M*K + K*N + M*N).
#pragma once
#include "morl_device.h"

namespace morl {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32, GEMM_THREADS = 256;
constexpr int GEMM_LD_T = 129;  // K-contiguous source: transposed scalar LDS stores
constexpr int GEMM_LD_N = 132;  // M/N-contiguous source: float4 LDS stores

enum GemmEpilogue {
    EPI_STORE = 0,      // C = acc
    EPI_BIAS = 1,       // C = acc + bias[n]
    EPI_BIAS_RELU = 2,  // C = max(acc + bias[n], 0)
    EPI_RELU_MASK = 3   // C = mask[m][n] > 0 ? acc : 0      (ReLU backward through the saved activation)
};

struct GemmProblem {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // [N]            (EPI_BIAS*)
    const float* mask;   // [M][ldmask]    (EPI_RELU_MASK)
    float* colsum;       // optional: colsum[z*colsum_stride + m] = sum over this split's k of A(m,k) (bias grad)
    int M, N, K;
    int lda, ldb, ldc, ldmask;
    int a_vec, b_vec;    // 16-byte global loads legal for A / B
    int k_per_split;     // K range of split z is [z*k_per_split, min(K, (z+1)*k_per_split))
    long long c_split_stride;       // C += z * c_split_stride (split-K slabs)
    long long colsum_stride;
    int tiles_m, tiles_n;
};

template <bool KC>
struct Stage {
    float4 v[4];
};

// global -> registers for one 128 x 32 operand chunk. rows = M (or N) extent, kend = exclusive K bound.
template <bool KC>
__device__ __forceinline__ void load_chunk(Stage<KC>& s, const float* __restrict__ P, int ld, int row0, int rows,
                                           int k0, int kend, int vec) {
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int f = tid + p * GEMM_THREADS;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            const int r = row0 + (f >> 3), k = k0 + ((f & 7) << 2);
            if (r < rows) {
                const float* src = P + (size_t)r * ld + k;
                if (vec && k + 3 < kend) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (k < kend) v.x = src[0];
                    if (k + 1 < kend) v.y = src[1];
                    if (k + 2 < kend) v.z = src[2];
                    if (k + 3 < kend) v.w = src[3];
                }
            }
        } else {
            const int k = k0 + (f >> 5), r = row0 + ((f & 31) << 2);
            if (k < kend) {
                const float* src = P + (size_t)k * ld + r;
                if (vec && r + 3 < rows) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (r < rows) v.x = src[0];
                    if (r + 1 < rows) v.y = src[1];
                    if (r + 2 < rows) v.z = src[2];
                    if (r + 3 < rows) v.w = src[3];
                }
            }
        }
        s.v[p] = v;
    }
}

// registers -> LDS (K-major image)
template <bool KC>
__device__ __forceinline__ void store_chunk(const Stage<KC>& s, float* __restrict__ sm) {
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int f = tid + p * GEMM_THREADS;
        const float4 v = s.v[p];
        if (KC) {
            const int r = f >> 3, k = (f & 7) << 2;
            sm[(k + 0) * GEMM_LD_T + r] = v.x;
            sm[(k + 1) * GEMM_LD_T + r] = v.y;
            sm[(k + 2) * GEMM_LD_T + r] = v.z;
            sm[(k + 3) * GEMM_LD_T + r] = v.w;
        } else {
            const int k = f >> 5, r = (f & 31) << 2;
            *reinterpret_cast<float4*>(sm + k * GEMM_LD_N + r) = v;
        }
    }
}

template <bool A_KC, bool B_KC, int EPI>
__device__ __forceinline__ void gemm_tile(const GemmProblem& g, int tile_m, int tile_n, int split) {
    __shared__ __attribute__((aligned(16))) float sA[GEMM_BK * GEMM_LD_N];
    __shared__ __attribute__((aligned(16))) float sB[GEMM_BK * GEMM_LD_N];
    constexpr int LDA = A_KC ? GEMM_LD_T : GEMM_LD_N;
    constexpr int LDB = B_KC ? GEMM_LD_T : GEMM_LD_N;

    const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int lane = lane_id(), wave = wave_id();
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float colsum = 0.f;
    const bool do_colsum = (g.colsum != nullptr) && (tile_n == 0);

    Stage<A_KC> ra;
    Stage<B_KC> rb;
    if (kbeg < kend) {
        load_chunk<A_KC>(ra, g.A, g.lda, m0, g.M, kbeg, kend, g.a_vec);
        load_chunk<B_KC>(rb, g.B, g.ldb, n0, g.N, kbeg, kend, g.b_vec);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
        store_chunk<A_KC>(ra, sA);
        store_chunk<B_KC>(rb, sB);
        __syncthreads();
        if (k0 + GEMM_BK < kend) {
            load_chunk<A_KC>(ra, g.A, g.lda, m0, g.M, k0 + GEMM_BK, kend, g.a_vec);
            load_chunk<B_KC>(rb, g.B, g.ldb, n0, g.N, k0 + GEMM_BK, kend, g.b_vec);
        }
        const float* pa = sA + h * LDA + wm * 64 + i;
        const float* pb = sB + h * LDB + wn * 64 + i;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK; kk += 2) {
            const float a0 = pa[kk * LDA], a1 = pa[kk * LDA + 32];
            const float b0 = pb[kk * LDB], b1 = pb[kk * LDB + 32];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
        }
        if (do_colsum && threadIdx.x < GEMM_BM) {
#pragma unroll 8
            for (int kk = 0; kk < GEMM_BK; ++kk) colsum += sA[kk * LDA + (int)threadIdx.x];
        }
        __syncthreads();
    }

    float* __restrict__ C = g.C + (size_t)split * g.c_split_stride;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wn * 64 + tn * 32 + i;
            float bias = 0.f;
            if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && col < g.N) bias = g.bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < g.M && col < g.N) {
                    float v = acc[tm][tn][r];
                    if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) v += bias;
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                    if (EPI == EPI_RELU_MASK) v = (g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v : 0.f;
                    C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    if (do_colsum && threadIdx.x < GEMM_BM && m0 + (int)threadIdx.x < g.M)
        g.colsum[(size_t)split * g.colsum_stride + m0 + (int)threadIdx.x] = colsum;
}

// ----------------------------------------------------------------------------------------------------------------
// Double-buffered variant for the weight-gradient GEMM (both operands M/N-contiguous, i.e. straight 16-byte copies
// into the K-major LDS image): one barrier per 32-deep chunk, the next chunk's global loads in flight under the MFMAs
// and written to the OTHER LDS buffer afterwards, operand reads of k-pair j+1 issued before the MFMAs of k-pair j.
// 67.6 KB of LDS -> two workgroups per CU overlap each other's barriers and epilogues.
// ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gemm_tile_tn_db(const GemmProblem& g, int tile_m, int tile_n, int split) {
    __shared__ __attribute__((aligned(16))) float sA[2][GEMM_BK * GEMM_LD_N];
    __shared__ __attribute__((aligned(16))) float sB[2][GEMM_BK * GEMM_LD_N];
    constexpr int LD = GEMM_LD_N;
    const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int lane = lane_id(), wave = wave_id();
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float colsum = 0.f;
    const bool do_colsum = (g.colsum != nullptr) && (tile_n == 0);

    Stage<false> ra, rb;
    if (kbeg < kend) {
        load_chunk<false>(ra, g.A, g.lda, m0, g.M, kbeg, kend, g.a_vec);
        load_chunk<false>(rb, g.B, g.ldb, n0, g.N, kbeg, kend, g.b_vec);
        store_chunk<false>(ra, sA[0]);
        store_chunk<false>(rb, sB[0]);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
        const bool more = k0 + GEMM_BK < kend;
        if (more) {
            load_chunk<false>(ra, g.A, g.lda, m0, g.M, k0 + GEMM_BK, kend, g.a_vec);
            load_chunk<false>(rb, g.B, g.ldb, n0, g.N, k0 + GEMM_BK, kend, g.b_vec);
        }
        const float* pa = sA[buf] + h * LD + wm * 64 + i;
        const float* pb = sB[buf] + h * LD + wn * 64 + i;
        float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
#pragma unroll
        for (int kk = 0; kk < GEMM_BK; kk += 2) {
            const int kn = (kk + 2 < GEMM_BK) ? kk + 2 : kk;
            const float na0 = pa[kn * LD], na1 = pa[kn * LD + 32];
            const float nb0 = pb[kn * LD], nb1 = pb[kn * LD + 32];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (do_colsum && threadIdx.x < GEMM_BM) {
#pragma unroll 8
            for (int kk = 0; kk < GEMM_BK; ++kk) colsum += sA[buf][kk * LD + (int)threadIdx.x];
        }
        if (more) {
            store_chunk<false>(ra, sA[buf ^ 1]);
            store_chunk<false>(rb, sB[buf ^ 1]);
        }
        __syncthreads();
        buf ^= 1;
    }
    float* __restrict__ C = g.C + (size_t)split * g.c_split_stride;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wn * 64 + tn * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < g.M && col < g.N) C[(size_t)row * g.ldc + col] = acc[tm][tn][r];
            }
        }
    if (do_colsum && threadIdx.x < GEMM_BM && m0 + (int)threadIdx.x < g.M)
        g.colsum[(size_t)split * g.colsum_stride + m0 + (int)threadIdx.x] = colsum;
}

}  // namespace morl
