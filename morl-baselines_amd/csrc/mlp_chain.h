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
// (64 accumulator registers; exact fp32).  Steps with N <= 32 (the Q head) instead split the contraction over the
// four waves (8 slices of every 32-deep chunk) and reduce the partial tiles through LDS in a fixed order; both
// shapes run the SAME inner loop (only operand offsets / trip counts differ) so the accumulators never move.
// LDS: activations K-major sAct[k][m] (stride 65 -> the transposed epilogue stores and the MFMA operand reads are
// both bank-conflict free) = 66.6 KB, plus a double-buffered 32 x 256 weight chunk (64 KB).  The weight stream is
// one flat sequence of chunks over all steps: the chunk after the one being multiplied (possibly the first chunk of
// the NEXT layer) is already in flight in registers (16-byte loads, scalar-base + per-thread-offset addressing, no
// per-element address math) and is written to the other LDS buffer after the MFMAs; one barrier per chunk.
// 1 workgroup per CU (133 KB LDS).  Widths up to 256; Bmat row strides must be multiples of 4 floats and Bmat must
// be zero in columns [N, ldb) (host guarantees both, else the per-layer GEMM path is used).
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
    const float* Bmat;   // [K][ldb] K-major operand (row k contiguous over n), zero in columns [N, ldb)
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
    long long* prof;        // optional [gridDim.x][8] cycle counters (development builds of the probe only)
};

struct ChainStage {
    float4 v[8];
};

// global -> registers: rows k0 + (tid>>6) + 4q, 16 bytes at column 4*(tid&63).  Branch-free on the full-chunk
// path: threads whose column lies beyond the row stride re-read column 0 (their LDS columns only ever feed output
// columns >= N, which the epilogue discards); rows beyond K are zero-filled (they meet zero activations, and
// 0 * garbage must not produce NaN).
__device__ __forceinline__ void chain_load(ChainStage& s, const ChainStep& st, int k0) {
    const int tid = (int)threadIdx.x;
    const int r0 = tid >> 6;
    int c = (tid & 63) << 2;
    if (c >= st.ldb) c = 0;
    const float* base = st.Bmat + (size_t)k0 * st.ldb;           // wave-uniform
    const int toff = r0 * st.ldb + c;                             // per-thread, step-constant
    if (k0 + CH_BK <= st.K) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s.v[q] = *reinterpret_cast<const float4*>(base + (size_t)(4 * q) * st.ldb + toff);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kr = min(k0 + r0 + 4 * q, st.K - 1) - k0 - r0;   // clamp the row, then zero it
            float4 v = *reinterpret_cast<const float4*>(base + (ptrdiff_t)kr * st.ldb + toff);
            if (k0 + r0 + 4 * q >= st.K) v = make_float4(0.f, 0.f, 0.f, 0.f);
            s.v[q] = v;
        }
    }
}

__device__ __forceinline__ void chain_store(const ChainStage& s, float* __restrict__ sb) {
    const int tid = (int)threadIdx.x;
    float* dst = sb + (tid >> 6) * CH_MAXW + ((tid & 63) << 2);
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(dst + q * 4 * CH_MAXW) = s.v[q];
}

// LDS-DMA variant of the weight stream (global_load_lds_dwordx4): each wave moves 8 whole 1-KiB rows of the chunk
// straight from L2 into the LDS image -- no staging VGPRs, no ds_write pass.  The LDS destination of one instruction
// is wave-uniform base + 16*lane, which is exactly one 256-float row.  Only for full chunks of full-width rows.
__device__ __forceinline__ bool chain_dma_ok(const ChainStep& st, int k0) {
    return st.ldb == CH_MAXW && k0 + CH_BK <= st.K;
}
__device__ __forceinline__ void chain_dma(const ChainStep& st, int k0, float* __restrict__ sb) {
    const int lane = lane_id(), wave = wave_id();
    const float* g = st.Bmat + (size_t)(k0 + wave * 8) * CH_MAXW + lane * 4;
    float* l = sb + wave * 8 * CH_MAXW;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + q * CH_MAXW),
                                         (__attribute__((address_space(3))) void*)(l + q * CH_MAXW), 16, 0, 0);
}

template <bool PROF, bool DMA>
__device__ __forceinline__ void mlp_chain_body(const ChainArgs& p) {
    __shared__ __attribute__((aligned(16))) float sAct[CH_MAXW * CH_LDM];
    __shared__ __attribute__((aligned(16))) float sB[2][CH_BK * CH_MAXW];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int h = lane >> 5, i = lane & 31;
    const int row0 = (int)blockIdx.x * CH_TM;
    long long t_in = 0, t_mfma = 0, t_stage = 0, t_epi = 0, t0 = 0;
    if (PROF) t0 = clock64();

    ChainStage stage;
    chain_load(stage, p.step[0], 0);   // the weight stream starts before the input tile is assembled

    // ---- input tile -> sAct[k][m] ------------------------------------------------------------
    {
        const int K0 = (p.in_mode == 0) ? (p.D + p.R) : p.K0;
        const int K0pad = min(CH_MAXW, (K0 + 31) & ~31);   // whole 32-row chunk defined (narrow steps read all of it)
        const int m = tid & (CH_TM - 1);
        const int row = row0 + m;
        int b = row, w = row;
        if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; w = row - b * p.W; }
            else if (p.row_order == 1) { w = row / p.B; b = row - w * p.B; }
        }
        const bool row_ok = row < p.rows;
        for (int k = tid >> 6; k < K0pad; k += CH_THREADS / CH_TM) {
            float v = 0.f;
            if (row_ok && k < K0) {
                if (p.in_mode == 0) v = (k < p.D) ? p.obs[(size_t)b * p.D + k] : p.weights[(size_t)w * p.R + (k - p.D)];
                else v = p.src[(size_t)row * p.ldsrc + k];
            }
            sAct[k * CH_LDM + m] = v;
        }
    }
    chain_store(stage, sB[0]);
    __syncthreads();
    if (PROF) { const long long t = clock64(); t_in += t - t0; t0 = t; }

    int buf = 0;
    for (int s = 0; s < p.n_steps; ++s) {
        const ChainStep& st = p.step[s];
        const int K = st.K, N = st.N;
        const int Kpad = (K + 1) & ~1;
        // Two shapes, ONE inner loop (so the 64 accumulators never move):
        //   wide   (N > 32):  acc[tm][tn] = A[tm rows][k] . B[k][64*wave + 32*tn + i]        k over the whole chunk
        //   narrow (N <= 32): the 4 waves x 2 "tn" slots split each 32-deep chunk into 8 slices of 4; acc[tm][tn]
        //                     holds the partial product of slice 2*wave+tn for output columns 0..31
        const bool narrow = (N <= 32);
        const int kk_begin = narrow ? wave * 8 : 0;
        const int a_off1 = narrow ? 4 * CH_LDM : 0;              // A offset of the tn = 1 operand (floats)
        const int b_off1 = narrow ? 4 * CH_MAXW : 32;            // B offset of the tn = 1 operand (floats)
        const int nbase = narrow ? 0 : wave * 64;

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        for (int k0 = 0; k0 < Kpad; k0 += CH_BK) {
            // next chunk of the flat stream: same step, or the first chunk of the next step
            const bool more_here = (k0 + CH_BK) < Kpad;
            const bool more = more_here || (s + 1 < p.n_steps);
            const ChainStep& nst = more_here ? st : p.step[more ? s + 1 : s];
            const int nk0 = more_here ? k0 + CH_BK : 0;
            // sB[buf ^ 1] is free: its last readers passed the barrier that ended the previous chunk
            const bool dma = DMA && more && chain_dma_ok(nst, nk0);
            if (dma) chain_dma(nst, nk0, sB[buf ^ 1]);
            else if (more) chain_load(stage, nst, nk0);
            if (PROF) { const long long t = clock64(); t_stage += t - t0; t0 = t; }
            const int kc = min(CH_BK, Kpad - k0);
            // narrow: slice pair [kk_begin, kk_begin+4) and [kk_begin+4, kk_begin+8); rows >= Kpad of both operands
            // are zero (sAct rows up to the next multiple of 32 are written by every epilogue / the input assembly)
            const int kk_end = narrow ? min(kc, kk_begin + 4) : kc;
            const float* pa = sAct + (k0 + h) * CH_LDM + i;
            const float* pb = &sB[buf][h * CH_MAXW + nbase + i];
            // Software-pipelined by hand with two named operand sets: the LDS reads of k-pair j+1 are issued before the
            // four MFMAs of k-pair j, so the ~100-cycle ds_read latency hides under the 256 MFMA cycles (one wave per
            // SIMD: there is no other wave to hide it).  No register rotation -> no v_mov between MFMAs.
            const int n_kk = (kk_end - kk_begin + 1) >> 1;
            const float* qa = pa + kk_begin * CH_LDM;
            const float* qb = pb + kk_begin * CH_MAXW;
#define CH_LD(S, J)                                                                                              \
    S##a00 = qa[(J) * 2 * CH_LDM]; S##a10 = qa[(J) * 2 * CH_LDM + 32];                                           \
    S##a01 = qa[(J) * 2 * CH_LDM + a_off1]; S##a11 = qa[(J) * 2 * CH_LDM + a_off1 + 32];                         \
    S##b0 = qb[(J) * 2 * CH_MAXW]; S##b1 = qb[(J) * 2 * CH_MAXW + b_off1];
#define CH_MM(S)                                   \
    acc[0][0] = mfma32(S##a00, S##b0, acc[0][0]);  \
    acc[0][1] = mfma32(S##a01, S##b1, acc[0][1]);  \
    acc[1][0] = mfma32(S##a10, S##b0, acc[1][0]);  \
    acc[1][1] = mfma32(S##a11, S##b1, acc[1][1]);
            if (n_kk > 0) {
                float xa00, xa10, xa01, xa11, xb0, xb1, ya00, ya10, ya01, ya11, yb0, yb1;
                CH_LD(x, 0)
                int j = 0;
                while (j + 1 < n_kk) {
                    CH_LD(y, j + 1)
                    CH_MM(x)
                    if (j + 2 < n_kk) { CH_LD(x, j + 2) }
                    CH_MM(y)
                    j += 2;
                }
                if (j < n_kk) { CH_MM(x) }
            }
#undef CH_LD
#undef CH_MM
            if (PROF) { const long long t = clock64(); t_mfma += t - t0; t0 = t; }
            if (more && !dma) chain_store(stage, sB[buf ^ 1]);
            __syncthreads();   // (hipcc drains vmcnt(0) here while an LDS-DMA is in flight)
            buf ^= 1;
            if (PROF) { const long long t = clock64(); t_stage += t - t0; t0 = t; }
        }
        // every wave is past its last read of sAct and of the consumed chunk buffer sB[buf ^ 1]
        int n_epi = 2;                                            // 32-column tiles this wave finishes
        if (narrow) {
            // 8 partial 64x32 tiles -> (slice pairs summed in registers) -> LDS -> wave 0 sums in wave order
            float* scr = sB[buf ^ 1];   // the buffer just consumed; sB[buf] may already hold the next step's chunk
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((wave * 2 + tm) * 16 + r) * 64 + lane] = acc[tm][0][r] + acc[tm][1][r];
            __syncthreads();
            n_epi = (wave == 0) ? 1 : 0;
            if (wave == 0) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = scr[((0 * 2 + tm) * 16 + r) * 64 + lane];
                        v += scr[((1 * 2 + tm) * 16 + r) * 64 + lane];
                        v += scr[((2 * 2 + tm) * 16 + r) * 64 + lane];
                        v += scr[((3 * 2 + tm) * 16 + r) * 64 + lane];
                        acc[tm][0][r] = v;
                    }
            }
        }
        // ---- epilogue: passes with the wave-uniform conditions hoisted out of the element loops -----------
        const bool feed_next = (s + 1 < p.n_steps);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            if (tn >= n_epi) continue;
            const int col = nbase + tn * 32 + i;
            const bool col_ok = col < N;
            const float bias = (st.bias != nullptr && col_ok) ? st.bias[col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[tm][tn][r] + bias;
                    if (st.relu) v = fmaxf(v, 0.f);
                    acc[tm][tn][r] = col_ok ? v : 0.f;
                }
                if (st.mask != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const bool ok = col_ok && row < p.rows;
                        const float mk = ok ? st.mask[(size_t)row * st.ldmask + col] : 0.f;
                        acc[tm][tn][r] = (mk > 0.f) ? acc[tm][tn][r] : 0.f;
                    }
                }
                if (feed_next) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sAct[col * CH_LDM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[tm][tn][r];
                }
                if (st.out != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (col_ok && row < p.rows) st.out[(size_t)row * st.ldout + col] = acc[tm][tn][r];
                    }
                }
            }
        }
        if (feed_next) __syncthreads();    // sAct of the next step complete before anyone multiplies it
        if (PROF) { const long long t = clock64(); t_epi += t - t0; t0 = t; }
    }
    if (PROF && p.prof != nullptr && tid == 0) {
        long long* o = p.prof + (size_t)blockIdx.x * 8;
        o[0] = t_in; o[1] = t_mfma; o[2] = t_stage; o[3] = t_epi;
    }
}

__global__ __launch_bounds__(CH_THREADS) void mlp_chain_kernel(ChainArgs p) { mlp_chain_body<false, false>(p); }
__global__ __launch_bounds__(CH_THREADS) void mlp_chain_dma_kernel(ChainArgs p) { mlp_chain_body<false, true>(p); }

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
