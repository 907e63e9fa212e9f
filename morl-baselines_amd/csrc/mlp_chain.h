// Layer-fused MLP engine: a workgroup carries a TM-row tile (TM = 64 or 32) through a whole chain of dense layers
// with the activations RESIDENT IN LDS -- they never round-trip through HBM between layers (gfx950, wave64).
//
//   step s:   out_s[TM][N_s] = epilogue_s( act[TM][K_s] @ Bmat_s[K_s][N_s] ),   act <- out_s
//
// Used three ways (same kernel, different step tables):
//   * no-grad forward      Q(s', w) slabs:  Bmat = W_l^T (pre-transposed copy), epilogue bias+ReLU, only Q leaves
//   * training forward     same, but every hidden activation is also saved to HBM for the backward pass
//   * backward (dX chain)  Bmat = W_l as stored ([out][in] is already K-major for g_l @ W_l), epilogue = ReLU mask
//                          from the saved activation; every g_l is written out for the weight-gradient GEMM
//
// Work split: 256 threads = 4 waves; the waves split the OUTPUT COLUMNS (wave w owns columns [64w, 64w+64)) and
// share the activation tile.  Consequences that shape the kernel:
//   * A operand (activations, shared by all waves): LDS, K-major sAct[k][m], row stride TM+1 -> the MFMA operand
//     read "lane (i, h) <- A[m0+i][k+h]" and the transposed epilogue store are both bank-conflict free.
//   * B operand (weights): every wave needs a DIFFERENT 64-column slice, nothing is shared inside the workgroup, so
//     weights go straight from L2 into registers -- no LDS staging, no per-chunk barrier.  The MFMA column slot
//     (tn, i) is mapped to physical column 64w + 2i + tn, so one 8-byte load per lane fetches both B operands of a
//     k-row and the epilogue writes 8-byte pairs (full 256-B segments per row).  One 32-deep chunk of B (16 float2
//     per lane) is consumed while the next is in flight (two named register sets, no rotation moves), including
//     across layer boundaries.
//   * barriers only at layer boundaries (2 per layer); 33.8 / 66.6 KB of LDS -> 2+ workgroups per CU overlap each
//     other's epilogues and barriers.
//   * v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain), 2 x (TM/32) tiles per wave.
//   * steps with N <= 32 (the Q head, N = A*R) would idle three waves: there the four waves split the contraction
//     instead (wave w: k in [64w, 64w+64), weights read N-major so that a lane's 8-byte loads run along k) and the
//     partial tiles are summed through LDS in wave order (deterministic).
// Widths up to 256; Bmat row strides must be multiples of 4 floats and Bmat must be zero in columns [N, ldb) (host
// guarantees both, else the per-layer GEMM path is used).
//
// Roofline: fp32 MFMA; per row sum_s 2*K_s*N_s flop; weights re-read from L2 (per-XCD resident); algorithmic HBM
// bytes = inputs + outputs only.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

constexpr int CH_MAXW = 256;        // widest layer
constexpr int CH_BK = 32;           // K chunk (one register set of B)
constexpr int CH_THREADS = 256;

struct ChainStep {
    const float* Bmat;   // [K][ldb] K-major operand (row k contiguous over n), zero in columns [N, ldb)
    const float* Bt;     // [N][ldbt] the same matrix N-major (row n contiguous over k); needed when N <= 32
    const float* bias;   // [N] or NULL
    const float* mask;   // [rows][ldmask] or NULL: result kept where mask > 0 (ReLU backward)
    const unsigned long long* bits_in;   // or NULL: the same mask as one 64-bit word per work-item (see bits_out)
    unsigned long long* bits_out;        // or NULL: (output > 0) of this wide step, packed per work-item in the
                                         // 64-row tiling: word[(row0/64)*256 + tid], bit (tm*2 + tn)*16 + r
    float* out;          // [rows][ldout] or NULL: global copy of this step's output
    int K, N, ldb, ldbt, ldmask, ldout;
    int relu;
    int kpad;            // mlp_chain2: rows of Bmat physically present (>= K, zero beyond K); 0 = exactly K
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
    float* x0_out;          // in_mode 0, or NULL: the assembled rows also go to HBM as [rows][ldx0] (zero padded) -- the
    int ldx0;               // weight-gradient GEMM of layer 0 reads them; saves the separate input-assembly launch
    long long* prof;        // development probe only (tools/probes): [blocks][8] phase cycle counters
    int fast;               // mlp_chain2: every wide step has ldb == 256 and kpad a multiple of 64 (constant-stride weight stream)
};

// One register set of B operands: 16 eight-byte loads per lane.
//   wide step  (N > 32): v[j] = Bmat[k0 + 2j + h][64w + 2i .. +1]           -> MFMA group j, columns (tn = 0, 1)
//   narrow step (N <= 32): v[j] = Bt[i][64w + 4j + 2h .. +1]                 -> MFMA groups 2j (.x) and 2j+1 (.y),
//                          i.e. wave w contracts k in [64w, 64w+64) for output column i (split-K over the waves)
struct ChainBSet {
    float2 v[16];
};

// Both shapes are one strided gather  v[j] = *(float2*)(base + lane_off + j*stride)  issued as 16 unconditional
// buffer_load_dwordx2 through a buffer resource that spans exactly the matrix: an element beyond the last row (K
// padding, dummy prefetches) is out of range and the hardware returns 0 for it -- no branch, no select on the loaded
// value (a select would be a *use* and would pull the s_waitcnt right behind the load, collapsing the one-chunk
// prefetch distance), and the number of loads in flight is static, so the s_waitcnt vmcnt counts are exact.
// Narrow steps need an even K (the pair .x/.y runs along k).
constexpr int CH_OOB = 0x40000000;   // byte offset beyond any matrix (forces the out-of-range zero)

struct ChainBDesc {
    __amdgpu_buffer_rsrc_t rsrc;   // wave-uniform: the whole matrix [rows][ld]
    int lane_off;                  // bytes; CH_OOB when this lane's column does not exist
    int stride;                    // bytes per j (wave-uniform)
    int kfirst, kstep, K;          // narrow only: contraction index of .x of v[j] is kfirst + j*kstep, valid while < K
    bool narrow;
};

__device__ __forceinline__ ChainBDesc chain_desc(const ChainStep& st, int k0, int wave, int i, int h) {
    ChainBDesc d;
    d.narrow = st.N <= 32;
    const int colw = wave * 64 + 2 * i;
    const float* base = d.narrow ? st.Bt : st.Bmat;
    const int bytes = (d.narrow ? st.N * st.ldbt : st.K * st.ldb) * 4;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    const bool lane_ok = d.narrow ? (i < st.N) : (colw < st.ldb);
    const int off = d.narrow ? (i * st.ldbt + wave * 64 + 2 * h) : ((k0 + h) * st.ldb + colw);
    d.lane_off = lane_ok ? off * 4 : CH_OOB;
    d.stride = (d.narrow ? 4 : 2 * st.ldb) * 4;
    d.kfirst = wave * 64 + 2 * h;
    d.kstep = 4;
    d.K = st.K;
    return d;
}

__device__ __forceinline__ void chain_load_b(ChainBSet& s, const ChainBDesc& d) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int off = d.lane_off + j * d.stride;
        // wide: rows >= K lie beyond the resource -> 0 by range check.  narrow: k >= K would run into the next row
        if (d.narrow && d.kfirst + j * d.kstep >= d.K) off = CH_OOB;
        s.v[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(d.rsrc, off, 0, 0));
    }
}

#define CH_TICK(slot)                                                   \
    if (PROF) { const long long t_ = clock64(); tacc[slot] += t_ - t0; t0 = t_; }

template <int TM, bool PROF = false>
__device__ __forceinline__ void mlp_chain_body(const ChainArgs& p, int tile) {
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    if (PROF) t0 = clock64();
    constexpr int LDM = TM + 1;
    constexpr int MT = TM / 32;                       // 32-row MFMA tiles per wave
    __shared__ float sAct[CH_MAXW * LDM];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int h = lane >> 5, i = lane & 31;
    const int row0 = tile * TM;
    const int colw = wave * 64 + 2 * i;               // first of this lane's two physical output columns (wide steps)

    ChainBSet bx, by;
    // the weight stream starts before the input tile is assembled
    chain_load_b(bx, chain_desc(p.step[0], 0, wave, i, h));

    // ---- input tile -> sAct[k][m], zero-padded to the rows the first step multiplies (a wide step: the next
    //      multiple of 64, and its epilogue then defines all 256 rows; a narrow first step reads the whole buffer), so
    //      that rows beyond a step's K -- met by zero weights -- are always finite --------------------------------
    {
        const int K0 = (p.in_mode == 0) ? (p.D + p.R) : p.K0;
        const int K0pad = (p.step[0].N > 32) ? min(CH_MAXW, (K0 + 63) & ~63) : CH_MAXW;
        const int m = tid % TM;
        const int row = row0 + m;
        int b = row, w = row;
        if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; w = row - b * p.W; }
            else if (p.row_order == 1) { w = row / p.B; b = row - w * p.B; }
        }
        const bool row_ok = row < p.rows;
        for (int k = tid / TM; k < K0pad; k += CH_THREADS / TM) {
            float v = 0.f;
            if (row_ok && k < K0) {
                if (p.in_mode == 0) v = (k < p.D) ? p.obs[(size_t)b * p.D + k] : p.weights[(size_t)w * p.R + (k - p.D)];
                else v = p.src[(size_t)row * p.ldsrc + k];
            }
            sAct[k * LDM + m] = v;
            if (p.x0_out != nullptr && row_ok && k < p.ldx0) p.x0_out[(size_t)row * p.ldx0 + k] = v;
        }
    }
    __syncthreads();
    CH_TICK(0)                                                    // input assembly + first weight set issued

    for (int s = 0; s < p.n_steps; ++s) {
        const ChainStep& st = p.step[s];
        const int K = st.K, N = st.N;
        const bool feed_next = (s + 1 < p.n_steps);
        const ChainStep& nxt = p.step[feed_next ? s + 1 : s];    // what the stream fetches after this step (or a dummy)

        if (N > 32) {
            // ======================= matrix-core path =======================================================
            f32x16 acc[MT][2];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

            // chunks are consumed in pairs (bx then by); K is treated as padded to a multiple of 64 with zero rows
            const int n_pairs = (K + 63) >> 6;
            chain_load_b(by, chain_desc(st, CH_BK, wave, i, h));
            for (int pr = 0; pr < n_pairs; ++pr) {
                const int k0 = pr * 64;
                const bool more = pr + 1 < n_pairs;
                const float* pa = sAct + (k0 + h) * LDM + i;
// the activation operands of k-pair j+1 are read before the MFMAs of k-pair j are issued (one wave per SIMD may
// be all there is: nothing else hides the ds_read latency)
#define CH_COMPUTE(SET, KOFF)                                                              \
    {                                                                                      \
        float a0 = pa[(KOFF) * LDM], a1 = (MT == 2) ? pa[(KOFF) * LDM + 32] : 0.f;         \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                   \
            const int jn = (j < 15) ? j + 1 : j;                                           \
            const float n0 = pa[((KOFF) + 2 * jn) * LDM];                                  \
            const float n1 = (MT == 2) ? pa[((KOFF) + 2 * jn) * LDM + 32] : 0.f;           \
            acc[0][0] = mfma32(a0, SET.v[j].x, acc[0][0]);                                 \
            acc[0][1] = mfma32(a0, SET.v[j].y, acc[0][1]);                                 \
            if (MT == 2) {                                                                 \
                acc[MT - 1][0] = mfma32(a1, SET.v[j].x, acc[MT - 1][0]);                   \
                acc[MT - 1][1] = mfma32(a1, SET.v[j].y, acc[MT - 1][1]);                   \
            }                                                                              \
            a0 = n0; a1 = n1;                                                              \
        }                                                                                  \
    }
                CH_COMPUTE(bx, 0)
                // bx is free again: fetch the chunk two ahead -- this step's, else the next step's first set (a wide
                // chunk 0 or the narrow head's K-slice); never skipped, so the number of loads in flight is static
                chain_load_b(bx, chain_desc(more ? st : nxt, more ? k0 + 64 : 0, wave, i, h));
                CH_COMPUTE(by, CH_BK)
                chain_load_b(by, chain_desc(st, more ? k0 + 96 : CH_BK, wave, i, h));   // (dummy re-read on the last pair)
#undef CH_COMPUTE
            }
            CH_TICK(1)                                           // MFMA pair loop (incl. operand prefetch issue)
            __syncthreads();     // every wave is past its last read of sAct
            CH_TICK(2)                                           // barrier wait after the loop

            // ---- epilogue (wave-uniform conditions hoisted out of the element loops) ------------------------
            const bool col_ok = colw < N;
            const bool col1_ok = colw + 1 < N;
            float bias0 = 0.f, bias1 = 0.f;
            if (st.bias != nullptr) {
                if (col_ok) bias0 = st.bias[colw];
                if (col1_ok) bias1 = st.bias[colw + 1];
            }
            // ReLU masks travel between the training forward and the backward chain as bits: 8 bytes per work-item
            // and layer instead of 64 floats (both chains tile [rows][N] identically; a 32-row tile uses its half)
            unsigned long long bits_w = 0ull, bits_r = 0ull;
            const size_t bits_idx = (size_t)(row0 >> 6) * CH_THREADS + tid;
            const int bits_shift = (TM == 32) ? ((row0 >> 5) & 1) * 32 : 0;
            if (st.bits_in != nullptr) bits_r = st.bits_in[bits_idx] >> bits_shift;
#pragma unroll
            for (int tm = 0; tm < MT; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v0 = acc[tm][0][r] + bias0, v1 = acc[tm][1][r] + bias1;
                    if (st.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    acc[tm][0][r] = col_ok ? v0 : 0.f;
                    acc[tm][1][r] = col1_ok ? v1 : 0.f;
                }
                if (st.bits_out != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (acc[tm][0][r] > 0.f) bits_w |= 1ull << ((tm * 2 + 0) * 16 + r);
                        if (acc[tm][1][r] > 0.f) bits_w |= 1ull << ((tm * 2 + 1) * 16 + r);
                    }
                }
                if (st.bits_in != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc[tm][0][r] = ((bits_r >> ((tm * 2 + 0) * 16 + r)) & 1ull) ? acc[tm][0][r] : 0.f;
                        acc[tm][1][r] = ((bits_r >> ((tm * 2 + 1) * 16 + r)) & 1ull) ? acc[tm][1][r] : 0.f;
                    }
                } else if (st.mask != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float2 mk = make_float2(0.f, 0.f);
                        if (col_ok && row < p.rows) {
                            if (col1_ok) mk = *reinterpret_cast<const float2*>(st.mask + (size_t)row * st.ldmask + colw);
                            else mk.x = st.mask[(size_t)row * st.ldmask + colw];
                        }
                        acc[tm][0][r] = (mk.x > 0.f) ? acc[tm][0][r] : 0.f;
                        acc[tm][1][r] = (mk.y > 0.f) ? acc[tm][1][r] : 0.f;
                    }
                }
                if (feed_next) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        sAct[colw * LDM + m] = acc[tm][0][r];
                        sAct[(colw + 1) * LDM + m] = acc[tm][1][r];
                    }
                }
                if (st.out != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (col_ok && row < p.rows) {
                            float* o = st.out + (size_t)row * st.ldout + colw;
                            if (col1_ok) *reinterpret_cast<float2*>(o) = make_float2(acc[tm][0][r], acc[tm][1][r]);
                            else o[0] = acc[tm][0][r];
                        }
                    }
                }
            }
            if (st.bits_out != nullptr) {
                if (TM == 64) st.bits_out[bits_idx] = bits_w;
                else   // a 32-row tile owns one 32-bit half of the word (rows 0-31 / 32-63 of the 64-row band)
                    reinterpret_cast<unsigned int*>(st.bits_out)[2 * bits_idx + ((row0 >> 5) & 1)] = (unsigned int)bits_w;
            }
            CH_TICK(3)                                           // wide epilogue
        } else {
            // ======================= narrow step (Q head): split-K over the four waves ========================
            // bx holds Bt[i][64w + 4j + 2h + {0,1}]: wave w contracts k in [64w, 64w+64) for output column i.
            f32x16 hacc[MT];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[a][r] = 0.f;
            const int ks = wave * 64 + 2 * h;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int k = ks + 4 * j + c;
                    const float bv = c ? bx.v[j].y : bx.v[j].x;
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm)   // rows >= K: finite stale values times zero weights
                        hacc[tm] = mfma32(sAct[k * LDM + tm * 32 + i], bv, hacc[tm]);
                }
            }
            __syncthreads();     // every wave is past its last read of sAct -> reuse it as the reduction scratch
            float* scr = sAct;
#pragma unroll
            for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((wave * MT + tm) * 16 + r) * 64 + lane] = hacc[tm][r];
            // the stream moves on while the partial tiles are reduced
            chain_load_b(bx, chain_desc(nxt, 0, wave, i, h));
            __syncthreads();
            // thread (rg = wave, lane) sums the four partials of registers 4rg..4rg+3 of every row tile, wave order
            float red[MT][4];
#pragma unroll
            for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = wave * 4 + q;
                    float v = scr[((0 * MT + tm) * 16 + r) * 64 + lane];
                    v += scr[((1 * MT + tm) * 16 + r) * 64 + lane];
                    v += scr[((2 * MT + tm) * 16 + r) * 64 + lane];
                    v += scr[((3 * MT + tm) * 16 + r) * 64 + lane];
                    red[tm][q] = v;
                }
            if (feed_next) __syncthreads();   // scratch fully consumed before sAct is rewritten
            const int n = i;
            const float bias = (st.bias != nullptr && n < N) ? st.bias[n] : 0.f;
#pragma unroll
            for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = wave * 4 + q;
                    const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int row = row0 + m;
                    const bool ok = n < N && row < p.rows;
                    float v = red[tm][q] + bias;
                    if (st.relu) v = fmaxf(v, 0.f);
                    if (st.mask != nullptr) v = (ok && st.mask[(size_t)row * st.ldmask + n] > 0.f) ? v : 0.f;
                    if (!ok) v = 0.f;
                    if (feed_next) sAct[n * LDM + m] = v;
                    if (st.out != nullptr && ok) st.out[(size_t)row * st.ldout + n] = v;
                }
            if (feed_next) {
                // the next step reads K' = N <= 32 padded to 64 rows: rows [32, 64) must be zero too
                for (int e = tid; e < 32 * TM; e += CH_THREADS) sAct[(32 + e / TM) * LDM + (e % TM)] = 0.f;
            }
            CH_TICK(4)                                           // narrow head (MFMAs, LDS reduction, epilogue)
        }
        if (feed_next) __syncthreads();    // sAct of the next step complete before anyone multiplies it
        CH_TICK(5)                                               // barrier before the next step
    }
    if (PROF && p.prof != nullptr && tid == 0)
        for (int q = 0; q < 8; ++q) p.prof[(size_t)blockIdx.x * 8 + q] = tacc[q];
}

// second launch-bound argument = waves per SIMD: two workgroups per CU must fit (<= 256 VGPR+AGPR per lane)
__global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain64_kernel(ChainArgs p) { mlp_chain_body<64>(p, (int)blockIdx.x); }
__global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain32_kernel(ChainArgs p) { mlp_chain_body<32>(p, (int)blockIdx.x); }

// Several independent chains in ONE launch (the three forward passes of an Envelope step: online / target network on
// the next-state rows, online network on the TD rows).  With 64-row tiles a single pass gives only one workgroup per
// CU; launched together the passes put 2 workgroups on every CU, whose barriers / epilogues / prologues overlap, while
// the weight stream per pass stays what one 64-row tiling costs.
constexpr int CH_MAX_MULTI = 3;
struct ChainMulti {
    ChainArgs p[CH_MAX_MULTI];
    int tile_start[CH_MAX_MULTI + 1];   // first block of each chain
    int n;
};
__global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain64_multi_kernel(ChainMulti m) {
    const int bid = (int)blockIdx.x;
    int q = 0;
    while (q + 1 < m.n && bid >= m.tile_start[q + 1]) ++q;
    mlp_chain_body<64>(m.p[q], bid - m.tile_start[q]);
}
// 32-row tiles, three workgroups per CU (<= 170 registers): 3 x rows/32 workgroups = two full rounds of the chip
__global__ __launch_bounds__(CH_THREADS, 3) void mlp_chain32_multi_kernel(ChainMulti m) {
    const int bid = (int)blockIdx.x;
    int q = 0;
    while (q + 1 < m.n && bid >= m.tile_start[q + 1]) ++q;
    mlp_chain_body<32>(m.p[q], bid - m.tile_start[q]);
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

// blockIdx.y = 0: params -> wt; 1: params2 -> wt2 (online and target net of one update step in a single launch)
__global__ __launch_bounds__(256) void transpose_params_kernel(const float* __restrict__ params,
                                                               float* __restrict__ wt, const float* __restrict__ params2,
                                                               float* __restrict__ wt2, TransposeArgs t) {
    if (blockIdx.y == 1) { params = params2; wt = wt2; }
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
