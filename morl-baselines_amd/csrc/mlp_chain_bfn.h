// Few-row form of the split-bf16 chain (round 6): a LATENCY-optimised layer-fused MLP chain for launches of at most a few thousand
// rows -- the lazily evaluated target rows of an Envelope step (1 200 - 3 200 of 16 384), acting / evaluation batches, the 2 048-row
// share of a rank of an 8-rank job -- where mlp_chain_bf.h's 64-row tiles leave most of the chip idle and every tile is one wave's
// serial chain per SIMD.
//
//   * tile = 16 or 32 rows, workgroup = 4 waves: the waves split the OUTPUT FEATURES of every wide (256-column) step, 64 each (feature
//     tiles 4 w .. 4 w + 3) -- a 256 x 256 step is 192 MFMAs per wave and 16 rows instead of 768 on one wave (mlp_chain_bf.h with a 16-row
//     tile).  What sets the tile size is bytes, not the matrix pipe: a tile streams the whole network (1.27 MB of split weights) through
//     one CU -- ~115 GB/s with every CU pulling, 11 us -- so a launch takes 16-row tiles while that is at most a tile per CU (more CUs
//     pull) and 32-row tiles beyond (two sub-tiles per wave: every fetched fragment feeds two MFMAs, half the tiles, half the L2 bytes);
//   * weights straight from L2 into REGISTERS: a wave reads only its own 64 features' fragments (a quarter of the stream, no sharing
//     between the waves, so nothing to stage in LDS, no barrier in the stream): the 12 fragment blocks of a k-step (4 tiles x 3 split
//     parts, 12 KB contiguous in bf_split_kernel's order) as 12 coalesced 16-byte buffer loads, two k-steps resident (the one
//     being multiplied and one in flight), the stream running ahead ACROSS layer boundaries
//     (addresses do not depend on data; beyond the last block the descriptor's range check returns zeros: no branch in the stream);
//   * activations exchanged through LDS once per layer: a wave's outputs ARE k-steps 2 w, 2 w + 1 of the next step's B operand in
//     mlp_chain_bf.h's slot order (slot (q, e) <-> feature 32 s + 16 (e >> 2) + 4 q + (e & 3)): epilogue (bias as the accumulator's
//     initial value, ReLU / mask, three-way split) lane-local as there, six 1 KB blocks written per wave, ONE barrier per layer
//     (the buffer alternates between layers), every wave reads all 24 blocks back as its B operands, a k-step ahead of their use.
// Same arithmetic as mlp_chain_bf.h (six products of three-way bf16 splits, fp32 accumulate: fp32-class), same weight stream, same
// sign-bit words (a wave owns bits [16 w, 16 w + 16) of a lane's word), same saved fp32 activations / gradients: the training
// forward (MODE 1) and the dX backward (MODE 2) of a small step interoperate with dw_bf.h / dw_tiles.h unchanged.
// Replaces: QNet.forward on few rows (envelope.py:300, :420, :429-439; common/networks.py:10-48) and the dX half of loss.backward()
// (envelope.py:323) of small steps.
#pragma once
#include "mlp_chain_bf.h"

namespace morl {

#ifndef BFN_PIN_
#define BFN_PIN_ 1
#endif
constexpr int BFN_RT = 2;                                    // 16-row sub-tiles a wave carries: 1 or 2 (mlp_chain_bfn16_kernel / mlp_chain_bfn_kernel)
constexpr int BFN_RT_MAX = 4;                                // ... and in the 64-row form (measured and dropped: see the end of this file)
#ifndef BFN_DEPTH_
#define BFN_DEPTH_ 2
#endif
// weight groups (k-steps) resident per wave: the one being multiplied + DEPTH - 1 in flight.  TWO: a CU's vector-memory path
// delivers LESS with more loads queued -- 145 GB/s at 8 in flight per lane, 134 at 24, 88 at 48 (tools/probes/stream_probe.hip,
// profiles/r05_stream_probe.txt) -- and a group is 12 loads per lane; a k-step of both sub-tiles is 768 matrix-pipe cycles of cover
constexpr int BFN_DEPTH = BFN_DEPTH_;
constexpr int BFN_TM = 16 * BFN_RT;                          // rows per workgroup
constexpr int BFN_THREADS = 256;                             // 4 waves
constexpr int BFN_XBLOCKS = 24;                              // (k-step, part) blocks of one layer's activations
constexpr int BFN_XBUF_BYTES = BFN_XBLOCKS * BF_BLOCK;
// activations in LDS: two layers' worth of both sub-tiles, alternating (32-row tiles: one barrier per layer), or ONE layer's worth of
// four sub-tiles (64-row tiles: a barrier in front of the epilogue's writes and one behind them) -- 96 KB either way
constexpr int BFN_LDS_BYTES = 2 * BFN_RT * BFN_XBUF_BYTES;
constexpr int BFN_BIAS_BYTES = BF_MAX_STEPS * BF_WIDE * 4;   // every step's bias, [step][256] floats (zeros beyond a step's columns)
constexpr int BFN_MAX_MULTI = 3;                             // (the three forward passes of an eagerly evaluated few-row step)

struct BfnMulti {
    BfChain c[BFN_MAX_MULTI];
    int tile_start[BFN_MAX_MULTI + 1];     // 16-row tiles of chain q: [tile_start[q], tile_start[q + 1])
    int n;
    int n_blocks[BFN_MAX_MULTI];           // 1 KB blocks of chain q's stream (range of the stream's descriptor)
};

// ---- the weight stream of ONE wave ------------------------------------------------------------------------------------------------
// group = the fragment blocks a wave multiplies in one k-step of one step: byte offset base[l] + s * stride[l]; `cl`, `cs`: the
// group the next issue fetches (two ahead of the one being multiplied).  Scalar state only.
struct BfnCursor {
    // every wide step's blocks are laid out alike -- (k-step s, tile t, part p) at ((16 s + t) * 3 + p) KB behind the step before --, so
    // this wave's wide groups are ONE arithmetic progression over the chain: group g at (12 wave + 48 g) KB, g < g_wide; then the
    // head's eight groups (its tile `wave`, if it has one).  No tables: nothing here is indexed at run time.
    int g;                       // the group the next issue fetches
    int g_wide, g_all;           // wide groups / all groups of this wave
    int wide0, head0, head_stride;
    int beyond;                  // a byte offset past the stream: groups after the last read as zeros
};
__device__ __forceinline__ int bfn_next_offset(BfnCursor& c) {
    const int g = c.g++;
    if (g < c.g_wide) return c.wide0 + g * (48 * BF_BLOCK);
    if (g < c.g_all) return c.head0 + (g - c.g_wide) * c.head_stride;
    return c.beyond;
}
// the 12 blocks of a group into ring slot SLOT (unconditional: a group beyond the stream reads zeros)
template <int SLOT>
__device__ __forceinline__ void bfn_issue(bf_u32x4 (&wr)[BFN_DEPTH][4][3], const __amdgpu_buffer_rsrc_t& rsrc, int voff, int off) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int b = 3 * j + p;
            wr[SLOT][j][p] = __builtin_bit_cast(bf_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (b & 3) * BF_BLOCK, off + (b >> 2) * 4 * BF_BLOCK, 0));
        }
}

// LDS barrier that leaves the weight loads in flight: this wave's LDS writes have landed, then the workgroup barrier
#define BFN_BARRIER() do { BF_PIN(); __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8)); __builtin_amdgcn_s_barrier(); BF_PIN(); } while (0)

// One step's products for this wave's NT output tiles over KSTEPS k-steps.  FIRST: the B operand is the assembled input (registers),
// else the previous step's activations in LDS (`xin`: 24 blocks, (k-step s, part p) at (3 s + p) KB), read a k-step ahead.
// Ring slot of k-step s: (PHASE0 + s) % 3; after its products the slot takes the group three ahead.
template <int RT, int PHASE0, int KSTEPS, bool FIRST, int NT>
__device__ __forceinline__ void bfn_products(f32x4 (&acc)[RT][4], bf_u32x4 (&wr)[BFN_DEPTH][4][3], const bf_u32x4 (&x0)[RT][2][3],
                                             const unsigned char* xin, const __amdgpu_buffer_rsrc_t& rsrc, int voff, BfnCursor& cur, int lane) {
    constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};
    bf_u32x4 xa[RT][2][3];
    if (!FIRST) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) xa[t][0][p] = *reinterpret_cast<const bf_u32x4*>(xin + t * BFN_XBUF_BYTES + p * BF_BLOCK + lane * 16);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int slot = (PHASE0 + s) % BFN_DEPTH;
        if (!FIRST && s + 1 < KSTEPS) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    xa[t][(s + 1) & 1][p] = *reinterpret_cast<const bf_u32x4*>(xin + t * BFN_XBUF_BYTES + (3 * (s + 1) + p) * BF_BLOCK + lane * 16);
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const bf_u32x4& xv = FIRST ? x0[t][s < 2 ? s : 1][px[p]] : xa[t][s & 1][px[p]];
                    // (one tile -- the head --: the k-steps alternate between two accumulators, summed by the caller: no chain of 48
                    // dependent MFMAs)
                    const int ai = (NT == 1) ? (s & 1) : j;
                    acc[t][ai] = bf_mfma(wr[slot][j][pw[p]], xv, acc[t][ai]);
                }
            }
        }
        // the slot's refill goes out HERE, behind its k-step's last product and in front of the next k-step's first: left to the
        // scheduler it drifted a k-step down the unrolled loop (one group of look-ahead instead of two)
        if (BFN_PIN_) BF_PIN();
        const int off = bfn_next_offset(cur);
        if (slot == 0) bfn_issue<0>(wr, rsrc, voff, off);
        else if (slot == 1) bfn_issue<1>(wr, rsrc, voff, off);
        else bfn_issue<(BFN_DEPTH > 2 ? 2 : 0)>(wr, rsrc, voff, off);
        if (BFN_PIN_) BF_PIN();
    }
}

// Epilogue of a wide step for this wave's four tiles (16 T + 4 q + r, T = 4 wave + j): ReLU / mask, fp32 copy + sign bits (MODE 1) or
// mask in + fp32 copy (MODE 2), split, and the six blocks of k-steps 2 wave, 2 wave + 1 of the next step's operand into `xout`.
template <int MODE>
__device__ __forceinline__ void bfn_epilogue(const f32x4 (&acc)[4], const BfStep& st, unsigned char* xout, int rows, int row, bool row_ok,
                                             size_t bits_idx, int wave, int lane, int q) {
    unsigned keep = 0xffffu, pos = 0u;
    if (MODE == 2 && st.bits_in != nullptr) keep = reinterpret_cast<const unsigned short*>(st.bits_in)[bits_idx * 4 + wave];
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)st.out, 0, (MODE != 0 && st.out != nullptr) ? rows * st.ldout * 4 : 0, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 2 * jj + (e >> 2), r = e & 3, k = 4 * j + r;          // bit k of this wave's 16-bit piece
            float a = acc[j][r];
            if (MODE != 2 && st.relu) a = bf_relu(a);
            if (MODE == 2) a = ((keep >> k) & 1u) ? a : 0.f;
            if (MODE == 1) pos |= (__builtin_bit_cast(unsigned, a) != 0u ? 1u : 0u) << k;
            v[e] = a;
        }
        if (MODE != 0) {
            const int voff = row_ok ? (row * st.ldout + 16 * (4 * wave + 2 * jj) + 4 * q) * 4 : CH_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(bf_u32x4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                                            __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])}, orsrc, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(bf_u32x4{__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]),
                                                            __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])}, orsrc, voff + 64, 0, 0);
        }
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bf_split2(v[2 * u], v[2 * u + 1], hi[u], mid[u], lo[u]);
        unsigned char* dst = xout + (3 * (2 * wave + jj)) * BF_BLOCK + lane * 16;
        *reinterpret_cast<bf_u32x4*>(dst) = bf_u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<bf_u32x4*>(dst + BF_BLOCK) = bf_u32x4{mid[0], mid[1], mid[2], mid[3]};
        *reinterpret_cast<bf_u32x4*>(dst + 2 * BF_BLOCK) = bf_u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
    // (after the ReLU a value is 0 or positive: "!= 0" is the sign bit mlp_chain_bf.h stores; -0 cannot occur: bf_relu maps it to +0)
    if (MODE == 1 && st.bits_out != nullptr && row_ok) reinterpret_cast<unsigned short*>(st.bits_out)[bits_idx * 4 + wave] = (unsigned short)pos;
}

// The biases of every step go to LDS ONCE, at the top of the kernel (one 4-byte buffer load per work-item and step, range-checked:
// zeros beyond a step's columns / without a bias), in front of the weight stream -- the oldest loads the wave ever issues.  A bias
// loaded at the top of its own step is waited for behind every weight group in flight (the counter retires in issue order: the ring
// drained once per layer), and one loaded a step ahead is more than 63 loads old when it is used, which the compiler's wait insertion
// answers with vmcnt(0) just the same; an accumulator's initial value then is an LDS read.
__device__ __forceinline__ f32x4 bfn_bias_at(const float* bias_lds, int step, int tile, int q) {
    return *reinterpret_cast<const f32x4*>(bias_lds + step * BF_WIDE + 16 * tile + 4 * q);
}

// ---- one 32-row tile (two sub-tiles) through a whole chain ------------------------------------------------------------------------
template <int RT, int K0S, int MODE>
__device__ __forceinline__ void bfn_chain_body(const BfChain& p, int n_blocks, int row0, int n_rows, unsigned char* lds) {
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int n_wide = p.n_steps - (p.head ? 1 : 0);
    const int head_tiles = p.head ? (p.step[p.n_steps - 1].N + 15) / 16 : 0;

    // ---- this wave's share of the stream --------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.stream, 0, n_blocks * BF_BLOCK, 0x00020000);
    const int voff = lane * 16;
    BfnCursor cur;
    cur.g = 0;
    cur.g_wide = K0S + 8 * (n_wide - 1);
    cur.g_all = cur.g_wide + ((p.head && wave < head_tiles) ? 8 : 0);
    cur.wide0 = 12 * wave * BF_BLOCK;
    cur.head0 = (cur.g_wide * 48 + 3 * wave) * BF_BLOCK;
    cur.head_stride = head_tiles * 3 * BF_BLOCK;
    cur.beyond = n_blocks * BF_BLOCK;
    float* bias_lds = reinterpret_cast<float*>(lds + 2 * RT * BFN_XBUF_BYTES);
    float bv[BF_MAX_STEPS];
#pragma unroll
    for (int st = 0; st < BF_MAX_STEPS; ++st) {
        const float* bp = p.step[st].bias;
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, (st < p.n_steps && bp != nullptr) ? p.step[st].N * 4 : 0, 0x00020000);
        bv[st] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, tid * 4, 0, 0));
    }
    bf_u32x4 wr[BFN_DEPTH][4][3];
    bfn_issue<0>(wr, rsrc, voff, bfn_next_offset(cur));
    bfn_issue<1>(wr, rsrc, voff, bfn_next_offset(cur));
    if (BFN_DEPTH > 2) bfn_issue<(BFN_DEPTH > 2 ? 2 : 0)>(wr, rsrc, voff, bfn_next_offset(cur));

    // ---- input rows: natural contraction order, slot (q, e) of k-step s <-> column 32 s + 8 q + e (every wave assembles them) -------
    bf_u32x4 x0[RT][2][3];
    int rowt[RT];
    bool okt[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = row0 + 16 * t + m;
        const bool row_ok = row < n_rows;
        rowt[t] = row; okt[t] = row_ok;
        int b = row, w = row;
        if (p.in_mode == 3) {
            const int n_pairs = p.B * p.W;
            int flat = p.pairs[row_ok ? row : (n_rows > 0 ? n_rows - 1 : 0)];
            flat = min(max(flat, 0), n_pairs - 1);
            b = flat / p.W; w = flat - b * p.W;
        } else if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; w = row - b * p.W; }
            else if (p.row_order == 1) { w = row / p.B; b = row - w * p.B; }
        }
        const bool cat = p.in_mode == 0 || p.in_mode == 3;
        const int K0 = cat ? p.D + p.R : p.K0;
        const float* src_a = cat ? p.obs + (size_t)b * p.D : p.src + (size_t)row * p.ldsrc;
        const float* src_w = cat ? p.weights + (size_t)w * p.R : nullptr;
#pragma unroll
        for (int s = 0; s < K0S; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * s + 8 * q + e;
                float a = 0.f;
                if (row_ok && k < K0) a = cat ? ((k < p.D) ? src_a[k] : src_w[k - p.D]) : src_a[k];
                v[e] = a;
            }
            if (p.x0_out != nullptr && row_ok && wave == 0) {
                const int k = 32 * s + 8 * q;
                float* o = p.x0_out + (size_t)row * p.ldx0 + k;
                if (k < p.ldx0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);            // (ldx0 is a multiple of 4)
                if (k + 4 < p.ldx0) *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bf_split2(v[2 * u], v[2 * u + 1], hi[u], mid[u], lo[u]);
            x0[t][s][0] = bf_u32x4{hi[0], hi[1], hi[2], hi[3]};
            x0[t][s][1] = bf_u32x4{mid[0], mid[1], mid[2], mid[3]};
            x0[t][s][2] = bf_u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
        if (K0S == 1) { x0[t][1][0] = x0[t][0][0]; x0[t][1][1] = x0[t][0][1]; x0[t][1][2] = x0[t][0][2]; }
    }

    // the bias table (behind the input rows' loads, in front of the first product: the weight groups stay in flight across the barrier)
#pragma unroll
    for (int st = 0; st < BF_MAX_STEPS; ++st) bias_lds[st * BF_WIDE + tid] = bv[st];
    BFN_BARRIER();
    f32x4 acc[RT][4];
    // ---- first step ---------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = bfn_bias_at(bias_lds, 0, 4 * wave + j, q);
    bfn_products<RT, 0, K0S, true, 4>(acc, wr, x0, nullptr, rsrc, voff, cur, lane);
#pragma unroll
    for (int t = 0; t < RT; ++t)
        bfn_epilogue<MODE>(acc[t], p.step[0], lds + t * BFN_XBUF_BYTES, n_rows, rowt[t], okt[t], (size_t)((row0 >> 4) + t) * 64 + lane, wave, lane, q);
    BFN_BARRIER();
    int phase = K0S % BFN_DEPTH;
    // ---- the 256 x 256 steps --------------------------------------------------------------------------------------------------------
    constexpr bool DB = RT <= 2;       // two alternating activation buffers, or one with a barrier on either side of the writes
    for (int s = 1; s < n_wide; ++s) {
        const unsigned char* xin = lds + (DB ? ((s - 1) & 1) * RT * BFN_XBUF_BYTES : 0);
        unsigned char* xout = lds + (DB ? (s & 1) * RT * BFN_XBUF_BYTES : 0);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = bfn_bias_at(bias_lds, s, 4 * wave + j, q);
        if (phase == 0) bfn_products<RT, 0, 8, false, 4>(acc, wr, x0, xin, rsrc, voff, cur, lane);
        else if (phase == 1) bfn_products<RT, 1, 8, false, 4>(acc, wr, x0, xin, rsrc, voff, cur, lane);
        else bfn_products<RT, (BFN_DEPTH > 2 ? 2 : 0), 8, false, 4>(acc, wr, x0, xin, rsrc, voff, cur, lane);
        phase = (phase + 8) % BFN_DEPTH;
        if (!DB) BFN_BARRIER();             // (every wave has read the step's input: its buffer takes the output)
#pragma unroll
        for (int t = 0; t < RT; ++t)
            bfn_epilogue<MODE>(acc[t], p.step[s], xout + t * BFN_XBUF_BYTES, n_rows, rowt[t], okt[t], (size_t)((row0 >> 4) + t) * 64 + lane, wave, lane, q);
        BFN_BARRIER();
    }
    // ---- the head: output tile `wave` (N <= 32: one or two tiles) ---------------------------------------------------------------------
    if (p.head && wave < head_tiles) {
        const BfStep& st = p.step[p.n_steps - 1];
        const unsigned char* xin = lds + (DB ? ((n_wide - 1) & 1) * RT * BFN_XBUF_BYTES : 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) { acc[t][0] = bfn_bias_at(bias_lds, p.n_steps - 1, wave, q); acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (phase == 0) bfn_products<RT, 0, 8, false, 1>(acc, wr, x0, xin, rsrc, voff, cur, lane);
        else if (phase == 1) bfn_products<RT, 1, 8, false, 1>(acc, wr, x0, xin, rsrc, voff, cur, lane);
        else bfn_products<RT, (BFN_DEPTH > 2 ? 2 : 0), 8, false, 1>(acc, wr, x0, xin, rsrc, voff, cur, lane);
#pragma unroll
        for (int t = 0; t < RT; ++t)
            if (st.out != nullptr && okt[t]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * wave + 4 * q + r;
                    if (n < st.N) st.out[(size_t)rowt[t] * st.ldout + n] = acc[t][0][r] + acc[t][1][r];
                    else if (n < st.ldout) st.out[(size_t)rowt[t] * st.ldout + n] = 0.f;          // (pad columns of a padded Q buffer)
                }
            }
    }
    if (p.head && head_tiles == 1 && wave == 1) {
        // (a head of <= 16 columns in a buffer padded beyond them: the pad columns of tile 1, which no wave computes)
        const BfStep& st = p.step[p.n_steps - 1];
#pragma unroll
        for (int t = 0; t < RT; ++t)
            if (st.out != nullptr && okt[t])
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 + 4 * q + r;
                    if (n < st.ldout) st.out[(size_t)rowt[t] * st.ldout + n] = 0.f;
                }
    }
}

template <int RT, int K0S>
__device__ __forceinline__ void bfn_dispatch(const BfChain& p, int n_blocks, int row0, int n_rows, unsigned char* lds) {
    const int mode = p.step[0].bits_in != nullptr ? 2 : (p.step[0].out != nullptr || p.step[0].bits_out != nullptr) ? 1 : 0;
    if (mode == 2) bfn_chain_body<RT, K0S, 2>(p, n_blocks, row0, n_rows, lds);
    else if (mode == 1) bfn_chain_body<RT, K0S, 1>(p, n_blocks, row0, n_rows, lds);
    else bfn_chain_body<RT, K0S, 0>(p, n_blocks, row0, n_rows, lds);
}

// grid: one workgroup per (16 RT)-row tile over the launch's chains.  Two instantiations: 16-row tiles (RT = 1) while the launch has at most
// a tile per CU that way (more CUs pull the stream: the backward pass of 2 048 rows 19.6 us against 24.2), 32-row tiles (RT = 2) beyond
// (the three forward passes of an eager step of 2 048 rows: 192 tiles, 29.8 us against 31.5 on 384 16-row tiles).
template <int RT>
__device__ __forceinline__ void bfn_kernel_body(const BfnMulti& m, unsigned char* lds) {
    const int tile = (int)blockIdx.x;
    int qn = 0;
    while (qn + 1 < m.n && tile >= m.tile_start[qn + 1]) ++qn;
    const BfChain& p = m.c[qn];
    const int row0 = (tile - m.tile_start[qn]) * (16 * RT);
    // in_mode 3 (the lazily evaluated target rows): the row count is what the arg-max launch left on the device; workgroup 0 reports it
    // to the host (ChainArgs::count_mirror of mlp_chain.h: the adaptive sizing of later steps' target launch)
    const int n_rows = p.rows_dev ? min(p.rows, *p.rows_dev) : p.rows;
    if (tile == m.tile_start[qn] && threadIdx.x == 0 && p.rows_dev != nullptr && p.count_mirror != nullptr)
        *p.count_mirror = ((unsigned long long)p.count_tag << 32) | (unsigned int)*p.rows_dev;
    if (row0 >= n_rows) return;
    if (p.k0_steps == 1) bfn_dispatch<RT, 1>(p, m.n_blocks[qn], row0, n_rows, lds);
    else bfn_dispatch<RT, 2>(p, m.n_blocks[qn], row0, n_rows, lds);
}
__global__ __launch_bounds__(BFN_THREADS) void mlp_chain_bfn_kernel(BfnMulti m) {            // 32-row tiles
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BFN_XBUF_BYTES + BFN_BIAS_BYTES];
    kernarg_warm<sizeof(BfnMulti)>();
    bfn_kernel_body<2>(m, lds);
}
__global__ __launch_bounds__(BFN_THREADS) void mlp_chain_bfn16_kernel(BfnMulti m) {          // 16-row tiles
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 1 * BFN_XBUF_BYTES + BFN_BIAS_BYTES];
    kernarg_warm<sizeof(BfnMulti)>();
    bfn_kernel_body<1>(m, lds);
}

// (Measured and dropped in round 6: the same chain with FOUR sub-tiles per wave -- 64-row tiles, one activation buffer with a barrier
// on either side of the epilogue's writes -- as the engine of the flagship step's big launches: 92.1 us against 70.3 for the two forward
// passes, 45.1 against 41.8 for the backward pass; bfn_dispatch<BFN_RT_MAX, ...> still compiles, nothing launches it.)

}  // namespace morl
