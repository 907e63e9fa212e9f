// Layer-fused MLP engine, FEW-ROW forward variant (gfx950, wave64): 8-row tiles on v_mfma_f32_4x4x1_16b_f32.
//
// A no-grad forward chain over a few hundred to a few thousand rows -- the lazily evaluated target rows of an Envelope step
// (envelope_kernels.h: 1 355 rows at the flagship shape) -- is pure latency on the 16-row tiles of mlp_chain16.h: 85 workgroups
// on 85 of 256 CUs, each carrying 16 rows x 256 columns x K per layer through the MFMA pipes of ONE CU (3.4 us per layer,
// 29 us for the five layers whatever the row count up to 4 096).  The f32 MFMA rate is the same for every shape (64 flop / clk /
// SIMD), so the only way down is FEWER ROWS PER CU: the 16-block form 4x4x1 computes, with the A operand the same in every
// block, a 4-row x 64-column outer-product step per instruction (K = 1, 8 cycles) -- one lane per column, the four rows in the
// four accumulator registers.  A workgroup carries 8 rows (two independent accumulator sets per lane: the dependent-issue
// latency of the instruction is hidden, and the weights a lane streams are used twice): 170 tiles on 170 CUs at the flagship
// shape, 1.7 us of MFMA per layer and tile -- equal to what streaming a 256 x 256 layer from L2 into one CU costs at 64 B / clk,
// which is the bound that remains; the stream runs three 32-deep register sets ahead of the MFMAs, across layer boundaries.
//
//   wave w owns columns [64w, 64w + 64); lane l is column 64w + l and supplies B[k][64w + l]; lane l reads A[row = l & 3 (+ 4)]
//   [k0 + (l >> 2)] from LDS, one dword per 16 contraction indices (block broadcast: see mfma4).  The contraction order inside
//   a 16-group is the one mlp_chain16.h's
//   16x16x4 steps produce (t outer, kq inner: k = 16c + 4kq + t), so a row's result is bit-identical whichever of the two
//   tilings carried it.  Only the constant-stride K4 weight layout (ChainArgs::fast == 1) is implemented; narrow steps
//   (N <= 32: the Q head) split the contraction over the four waves -- same instruction, lane = output column, operand read
//   N-major -- with the same fma chains and the same wave-order sum as the 16-row tiles.  No saves (hidden activations, sign
//   bits), no masks: forward only.
#pragma once
#include "mlp_chain16.h"

namespace morl {

constexpr int C4_RG = 2;                    // row groups of 4 per workgroup
constexpr int C4_TM = 4 * C4_RG;            // rows per workgroup
constexpr int C4_LDK = CH_MAXW + 8;         // activation row stride = 8 banks mod 32: a group's 4 rows x 16 contraction indices are read
                                            // conflict-free, one dword per lane
constexpr int C4_CHUNK = 2;                 // 16-groups per register set of B (32 contraction indices)
constexpr int C4_RING = 4;                  // register sets: the weight stream runs three sets ahead of the MFMAs
static_assert(4 % C4_CHUNK == 0 && C4_RING >= 4 / C4_CHUNK, "a narrow step's 64-deep operand must fit the ring");

// A operand: lane l holds A[row l & 3][k0 + (l >> 2)] -- block j = l >> 2 of the 16 carries contraction index k0 + j -- and the
// instruction's block broadcast (cbsz = 4: one block's A for all sixteen, abid = which) picks the index: ONE LDS dword per lane
// feeds sixteen MFMA steps.  (The first version read a b128 quad per lane and 4 indices: 8 KB of LDS reads per wave and group,
// as many LDS cycles as MFMA cycles.)
template <int ABID>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }

struct C4BSet {
    float4 v[4 * C4_CHUNK];     // v[4g + kq] = B[32ch + 16g + 4kq .. + 3][column of this lane]
};

// K4 layout (mlp_chain2.h): element (k, n) at ((k >> 2) * 256 + n) * 4 + (k & 3); k4-rows beyond the matrix read as zeros
__device__ __forceinline__ void c4_load(C4BSet& s, const ChainStep& st, int col, int chunk) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)st.Bmat, 0, (st.kpad > st.K ? st.kpad : st.K) * 256 * 4, 0x00020000);
#pragma unroll
    for (int q = 0; q < 4 * C4_CHUNK; ++q)
        s.v[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, col * 16, (4 * C4_CHUNK * chunk + q) * 256 * 16, 0));
}

__device__ __forceinline__ bool c4_wide(const ChainStep& st) { return st.N > 32; }
// chunks of a wide step: K padded to 64 with zero rows (the weight matrices are), then to a whole turn of the ring -- the slot of a
// step's first chunk is then always 0 (beyond the matrix the buffer loads return zeros without touching memory; the activation
// columns they meet are zero-filled by the input stage, and exist for K >= 128 only there)
__device__ __forceinline__ int c4_chunks(const ChainStep& st) {
    const int c = ((st.K + 63) >> 6) * (4 / C4_CHUNK);
    return (c + C4_RING - 1) / C4_RING * C4_RING;
}

// The weight stream: the 32-deep chunks of consecutive wide steps in order.  Streaming a 256 x 256 layer into one CU takes as
// long as its MFMAs (64 B / clk against 8 rows x 256 x 256 MACs); with ONE set in flight every chunk also paid the L2 latency
// (24 us for the flagship chain instead of 28 on 16-row tiles: still latency); three sets ahead hide it.
struct C4Stream {
    int step, chunk, end;       // next chunk to load; step == end: exhausted (end: first step that is not wide, or n_steps)
};
__device__ __forceinline__ void c4_stream_load(C4BSet& s, C4Stream& w, const ChainArgs& p, int col) {
    if (w.step >= w.end) return;                        // (workgroup-uniform)
    c4_load(s, p.step[w.step], col, w.chunk);
    if (++w.chunk >= c4_chunks(p.step[w.step])) { ++w.step; w.chunk = 0; }
}

// one 16-group: the 16 steps in mlp_chain16.h's order (t outer, kq inner: index 4 kq + t)
template <int T, int KQ>
__device__ __forceinline__ void c4_group_steps(const float4 (&b)[4], const float (&a)[C4_RG], f32x4 (&acc)[C4_RG]) {
    const float bv = c16_elem(b[KQ], T);
#pragma unroll
    for (int rg = 0; rg < C4_RG; ++rg) acc[rg] = mfma4<4 * KQ + T>(a[rg], bv, acc[rg]);
    if constexpr (KQ < 3) c4_group_steps<T, KQ + 1>(b, a, acc);
    else if constexpr (T < 3) c4_group_steps<T + 1, 0>(b, a, acc);
}

// one chunk: C4_CHUNK groups x 16 MFMA steps x C4_RG accumulator sets; pa: cur + (lane & 3) * C4_LDK + (lane >> 2)
__device__ __forceinline__ void c4_compute(const C4BSet& b, const float* pa, int k0, f32x4 (&acc)[C4_RG]) {
#pragma unroll
    for (int g = 0; g < C4_CHUNK; ++g) {
        float a[C4_RG];
#pragma unroll
        for (int rg = 0; rg < C4_RG; ++rg) a[rg] = pa[rg * 4 * C4_LDK + k0 + 16 * g];
        const float4 bq[4] = {b.v[4 * g + 0], b.v[4 * g + 1], b.v[4 * g + 2], b.v[4 * g + 3]};
        c4_group_steps<0, 0>(bq, a, acc);
    }
}

// the chunks of one wide step (a multiple of C4_RING: its first chunk is in slot 0)
__device__ __forceinline__ void c4_wide_chunks(C4BSet (&ring)[C4_RING], C4Stream& w, const ChainArgs& p, int col, const float* pa,
                                               int n_chunks, f32x4 (&acc)[C4_RG]) {
    for (int ch = 0; ch < n_chunks; ch += C4_RING) {
#pragma unroll
        for (int u = 0; u < C4_RING; ++u) {
            c4_stream_load(ring[(u + C4_RING - 1) % C4_RING], w, p, col);       // the set consumed last is free
            c4_compute(ring[u], pa, 16 * C4_CHUNK * (ch + u), acc);
        }
    }
}

// rows [row0, row0 + 8) of the chain; sAct: two activation buffers of C4_TM x C4_LDK floats, sRed: 4 x C4_TM x 32 floats.
// `flat`: in_mode 3, the pair of this thread's input row (loaded by the caller next to the row count)
__device__ __forceinline__ void mlp_chain4_body(const ChainArgs& p, int row0, int n_rows, int flat, float* sAct, float* sRed) {
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int col = wave * 64 + lane;
    float* cur = sAct;
    float* nxt = sAct + C4_TM * C4_LDK;

    C4BSet ring[C4_RING];
    C4Stream w;
    w.step = 0; w.chunk = 0; w.end = 0;
    while (w.end < p.n_steps && c4_wide(p.step[w.end])) ++w.end;       // (the first step is wide: host guarantee)
#pragma unroll
    for (int u = 0; u < C4_RING - 1; ++u) c4_stream_load(ring[u], w, p, col);

    // ---- input tile -> cur[m][k], zero-padded to the first step's contraction length -------------------------------------
    {
        const bool cat_mode = p.in_mode == 0 || p.in_mode == 3;
        const int K0 = cat_mode ? (p.D + p.R) : p.K0;
        const int K0pad = min(CH_MAXW, c4_chunks(p.step[0]) * 16 * C4_CHUNK);     // what the first step's chunks read
        const int m = tid >> 5, q = tid & 31;          // 32 threads per row
        const int row = row0 + m;
        const bool row_ok = row < n_rows;
        int b = row, wj = row;
        if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; wj = row - b * p.W; }
            else if (p.row_order == 1) { wj = row / p.B; b = row - wj * p.B; }
        } else if (p.in_mode == 3) {
            const int f = row_ok ? flat : 0;
            b = f / p.W; wj = f - b * p.W;
        }
        const float* src_a = cat_mode ? p.obs + (size_t)b * p.D : p.src + (size_t)row * p.ldsrc;
        const float* src_w = p.weights + (size_t)wj * p.R;
        for (int k = q; k < K0pad; k += 32) {
            float x = 0.f;
            if (row_ok && k < K0) x = cat_mode ? ((k < p.D) ? src_a[k] : src_w[k - p.D]) : src_a[k];
            cur[m * C4_LDK + k] = x;
        }
    }
    __syncthreads();

    for (int s = 0; s < p.n_steps; ++s) {
        const ChainStep& st = p.step[s];
        const int K = st.K, N = st.N;
        const bool feed_next = (s + 1 < p.n_steps);
        const ChainStep& ns = p.step[feed_next ? s + 1 : s];
        const bool nxt_narrow = feed_next && !c4_wide(ns);
        const float* pa = cur + (lane & 3) * C4_LDK + (lane >> 2);

        if (c4_wide(st)) {
            // ======================= matrix-core path: 4 rows x 64 columns x 1 per instruction =============================
            f32x4 acc[C4_RG];
#pragma unroll
            for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rg][r] = 0.f;
            const int n_chunks = c4_chunks(st);
            c4_wide_chunks(ring, w, p, col, pa, n_chunks, acc);
            if (nxt_narrow) {
                // the stream ends here (every set is free): the next step's N-major operand rows ride under the epilogue and the barrier
                const float* pb = ns.Bt + (size_t)lane * ns.ldbt + wave * 64;
#pragma unroll
                for (int h = 0; h < 4 / C4_CHUNK; ++h)
#pragma unroll
                    for (int q = 0; q < 4 * C4_CHUNK; ++q) {
                        const int ko = 16 * C4_CHUNK * h + 4 * q;
                        ring[h].v[q] = (lane < ns.N && wave * 64 + ko < ns.K) ? *reinterpret_cast<const float4*>(pb + ko)
                                                                              : make_float4(0.f, 0.f, 0.f, 0.f);       // K is a multiple of 4
                    }
            }
            // ---- epilogue: bias, ReLU -> the other activation buffer (and / or HBM for a wide last step) ------------------
            const float bias = (st.bias != nullptr && col < N) ? st.bias[col] : 0.f;
#pragma unroll
            for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[rg][r] + bias;
                    if (st.relu) x = fmaxf(x, 0.f);
                    x = (col < N) ? x : 0.f;
                    const int m = rg * 4 + r;
                    if (feed_next) nxt[m * C4_LDK + col] = x;
                    if (st.out != nullptr && col < N && row0 + m < n_rows) st.out[(size_t)(row0 + m) * st.ldout + col] = x;
                }
        } else {
            // ======================= narrow step: split-K over the four waves ================================================
            // wave w contracts k in [64w, 64w + 64) with the same instruction -- lane l < N is output column l, its operand row
            // Bt[l][.] read N-major (into the first sets of the ring, by the wide step in front) -- i.e. the fma chains of
            // mlp_chain16.h's narrow step (t outer / kq inner inside a 16-group); the four partial tiles are summed through LDS
            // in wave order
            {
                f32x4 acc[C4_RG];
#pragma unroll
                for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rg][r] = 0.f;
#pragma unroll
                for (int h = 0; h < 4 / C4_CHUNK; ++h) c4_compute(ring[h], pa, wave * 64 + 16 * C4_CHUNK * h, acc);
                if (lane < N)
#pragma unroll
                    for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sRed[wave * (C4_TM * 32) + (rg * 4 + r) * N + lane] = acc[rg][r];
            }
            const int n_out = C4_TM * N;
            // a wide step behind this one: its stream starts over
            w.step = s + 1; w.chunk = 0; w.end = s + 1;
            while (w.end < p.n_steps && c4_wide(p.step[w.end])) ++w.end;
#pragma unroll
            for (int u = 0; u < C4_RING - 1; ++u) c4_stream_load(ring[u], w, p, col);
            __syncthreads();
            for (int o = tid; o < n_out; o += CH_THREADS) {
                const int m = o / N, n = o - m * N;
                float v = sRed[o];
                v += sRed[1 * (C4_TM * 32) + o];
                v += sRed[2 * (C4_TM * 32) + o];
                v += sRed[3 * (C4_TM * 32) + o];
                v += (st.bias != nullptr) ? st.bias[n] : 0.f;
                if (st.relu) v = fmaxf(v, 0.f);
                const bool ok = row0 + m < n_rows;
                if (!ok) v = 0.f;
                if (feed_next) nxt[m * C4_LDK + n] = v;
                if (st.out != nullptr && ok) st.out[(size_t)(row0 + m) * st.ldout + n] = v;
            }
            if (feed_next)      // the next step reads K' = N <= 32 padded to 64 columns
                for (int e = tid; e < C4_TM * 64; e += CH_THREADS) {
                    const int m = e >> 6, k = e & 63;
                    if (k >= N) nxt[m * C4_LDK + k] = 0.f;
                }
        }
        __syncthreads();          // nxt complete, every wave past its last read of cur
        float* t = cur; cur = nxt; nxt = t;
    }
}

// One workgroup per 8-row tile of ONE chain (single network: nb <= 1)
static __global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain4_kernel(ChainArgs p) {
    __shared__ __attribute__((aligned(16))) float sAct[2 * C4_TM * C4_LDK];
    __shared__ float sRed[4 * C4_TM * 32];
    const int row0 = (int)blockIdx.x * C4_TM;
    // the row count and this thread's pair are fetched together (a stale list entry is a pair of an earlier step: in range once
    // clamped; it is only used if the row turns out to exist)
    const int n_pairs = p.B * p.W;
    int flat = (p.in_mode == 3) ? p.pairs[min(row0 + ((int)threadIdx.x >> 5), p.rows - 1)] : 0;
    const int n_rows = p.rows_dev ? min(p.rows, *p.rows_dev) : p.rows;        // (device-side count: in_mode 3)
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.rows_dev != nullptr && p.count_mirror != nullptr)     // (see ChainArgs::count_mirror)
        *p.count_mirror = ((unsigned long long)p.count_tag << 32) | (unsigned int)*p.rows_dev;
    if (row0 >= n_rows) return;                                               // (workgroup-uniform)
    flat = min(max(flat, 0), n_pairs - 1);
    mlp_chain4_body(p, row0, n_rows, flat, sAct, sRed);
}

constexpr int C4_MAX_ROWS = 2048;           // (general forward chains: one tile per CU at most; the lazy target rows always take these tiles)

// Host side: can this chain run on the 8-row tiles?
inline bool chain4_ok(const ChainArgs& a) {
    if (a.fast != 1 || a.n_steps < 1 || a.nb > 1 || a.x0_out != nullptr) return false;
    if (!(a.in_mode == 0 || a.in_mode == 3)) return false;          // (a dense input matrix, in_mode 1, is handled by the body but has no caller yet)
    if (!(a.step[0].N > 32)) return false;
    for (int s = 0; s < a.n_steps; ++s) {
        const ChainStep& st = a.step[s];
        if (st.mask != nullptr || st.bits_in != nullptr || st.bits_out != nullptr) return false;
        if (st.out != nullptr && s + 1 < a.n_steps) return false;            // no hidden saves
        if (st.N > 32 && (st.ldb != 256 || (st.kpad & 63) != 0)) return false;
        if (st.N <= 32 && (st.Bt == nullptr || (st.K & 3) != 0 || (st.ldbt & 3) != 0 || (reinterpret_cast<uintptr_t>(st.Bt) & 15) != 0)) return false;
        if (st.K > CH_MAXW || st.N > CH_MAXW) return false;
        if (st.N <= 32 && !(a.step[s - 1].N > 32)) return false;            // a narrow step's operand is fetched by the wide step in front of it
        // a narrow step is the LAST one: a wide step behind it would contract over a whole ring turn (128 columns) of which the
        // narrow step only rewrote / zeroed the first 64 -- stale LDS times zero weights, i.e. NaN if the stale value is Inf
        if (st.N <= 32 && s + 1 < a.n_steps) return false;
    }
    return true;
}

}  // namespace morl
