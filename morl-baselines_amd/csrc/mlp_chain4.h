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
// shape.  A 256 x 256 layer is 512 MFMAs per wave -- 2.4 us at the instruction's measured issue rate (4.7 ns with two or more
// accumulators, profiles/r03_mfma_4x4x1_probe.txt) -- beside 1.8 us of weight stream (256 KB into one CU at 64 B / clk:
// tools/probes/stream_probe.hip reaches 145 GB/s per CU with 8 or more loads in flight per lane); the stream runs three 32-deep
// register sets ahead of the MFMAs, across layer boundaries.
//
// Round 5, from wall-clock stamps of one tile (-DC4_PROF, `MORL_C4_PROF=1 python build.py`; us after kernel entry, flagship chain
// 35 -> 256 x 4 -> 18, before -> after): prologue to the first MFMA 3.6 -> 2.9, input step 2.4 -> 1.5, each 256 x 256 step
// 3.5 - 4.4 -> 3.3 (MFMA loop 2.85, epilogue 0.36, barrier 0.1), head 1.8 -> 1.1; mlp_chain4_kernel 23.2 -> 18.7 us inside the
// step.  What was wrong was not the MFMA loop: (1) the stream's loads stood under `if (stream not exhausted)`, so the compiler's
// wait-count pass drained the ring once per turn; (2) every chunk fetched its step descriptor with scalar loads in the loop --
// a scalar load's out-of-order return turns every LDS wait behind it into a full one; (3) each layer's bias was a global load
// behind the stream (loads return in order: a drain plus a round trip per layer); (4) wide and narrow steps shared one loop body
// and the ring went through register copies behind a full drain at every step boundary; (5) the prologue tested 45 optional
// fetches one by one, a scalar round trip and a branch each.  What remains is the instruction's issue rate (the loops run at
// 0.84 of it) and ~6 us of launch, argument, input and head latency around four layers.
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
static_assert(4 % C4_CHUNK == 0, "a narrow step's 64-deep operand (staged in LDS by the prologue) is consumed in whole chunks");
static_assert(C4_TM * 32 <= CH_THREADS, "a narrow step's outputs (8 rows x at most 32 columns): one per thread");

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
__device__ __forceinline__ void c4_load(C4BSet& s, __amdgpu_buffer_rsrc_t rsrc, int col, int chunk) {
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
//
// Every call issues its eight loads UNCONDITIONALLY -- past the last wide step from a zero-sized buffer (zeros, no memory access)
// -- and the descriptor of a step (base, size, chunk count: scalar loads from the argument block) is fetched one whole step ahead.
// (Round 5: with the loads under `if (stream not exhausted)` the compiler's wait-count pass has to assume the branch not taken
// and drained the whole ring -- s_waitcnt vmcnt(7..0) instead of vmcnt(23..16) -- once per turn of the ring, and the per-chunk
// descriptor fetch stalled the wave's issue for a scalar-cache round trip in front of every chunk.)
struct C4Desc {
    __amdgpu_buffer_rsrc_t rsrc;
    int n_chunks;
};
__device__ __forceinline__ C4Desc c4_desc(const ChainArgs& p, int s, int end) {
    const bool live = s < end;                          // (workgroup-uniform)
    const ChainStep& st = p.step[live ? s : 0];
    C4Desc d;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)st.Bmat, 0, live ? (st.kpad > st.K ? st.kpad : st.K) * 256 * 4 : 0, 0x00020000);
    d.n_chunks = live ? c4_chunks(st) : (1 << 30);
    return d;
}
struct C4Stream {
    C4Desc cur, nxt;
    int step, chunk, end;       // next chunk to load; step >= end: exhausted (end: first step that is not wide, or n_steps)
};
__device__ __forceinline__ void c4_stream_begin(C4Stream& w, const ChainArgs& p, int n_wide) {
    w.step = 0; w.chunk = 0; w.end = n_wide;
    w.cur = c4_desc(p, 0, n_wide);
    w.nxt = c4_desc(p, 1, n_wide);
}
// (no scalar fetch in here: a scalar load's out-of-order return makes every LDS wait behind it a full one -- the compiler can
// no longer count -- and this sits inside the MFMA loop; `nxt` is refreshed once per step by c4_wide_step.  The stream runs three
// chunks ahead and a step has at least four: it crosses exactly one step boundary per step.)
__device__ __forceinline__ void c4_stream_load(C4BSet& s, C4Stream& w, int col) {
    c4_load(s, w.cur.rsrc, col, w.chunk);
    if (++w.chunk >= w.cur.n_chunks) {                  // (workgroup-uniform)
        w.cur = w.nxt;
        ++w.step; w.chunk = 0;
    }
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

// The A values of one chunk: one LDS dword per lane, 16-group and row group; pa: cur + (lane & 3) * C4_LDK + (lane >> 2)
struct C4A {
    float v[C4_CHUNK][C4_RG];
};
__device__ __forceinline__ void c4_read_a(C4A& a, const float* pa, int k0) {
#pragma unroll
    for (int g = 0; g < C4_CHUNK; ++g)
#pragma unroll
        for (int rg = 0; rg < C4_RG; ++rg) a.v[g][rg] = pa[rg * 4 * C4_LDK + k0 + 16 * g];
}

// one chunk: C4_CHUNK groups x 16 MFMA steps x C4_RG accumulator sets.  `a` holds this chunk's A values on entry and the NEXT
// chunk's (k0 + 32 ..) on return: their LDS reads are issued in front of this chunk's MFMAs (read next to their use, every
// chunk's 64 MFMAs waited for an LDS round trip first: 0.4 us of a 256 x 256 layer's 3.2).  Behind a step's last chunk the read
// runs up to 32 + 15 floats past the row -- into the next row, buffer or array of C4Shared -- and is dropped.  The order -- this
// segment's weight loads (the caller's), the next chunk's A reads, then the MFMAs -- is pinned: left alone the scheduler sinks
// the loads into the MFMAs and the stream runs one set ahead instead of three.
__device__ __forceinline__ void c4_compute(const C4BSet& b, const float* pa, int k0, f32x4 (&acc)[C4_RG], C4A& a) {
    const C4A cur = a;
    c4_read_a(a, pa, k0 + 16 * C4_CHUNK);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < C4_CHUNK; ++g) {
        const float av[C4_RG] = {cur.v[g][0], cur.v[g][1]};
        const float4 bq[4] = {b.v[4 * g + 0], b.v[4 * g + 1], b.v[4 * g + 2], b.v[4 * g + 3]};
        c4_group_steps<0, 0>(bq, av, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
}
static_assert(C4_RG == 2, "c4_compute spells out the two row groups");

// the chunks of one wide step (a multiple of C4_RING: its first chunk is in slot 0)
__device__ __forceinline__ void c4_wide_chunks(C4BSet (&ring)[C4_RING], C4Stream& w, int col, const float* pa,
                                               int n_chunks, int k_live, f32x4 (&acc)[C4_RG]) {
    C4A a;
    c4_read_a(a, pa, 0);
    for (int ch = 0; ch < n_chunks; ch += C4_RING) {
#pragma unroll
        for (int u = 0; u < C4_RING; ++u) {
            c4_stream_load(ring[(u + C4_RING - 1) % C4_RING], w, col);       // the set consumed last is free
            // (chunks that only pad the step to a whole turn of the ring -- K <= 64: the input step -- are loaded, as zeros from
            // beyond the matrix, to keep the ring turning, but not multiplied)
            if (16 * C4_CHUNK * (ch + u) < k_live) c4_compute(ring[u], pa, 16 * C4_CHUNK * (ch + u), acc, a);
        }
    }
}

constexpr int C4_NAR_LD = 68;               // floats per lane of a narrow step's staged operand (= 4 banks mod 32: the b128 reads are conflict-free)
struct C4Shared {
    float act[2 * C4_TM * C4_LDK];          // two activation buffers
    float red[4 * C4_TM * 32];              // a narrow step's four partial tiles
    float bias[MORL_MAX_LAYERS * CH_MAXW];  // bias[s][column] of the wide steps (zeros beyond N)
    float nar[4 * 32 * C4_NAR_LD];          // nar[wave][lane < N][k - 64 wave]: a narrow last step's operand rows
};

// One wide step of a tile: the MFMA loop over the step's chunks (the stream running ahead into the next wide step), bias / ReLU,
// the result into the other activation buffer (`feed`) and / or HBM.  The bias comes from LDS (staged by the prologue): a global
// load here would sit at the tail of the weight stream's queue -- loads return in order, so waiting for it drains the stream and
// adds its own round trip, once per layer (round 5: 3.5 - 4.4 us per 256 x 256 layer against 1.8 us of stream).
#ifdef C4_PROF
#define C4_PROF_PARAMS , long long (&pt)[40], int& pti
#define C4_PROF_ARGS , pt, pti
#define C4_T() { if (pti < 40) pt[pti++] = wall_clock64(); }
#else
#define C4_PROF_PARAMS
#define C4_PROF_ARGS
#define C4_T()
#endif
__device__ __forceinline__ void c4_wide_step(const ChainArgs& p, int s, bool feed, C4BSet (&ring)[C4_RING], C4Stream& w, int col,
                                             const float* cur, float* nxt, const float* sBias, int row0, int n_rows C4_PROF_PARAMS) {
    const int lane = lane_id();
    const ChainStep& st = p.step[s];
    const int N = st.N;
    const float* pa = cur + (lane & 3) * C4_LDK + (lane >> 2);
    f32x4 acc[C4_RG];
#pragma unroll
    for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rg][r] = 0.f;
    const int n_chunks = c4_chunks(st);
    w.nxt = c4_desc(p, s + 1, w.end);           // the step the stream moves on to during this one
    const bool relu = st.relu != 0;             // (the scalar fields the epilogue needs: fetched in front of the MFMA loop)
    float* const outp = st.out;
    const int ldout = st.ldout;
    C4_T()
    c4_wide_chunks(ring, w, col, pa, n_chunks, ((st.K + 63) >> 6) << 6, acc);
    C4_T()
    const float bias = sBias[s * CH_MAXW + col];
#pragma unroll
    for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = acc[rg][r] + bias;
            if (relu) x = fmaxf(x, 0.f);
            x = (col < N) ? x : 0.f;
            const int m = rg * 4 + r;
            if (feed) nxt[m * C4_LDK + col] = x;
            if (outp != nullptr && col < N && row0 + m < n_rows) outp[(size_t)(row0 + m) * ldout + col] = x;
        }
}

// rows [row0, row0 + 8) of the chain.  `flat`: in_mode 3, the pair of this thread's input row (loaded by the caller next to the
// row count).
//
// Everything a tile needs besides the weight stream is fetched by the PROLOGUE, in front of the stream's first sets -- the input
// columns (their addresses hang off the pair, itself the end of a dependent chain of fetches: argument block -> list -> pair),
// every wide step's bias, a narrow last step's operand rows and bias -- and parked in LDS / registers; from there on the only
// vector-memory traffic is the stream, which then never drains before the chain's end.
__device__ __forceinline__ void mlp_chain4_body(const ChainArgs& p, int row0, int n_rows, int flat, C4Shared& sh) {
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int col = wave * 64 + lane;
    float* cur = sh.act;
    float* nxt = sh.act + C4_TM * C4_LDK;

#ifdef C4_PROF
    long long pt[40];
    int pti = 0;
#endif
    C4_T()
    // ---- prologue fetches: straight-line code -----------------------------------------------------------------------------
    // No branch depends on a scalar field in here: every optional fetch is a buffer load whose descriptor's size says whether
    // (and for which lanes) it touches memory, so the scalar fields of all eight step descriptors arrive in ONE batch of scalar
    // loads -- the first form tested each pointer / count in turn, a scalar-cache round trip and a branch per vector load (45 of
    // them in a row: 5 us from kernel entry to the first MFMA).
    // chain4_ok(): step 0 is wide, a narrow step can only be the last one; steps beyond n_steps count as absent
    int stN[MORL_MAX_LAYERS];
    int n_wide = 0;
#pragma unroll
    for (int s = 0; s < MORL_MAX_LAYERS; ++s) {
        stN[s] = (s < p.n_steps) ? p.step[s].N : 0;
        n_wide += (stN[s] > 32) ? 1 : 0;
    }
    const bool narrow_last = n_wide < p.n_steps;

    const bool cat_mode = p.in_mode == 0 || p.in_mode == 3;
    const int K0 = cat_mode ? (p.D + p.R) : p.K0;
    const int in_m = tid >> 5, in_q = tid & 31;          // 32 threads per input row
    const float* src_a;
    const float* src_w;
    bool row_ok;
    {
        const int row = row0 + in_m;
        row_ok = row < n_rows;
        int b = row, wj = row;
        if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; wj = row - b * p.W; }
            else if (p.row_order == 1) { wj = row / p.B; b = row - wj * p.B; }
        } else if (p.in_mode == 3) {
            const int f = row_ok ? flat : 0;
            b = f / p.W; wj = f - b * p.W;
        }
        src_a = cat_mode ? p.obs + (size_t)b * p.D : p.src + (size_t)row * p.ldsrc;
        src_w = p.weights + (size_t)wj * p.R;
    }
    auto in_elem = [&](int k) -> float {
        float x = 0.f;
        if (row_ok && k < K0) x = cat_mode ? ((k < p.D) ? src_a[k] : src_w[k - p.D]) : src_a[k];
        return x;
    };
    const float x0 = in_elem(in_q), x1 = in_elem(in_q + 32);
    float bias_w[MORL_MAX_LAYERS];
#pragma unroll
    for (int s = 0; s < MORL_MAX_LAYERS; ++s) {
        const float* bp = p.step[s].bias;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, (bp != nullptr && stN[s] > 32) ? stN[s] * 4 : 0, 0x00020000);
        bias_w[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, col * 4, 0, 0));       // zero beyond N / without a bias
    }
    // a narrow last step (N <= 32 outputs): lane l < N of wave w contracts k in [64w, 64w + 64) with row l of the N-major operand
    float4 nar[16];
    float nbias;
    int nN;
    {
        const ChainStep& ns = p.step[min(n_wide, MORL_MAX_LAYERS - 1)];
        nN = narrow_last ? max(ns.N, 1) : 1;
        constexpr int OOR = 0x40000000;              // beyond any descriptor's size: the load returns zeros
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)ns.Bt, 0, narrow_last ? nN * ns.ldbt * 4 : 0, 0x00020000);
        const int lim = ns.K - wave * 64;            // K is a multiple of 4
        const int voff = (lane * ns.ldbt + wave * 64) * 4;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            nar[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, (4 * q < lim) ? voff + 16 * q : OOR, 0, 0));
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)ns.bias, 0, (narrow_last && ns.bias != nullptr) ? nN * 4 : 0, 0x00020000);
        nbias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rn, (tid < C4_TM * nN) ? (tid % nN) * 4 : OOR, 0, 0));
    }

    C4BSet ring[C4_RING];
    C4Stream w;
    c4_stream_begin(w, p, n_wide);
#pragma unroll
    for (int u = 0; u < C4_RING - 1; ++u) c4_stream_load(ring[u], w, col);

    // ---- park them: input tile -> cur[m][k], zero-padded to the first step's contraction length; biases; the narrow operand ----
    {
        const int K0pad = min(CH_MAXW, c4_chunks(p.step[0]) * 16 * C4_CHUNK);     // what the first step's chunks read: >= 128
        cur[in_m * C4_LDK + in_q] = x0;
        cur[in_m * C4_LDK + in_q + 32] = x1;
        if (K0 > 64) {                                   // (workgroup-uniform; the wait for these loads drains the ring)
            for (int k = in_q + 64; k < K0pad; k += 32) cur[in_m * C4_LDK + k] = in_elem(k);
        } else {
            for (int k = in_q + 64; k < K0pad; k += 32) cur[in_m * C4_LDK + k] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < MORL_MAX_LAYERS; ++s)
            sh.bias[s * CH_MAXW + col] = bias_w[s];                               // (read back by this very thread)
        if (narrow_last && lane < 32) {
            float* d = sh.nar + (wave * 32 + lane) * C4_NAR_LD;
#pragma unroll
            for (int q = 0; q < 16; ++q) *reinterpret_cast<float4*>(d + 4 * q) = nar[q];
        }
    }
    __syncthreads();
    C4_T()

    // ---- the wide steps -----------------------------------------------------------------------------------------------------
    for (int s = 0; s < n_wide; ++s) {
        const bool feed = s + 1 < p.n_steps;
        c4_wide_step(p, s, feed, ring, w, col, cur, nxt, sh.bias, row0, n_rows C4_PROF_ARGS);
        C4_T()
        if (feed) __syncthreads();          // nxt complete, every wave past its last read of cur
        C4_T()
        float* t = cur; cur = nxt; nxt = t;
    }

    if (narrow_last) {
        // ======================= narrow last step: split-K over the four waves ================================================
        // wave w contracts k in [64w, 64w + 64) with the same instruction -- lane l < N is output column l, its operand row
        // Bt[l][.] read N-major (staged in LDS by the prologue) -- i.e. the fma chains of mlp_chain16.h's narrow step (t outer /
        // kq inner inside a 16-group); the four partial tiles are summed through LDS in wave order
        const ChainStep& st = p.step[n_wide];
        const int N = nN;
        const float* pa = cur + (lane & 3) * C4_LDK + (lane >> 2);
        const int n_out = C4_TM * N;               // <= CH_THREADS: one output per thread
        const int o_m = tid / N, o_n = tid - o_m * N;
        const bool relu = st.relu != 0;
        float* const outp = st.out;
        const int ldout = st.ldout;
        {
            C4BSet nb[4 / C4_CHUNK];
            const float* d = sh.nar + (wave * 32 + (lane & 31)) * C4_NAR_LD;
#pragma unroll
            for (int h = 0; h < 4 / C4_CHUNK; ++h)
#pragma unroll
                for (int q = 0; q < 4 * C4_CHUNK; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(d + 16 * C4_CHUNK * h + 4 * q);
                    nb[h].v[q] = (lane < 32) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            f32x4 acc[C4_RG];
#pragma unroll
            for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rg][r] = 0.f;
            C4A a;
            c4_read_a(a, pa, wave * 64);
#pragma unroll
            for (int h = 0; h < 4 / C4_CHUNK; ++h) c4_compute(nb[h], pa, wave * 64 + 16 * C4_CHUNK * h, acc, a);
            if (lane < N)
#pragma unroll
                for (int rg = 0; rg < C4_RG; ++rg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sh.red[wave * (C4_TM * 32) + (rg * 4 + r) * N + lane] = acc[rg][r];
        }
        __syncthreads();
        if (tid < n_out) {
            float v = sh.red[tid];
            v += sh.red[1 * (C4_TM * 32) + tid];
            v += sh.red[2 * (C4_TM * 32) + tid];
            v += sh.red[3 * (C4_TM * 32) + tid];
            v += nbias;
            if (relu) v = fmaxf(v, 0.f);
            if (outp != nullptr && row0 + o_m < n_rows) outp[(size_t)(row0 + o_m) * ldout + o_n] = v;
        }
        C4_T()
    }
#ifdef C4_PROF
    if (p.prof != nullptr && tid == 0) {
        long long* o = p.prof + (size_t)blockIdx.x * 48;
        o[0] = pti;
        for (int i = 0; i < pti; ++i) o[2 + i] = pt[i];
    }
#endif
}
#undef C4_T
#undef C4_PROF_PARAMS
#undef C4_PROF_ARGS

// One workgroup per 8-row tile of ONE chain (single network: nb <= 1)
static __global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain4_kernel(ChainArgs p) {
    __shared__ __attribute__((aligned(16))) C4Shared sh;
    const int row0 = (int)blockIdx.x * C4_TM;
    kernarg_warm<sizeof(ChainArgs)>();       // (the prologue reads the block in several dependent batches)
#ifdef C4_PROF
    if (p.prof != nullptr && threadIdx.x == 0) p.prof[48 * 4096 + 1 + blockIdx.x] = wall_clock64();      // kernel entry, before any argument is read
#endif
    // the row count and this thread's pair are fetched together (a stale list entry is a pair of an earlier step: in range once
    // clamped; it is only used if the row turns out to exist)
    const int n_pairs = p.B * p.W;
    int flat = (p.in_mode == 3) ? p.pairs[min(row0 + ((int)threadIdx.x >> 5), p.rows - 1)] : 0;
    const int n_rows = p.rows_dev ? min(p.rows, *p.rows_dev) : p.rows;        // (device-side count: in_mode 3)
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.rows_dev != nullptr && p.count_mirror != nullptr)     // (see ChainArgs::count_mirror)
        *p.count_mirror = ((unsigned long long)p.count_tag << 32) | (unsigned int)*p.rows_dev;
    if (row0 >= n_rows) return;                                               // (workgroup-uniform)
    flat = min(max(flat, 0), n_pairs - 1);
    mlp_chain4_body(p, row0, n_rows, flat, sh);
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
