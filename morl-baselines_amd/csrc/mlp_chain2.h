// Layer-fused MLP engine (gfx950, wave64), kernels; contract and argument structures in mlp_chain.h (a workgroup carries a
// TM-row tile through a whole chain of dense layers, activations resident in LDS).  Second generation, laid out around what
// round 1's profiles showed:
//
//   * activations are M-MAJOR in LDS, sAct[m][k] with row stride 260 floats: the MFMA A operand of lane (i, h) for FOUR
//     consecutive k-pairs is ONE ds_read_b128 (k = 8c + 4h + 0..3; the contraction order inside a chunk is permuted
//     accordingly -- any order is a valid exact-fp32 chain, and it is the same for every tile size, so a row's result does
//     not depend on the tile that carried it).  16 MFMAs per LDS read per row tile instead of 4, and the read of group g+1
//     is issued a whole group (>= 512 cycles) ahead of its first use.  Row stride 260 = 4 (mod 64) banks: the 16-lane
//     groups of a b128 access cover all 64 banks once.
//   * the epilogue stores column PAIRS (ds_write_b64, 32 per lane and layer instead of 128 scalar stores) -- rows of a
//     half-wave are contiguous 256-byte segments.
//   * activations / gradients that must also reach HBM (saved h_l of the training forward, g_l of the backward chain) are
//     NOT stored from the accumulators in the epilogue: they sit in LDS as the next step's input anyway, so they are copied
//     LDS -> HBM in 16-byte pieces (one wave = one full 1-KB row per instruction) INSIDE the next step's MFMA loop, where
//     the memory pipes are idle.
//   * weights stream L2 -> registers exactly as before (the waves split the output columns, nothing is shared): one
//     buffer_load_dwordx2 per k-row and lane, two named register sets; the per-load address is one v_add of a scalar.
//   * persistent schedule: the grid is 2 workgroups per CU; every workgroup takes 64-row tiles in full rounds and the
//     remainder as 32-row half tiles, so that all slots finish together (3 x 16 384 rows = 768 tiles on 512 slots used to
//     run as one full and one half-empty round), and the second half of the grid takes its jobs in the opposite order, which
//     staggers the two workgroups of a CU: one is in its MFMA loop while the other is in an epilogue.
#pragma once
#include "mlp_chain.h"

namespace morl {

constexpr int C2_LDK = CH_MAXW + 4;      // floats per activation row in LDS
constexpr int C2_TM = 64;                // rows of the LDS tile (32-row jobs use the first half)

struct C2BSet {
    float2 v[16];
};
struct C2ASet {
    float4 a[2];
};

// wide step: lane (i, h) of wave w loads B[k][64w + 2i .. +1]; v[j] <-> k = k0 + 8*(j >> 2) + 4h + (j & 3)
struct C2WideDesc {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane_off;     // bytes: (4h * ldb + colw) * 4, or CH_OOB when the lane's columns do not exist
    int stride;       // bytes per k-row (wave-uniform)
    int lane_off1;    // N-major stream (LD = 2) only: the lane's second column, see c2_wide_desc_n
};

__device__ __forceinline__ C2WideDesc c2_wide_desc(const ChainStep& st, int wave, int i, int h, int g = 0) {
    C2WideDesc d;
    const int colw = wave * 64 + 2 * i;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(st.Bmat + g * st.sW), 0, (st.kpad > st.K ? st.kpad : st.K) * st.ldb * 4, 0x00020000);
    d.lane_off = (colw < st.ldb) ? (4 * h * st.ldb + colw) * 4 : CH_OOB;
    d.stride = st.ldb * 4;
    d.lane_off1 = (h * 256 + colw) * 16;      // constant-stride stream (c2_load_fast): the lane's first 16-byte piece, K4 layout
    return d;
}

// N-major stream (LD = 2): the operand is read from the nn.Linear matrix itself, Bt [N][ldbt] (row n contiguous over k), so a
// forward pass needs no K-major shadow copy of its weights.  Lane (i, h) still multiplies k = k0 + 8c + 4h + t in MFMA step t of
// group c -- four consecutive k of one row = ONE 16-byte load per column and group (8 loads per set instead of 16): same
// values, same MFMA k order, bit-identical results.  A quad that runs past its row's K reads the next row's first weights
// (finite) against activation columns that are exactly zero there, past the matrix the range check returns 0.
__device__ __forceinline__ C2WideDesc c2_wide_desc_n(const ChainStep& st, int wave, int i, int h, int g = 0) {
    C2WideDesc d;
    const int colw = wave * 64 + 2 * i;
    // (K not a multiple of 4: the last row's last quad straddles the end of the matrix; the range covers the whole quad -- the
    // bias vector follows every matrix in the flat parameter layout -- so that no valid weight is cut by a partial range check)
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(st.Bt + g * st.sW), 0, (st.N * st.ldbt + ((st.K & 3) ? 4 : 0)) * 4, 0x00020000);
    d.lane_off = (colw < st.N) ? (colw * st.ldbt + 4 * h) * 4 : CH_OOB;
    d.lane_off1 = (colw + 1 < st.N) ? ((colw + 1) * st.ldbt + 4 * h) * 4 : CH_OOB;
    d.stride = 0;
    return d;
}

__device__ __forceinline__ void c2_load_nmajor(C2BSet& s, const C2WideDesc& d, int k0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kb = (k0 + 8 * c) * 4;
        const float4 q0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, d.lane_off + kb, 0, 0));
        const float4 q1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, d.lane_off1 + kb, 0, 0));
        s.v[4 * c + 0] = make_float2(q0.x, q1.x);
        s.v[4 * c + 1] = make_float2(q0.y, q1.y);
        s.v[4 * c + 2] = make_float2(q0.z, q1.z);
        s.v[4 * c + 3] = make_float2(q0.w, q1.w);
    }
}

// rows >= K lie beyond the resource: the hardware returns 0 (K padding, dummy prefetches) -- no branch, no select
__device__ __forceinline__ void c2_load_wide(C2BSet& s, const C2WideDesc& d, int k0) {
    const int base = d.lane_off + k0 * d.stride;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int off = base + (8 * (j >> 2) + (j & 3)) * d.stride;
        s.v[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(d.rsrc, off, 0, 0));
    }
}

// constant-stride form (ChainArgs::fast == 1: every wide step has ldb == 256 and kpad a multiple of 64, and the operand is in the
// K4 LAYOUT: element (k, n) of the K-major matrix at ((k >> 2) * 256 + n) * 4 + (k & 3) -- the four consecutive k a lane
// multiplies in one group are 16 contiguous bytes, its second column the next 16).  Two 16-byte loads per group and lane (8 per
// register set instead of 16 eight-byte ones), still 1 KB contiguous per half-wave and instruction; the per-lane offset is
// loop-invariant, the k4-row goes through the SGPR offset, the second column through the instruction's immediate -- no VALU
// work in the stream.  (Forward x1 64.3 -> 62.5 us, with saves 69.6 -> 66.5, backward 64.7 -> 62.7 on the probe.)
__device__ __forceinline__ void c2_load_fast(C2BSet& s, const C2WideDesc& d, int k0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int so = ((k0 >> 2) + 2 * c) * 4096;
        const float4 q0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, d.lane_off1, so, 0));
        const float4 q1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, d.lane_off1 + 16, so, 0));
        s.v[4 * c + 0] = make_float2(q0.x, q1.x);
        s.v[4 * c + 1] = make_float2(q0.y, q1.y);
        s.v[4 * c + 2] = make_float2(q0.z, q1.z);
        s.v[4 * c + 3] = make_float2(q0.w, q1.w);
    }
}

// narrow step (N <= 32): the four waves split the contraction, wave w: k in [64w, 64w + 64); lane (i = n, h) loads
// Bt[n][64w + 8c + 4h .. +3] for c = 0..7 as eight float4 = the 16 float2 of the set (v[2c], v[2c+1])
__device__ __forceinline__ void c2_load_narrow(C2BSet& s, const ChainStep& st, int wave, int i, int h, int g = 0) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(st.Bt + g * st.sW), 0, st.N * st.ldbt * 4, 0x00020000);
    const int kbase = wave * 64 + 4 * h;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = kbase + 8 * c;
        // K is a multiple of 4 (host guarantee), so a quad is wholly inside or wholly outside the row
        const int off = (i < st.N && k < st.K) ? (i * st.ldbt + k) * 4 : CH_OOB;
        s.v[2 * c] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
        s.v[2 * c + 1] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 8, 0, 0));
    }
}

// the same operand from the K-major matrix Bmat [K][ldb] when no N-major copy exists (st.Bt == NULL: the first-layer dX step of
// a backward chain reads the nn.Linear matrix [out = K][in = N] as it is): four dword loads instead of one 16-byte load
__device__ __forceinline__ void c2_load_narrow_k(C2BSet& s, const ChainStep& st, int wave, int i, int h, int g = 0) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(st.Bmat + g * st.sW), 0, st.K * st.ldb * 4, 0x00020000);
    const int kbase = wave * 64 + 4 * h;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = kbase + 8 * c;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = (i < st.N && k + t < st.K) ? ((k + t) * st.ldb + i) * 4 : CH_OOB;
            v[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
        }
        s.v[2 * c] = make_float2(v[0], v[1]);
        s.v[2 * c + 1] = make_float2(v[2], v[3]);
    }
}

// LDS -> HBM copy of the tile in sAct (the step's input = the previous step's output) in TM / 4 pieces: piece `it` moves
// rows 4*it .. 4*it + 3, one wave = one full row (64 x 16 bytes) per instruction.  Branch-free: the destination is a buffer
// resource spanning exactly out[rows][ldout], so rows beyond the matrix (ragged last tile) are dropped by the range check,
// and lanes whose columns do not exist carry an out-of-range offset.  Per piece: one ds_read_b128 with an immediate
// offset, one v_add, one buffer_store_dwordx4.
struct C2CopyDst {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff0;        // bytes: ((row0 + tid / 64) * ldout + 4 * (tid % 64)) * 4, or CH_OOB
    int step;         // bytes between pieces: 4 rows
    int lds0;         // floats: (tid / 64) * C2_LDK + 4 * (tid % 64)
};

__device__ __forceinline__ C2CopyDst c2_copy_dst(float* out, int ldout, int n, int rows, int row0, int tid) {
    C2CopyDst d;
    // The descriptor is workgroup-uniform, and the compiler must KNOW it (the network index in `out` comes out of an integer
    // division, i.e. out of the vector ALU): a descriptor it takes for divergent is stored through a "waterfall" loop -- one pass
    // per distinct value -- and a loop with a store in it inside the MFMA loop turns every wait for the weight stream into a
    // wait for (nearly) all of it, since stores and loads share the counter (round 5: the 16-row chain's look-ahead was lost here).
    const unsigned long long a = (unsigned long long)(uintptr_t)out;
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(a >> 32));
    out = (float*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, __builtin_amdgcn_readfirstlane(rows * ldout * 4), 0x00020000);
    const int m = tid >> 6, c4 = (tid & 63) << 2;
    d.voff0 = (c4 < n) ? ((row0 + m) * ldout + c4) * 4 : CH_OOB;
    d.step = 4 * ldout * 4;
    d.lds0 = m * C2_LDK + c4;
    return d;
}

__device__ __forceinline__ void c2_copy_piece(const float* sAct, const C2CopyDst& d, int it) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sAct + d.lds0 + it * 4 * C2_LDK);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c2_u32x4, v), d.rsrc, d.voff0 + it * d.step, 0, 0);
}

// ---- the MFMA loop of a wide step ---------------------------------------------------------------------------------------
// SCHED = 1 pins the instruction interleave with sched_group_barrier: the look-ahead operand reads FIRST in every group
// (hipcc otherwise sinks each ds_read next to its first use and waits for it on the spot), the weight loads and their address
// adds spread over the MFMAs of the other register set, one copy piece per group.
#define C2_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// LD: the weight stream -- 0 K-major, any stride; 1 K-major, constant stride (ChainArgs::fast == 1); 2 N-major (fast == 2)
template <int LD>
__device__ __forceinline__ void c2_load(C2BSet& s, const C2WideDesc& d, int k0) {
    if (LD == 1) c2_load_fast(s, d, k0);
    else if (LD == 2) c2_load_nmajor(s, d, k0);
    else c2_load_wide(s, d, k0);
}
template <int LD>
__device__ __forceinline__ C2WideDesc c2_desc(const ChainStep& st, int wave, int i, int h, int g) {
    return LD == 2 ? c2_wide_desc_n(st, wave, i, h, g) : c2_wide_desc(st, wave, i, h, g);
}

template <int TM, bool COPY, int SCHED, int LD>
__device__ __forceinline__ void c2_wide_loop(f32x16 (&acc)[TM / 32][2], C2BSet& bx, C2BSet& by, const float* sAct, const float* pa,
                                             const C2WideDesc& dcur, const C2WideDesc& dnext, bool nxt_wide, int n_pairs,
                                             const C2CopyDst& cd) {
    constexpr int MT = TM / 32;
    constexpr int N_PIECES = TM / 4;
    constexpr int PP = TM / 16;             // copy pieces per 64-deep pair: all of them within the 4 pairs of a 256-wide step
    C2ASet ax, ay;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) ax.a[tm] = *reinterpret_cast<const float4*>(pa + tm * 32 * C2_LDK);
    int piece = 0;
    for (int pr = 0; pr < n_pairs; ++pr) {
        const int k0 = pr * 64;
        const bool more = pr + 1 < n_pairs;
        // bx's refill: this step's chunk two ahead, else the next wide step's chunk 0, else a dummy re-read (a narrow step
        // loads its own operands); by's refill: this step's chunk three ahead, else a dummy.  Never skipped and branch-free,
        // so the loop body is one scheduling region and the number of loads in flight is static.
        const bool from_next = !more && nxt_wide;
        C2WideDesc dx;
        dx.rsrc = from_next ? dnext.rsrc : dcur.rsrc;
        dx.lane_off = from_next ? dnext.lane_off : dcur.lane_off;
        dx.stride = from_next ? dnext.stride : dcur.stride;
        dx.lane_off1 = from_next ? dnext.lane_off1 : dcur.lane_off1;
        const int kx = more ? k0 + 64 : 0;
        const int ky = more ? k0 + 96 : CH_BK;
// one group = 8 contraction indices = 4 MFMA k-pairs x (2 * MT) tiles; the A quad of the NEXT group is read first
#define C2_GROUP(SET, C, ACUR, ANEXT, KNEXT)                                                               \
    {                                                                                                      \
        _Pragma("unroll") for (int tm = 0; tm < MT; ++tm)                                                  \
            ANEXT.a[tm] = *reinterpret_cast<const float4*>(pa + tm * 32 * C2_LDK + (KNEXT));              \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
            const float2 bv = SET.v[4 * (C) + t];                                                          \
            _Pragma("unroll") for (int tm = 0; tm < MT; ++tm) {                                            \
                const float av = (t == 0) ? ACUR.a[tm].x : (t == 1) ? ACUR.a[tm].y : (t == 2) ? ACUR.a[tm].z : ACUR.a[tm].w; \
                acc[tm][0] = mfma32(av, bv.x, acc[tm][0]);                                                 \
                acc[tm][1] = mfma32(av, bv.y, acc[tm][1]);                                                 \
            }                                                                                              \
        }                                                                                                  \
    }
        C2_GROUP(bx, 0, ax, ay, k0 + 8)
        C2_GROUP(bx, 1, ay, ax, k0 + 16)
        C2_GROUP(bx, 2, ax, ay, k0 + 24)
        C2_GROUP(bx, 3, ay, ax, k0 + 32)
        c2_load<LD>(bx, dx, kx);
        C2_GROUP(by, 0, ax, ay, k0 + 40)
        C2_GROUP(by, 1, ay, ax, k0 + 48)
        C2_GROUP(by, 2, ax, ay, k0 + 56)
        // (the last group's look-ahead read stays inside the buffer: column k0 + 64 + 7 <= 263, 8 spare floats at its end)
        C2_GROUP(by, 3, ay, ax, k0 + 64)
        c2_load<LD>(by, dcur, ky);
#undef C2_GROUP
        if (COPY) {
#pragma unroll
            for (int u = 0; u < PP; ++u) c2_copy_piece(sAct, cd, piece + u);
            piece += PP;
        }
        if (SCHED) {
            // program order above: 8 x [MT ds_read, 8*MT MFMA], bx loads after group 3, by loads after group 7, 4 copy pieces.
            // wanted: groups 0-3 (bx) carry the copy pieces, groups 4-5 (by) carry bx's 16 loads, by's loads trail (they
            // have the whole next bx half to land)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                C2_SGB(0x100, MT);                              // look-ahead A reads
                if (COPY && g < PP) C2_SGB(0x100, 1);           // a copy piece's LDS read ...
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    C2_SGB(0x008, 2 * MT);
                    // bx's 16 refill loads, two per MFMA quad of groups 4 and 5 (one per quad over groups 4-7 measured
                    // 184 us for the three forward passes, this 181 us; issuing the copy stores after them instead of
                    // before -- vmcnt retires loads and stores in order -- measured worse, 187 us)
                    if (g == 4 || g == 5) { if (LD == 0) C2_SGB(0x002, 2); C2_SGB(0x020, LD == 0 ? 2 : 1); }
                }
                if (COPY && g < PP) { C2_SGB(0x002, 1); C2_SGB(0x040, 1); }   // ... its address add and its store
            }
        }
    }
    if (COPY)
        for (; piece < N_PIECES; ++piece) c2_copy_piece(sAct, cd, piece);
}

// PROF (development probes only, tools/probes/chain2_probe.hip): wave 0 stamps s_memtime at the phase boundaries of the job
#define C2_TICK() if (PROF) { if (n_tick < 24) ticks[n_tick] = clock64(); ++n_tick; }

template <int TM, int SCHED, int LD, bool PROF = false>
__device__ __forceinline__ void mlp_chain2_body(const ChainArgs& p, int row0, float* sAct, long long* prof_out = nullptr, int g = 0,
                                                int n_rows_dev = -1) {
    const int n_rows = n_rows_dev >= 0 ? n_rows_dev : p.rows;      // (in_mode 3: the device-side count of the listed pairs)
    long long ticks[24];
    int n_tick = 0;
    C2_TICK()
    constexpr int MT = TM / 32;                       // 32-row MFMA tiles per wave
    constexpr int N_PIECES = TM / 4;                  // 16-byte copy pieces per thread and tile (256 floats per row)
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int h = lane >> 5, i = lane & 31;
    const int colw = wave * 64 + 2 * i;               // first of this lane's two physical output columns (wide steps)

    C2BSet bx, by;
    const bool first_wide = p.step[0].N > 32;
    C2WideDesc dcur = c2_desc<LD>(p.step[0], wave, i, h, g);
    // the weight stream starts before the input tile is assembled
    if (first_wide) c2_load<LD>(bx, dcur, 0);
    else if (p.step[0].Bt != nullptr) c2_load_narrow(bx, p.step[0], wave, i, h, g);
    else c2_load_narrow_k(bx, p.step[0], wave, i, h, g);

    // ---- input tile -> sAct[m][k], zero-padded to the columns the first step multiplies -------------------------------
    {
        const bool cat_mode = p.in_mode == 0 || p.in_mode == 3;
        const int K0 = cat_mode ? (p.D + p.R) : p.K0;
        const int K0pad = first_wide ? min(CH_MAXW, (K0 + 63) & ~63) : CH_MAXW;
        const int m = tid & (TM - 1);
        constexpr int TPR = CH_THREADS / TM;           // threads per row
        const int q = tid / TM;
        const int row = row0 + m;
        const bool row_ok = row < n_rows;
        int b = row, w = row;
        if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; w = row - b * p.W; }
            else if (p.row_order == 1) { w = row / p.B; b = row - w * p.B; }
        } else if (p.in_mode == 3) {                   // row r of the compact list is the pair pairs[r] = b * W + j
            const int flat = min(max(row_ok ? p.pairs[row] : 0, 0), p.B * p.W - 1);
            b = flat / p.W; w = flat - b * p.W;
        }
        const float* src_a = cat_mode ? p.obs + (size_t)b * p.D
                                      : p.src + (p.nb > 1 ? (g / p.src_div) * p.sSrc : 0) + (size_t)row * p.ldsrc;
        const float* src_w = p.weights + (size_t)w * p.R;
        for (int kb = q * 16; kb < K0pad; kb += TPR * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = kb + u;
                float x = 0.f;
                if (row_ok && k < K0) {
                    if (cat_mode) x = (k < p.D) ? src_a[k] : src_w[k - p.D];
                    else x = src_a[k];
                }
                v[u] = x;
            }
#pragma unroll
            for (int u = 0; u < 16; u += 4)
                *reinterpret_cast<float4*>(sAct + m * C2_LDK + kb + u) = make_float4(v[u], v[u + 1], v[u + 2], v[u + 3]);
            if (p.x0_out != nullptr && row_ok) {
#pragma unroll
                for (int u = 0; u < 16; u += 4)
                    if (kb + u < p.ldx0)      // ldx0 is a multiple of 4
                        *reinterpret_cast<float4*>(p.x0_out + (size_t)row * p.ldx0 + kb + u) =
                            make_float4(v[u], v[u + 1], v[u + 2], v[u + 3]);
            }
        }
    }
    __syncthreads();
    C2_TICK()

    bool do_copy = false;                // the tile now in sAct must also be written to HBM (deferred store) ...
    C2CopyDst cdst = c2_copy_dst(sAct, 0, 0, 0, 0, tid);   // ... to here

    for (int s = 0; s < p.n_steps; ++s) {
        const ChainStep& st = p.step[s];
        const int K = st.K, N = st.N;
        const bool feed_next = (s + 1 < p.n_steps);
        const ChainStep& nxt = p.step[feed_next ? s + 1 : s];    // what the stream fetches after this step (or a dummy)
        const bool nxt_wide = nxt.N > 32;

        if (N > 32) {
            // ======================= matrix-core path =======================================================
            f32x16 acc[MT][2];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

            const int n_pairs = (K + 63) >> 6;         // K is treated as padded to a multiple of 64 with zero rows
            c2_load<LD>(by, dcur, CH_BK);
            const float* pa = sAct + i * C2_LDK + 4 * h;
            const C2WideDesc dnext = c2_desc<LD>(nxt, wave, i, h, g);
            if (do_copy) c2_wide_loop<TM, true, SCHED, LD>(acc, bx, by, sAct, pa, dcur, dnext, nxt_wide, n_pairs, cdst);
            else c2_wide_loop<TM, false, SCHED, LD>(acc, bx, by, sAct, pa, dcur, dnext, nxt_wide, n_pairs, cdst);
            dcur = dnext;
            C2_TICK()
            __syncthreads();     // every wave is past its last read of sAct (MFMA operands and copy pieces)
            C2_TICK()

            // ---- epilogue ---------------------------------------------------------------------------------
            const bool col_ok = colw < N;
            const bool col1_ok = colw + 1 < N;
            float bias0 = 0.f, bias1 = 0.f;
            if (st.bias != nullptr) {
                if (col_ok) bias0 = st.bias[g * st.sW + colw];
                if (col1_ok) bias1 = st.bias[g * st.sW + colw + 1];
            }
            unsigned long long bits_w = 0ull, bits_r = 0ull;
            const size_t bits_idx = (size_t)g * st.sBits + (size_t)(row0 >> 6) * CH_THREADS + tid;
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
                }
                // the tile goes to LDS whenever somebody reads it from there: the next step, or the deferred HBM copy
                if (feed_next || st.out != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        *reinterpret_cast<float2*>(sAct + m * C2_LDK + colw) = make_float2(acc[tm][0][r], acc[tm][1][r]);
                    }
                }
            }
            if (st.bits_out != nullptr) {
                if (TM == 64) st.bits_out[bits_idx] = bits_w;
                else   // a 32-row tile owns one 32-bit half of the word (rows 0-31 / 32-63 of the 64-row band)
                    reinterpret_cast<unsigned int*>(st.bits_out)[2 * bits_idx + ((row0 >> 5) & 1)] = (unsigned int)bits_w;
            }
            do_copy = st.out != nullptr;
            if (do_copy) cdst = c2_copy_dst(st.out + g * st.sOut, st.ldout, N, n_rows, row0, tid);
        } else {
            // ======================= narrow step (Q head): split-K over the four waves ========================
            // bx holds Bt[i][64w + 8c + 4h + 0..3] (c = 0..7): wave w contracts k in [64w, 64w + 64) for output column i
            f32x16 hacc[MT];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[a][r] = 0.f;
            const float* pa = sAct + i * C2_LDK + wave * 64 + 4 * h;
            // the saved input of this step goes to HBM first (its LDS image becomes the reduction scratch below)
            if (s > 0) {   // (step 0's operands were fetched by the prologue)
                if (st.Bt != nullptr) c2_load_narrow(bx, st, wave, i, h, g);
                else c2_load_narrow_k(bx, st, wave, i, h, g);
            }
            if (do_copy)
                for (int piece = 0; piece < N_PIECES; ++piece) c2_copy_piece(sAct, cdst, piece);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float4 a4[MT];
#pragma unroll
                for (int tm = 0; tm < MT; ++tm) a4[tm] = *reinterpret_cast<const float4*>(pa + tm * 32 * C2_LDK + 8 * c);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float bv = (t == 0) ? bx.v[2 * c].x : (t == 1) ? bx.v[2 * c].y : (t == 2) ? bx.v[2 * c + 1].x : bx.v[2 * c + 1].y;
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm) {   // columns >= K of sAct: finite stale values times zero weights
                        const float av = (t == 0) ? a4[tm].x : (t == 1) ? a4[tm].y : (t == 2) ? a4[tm].z : a4[tm].w;
                        hacc[tm] = mfma32(av, bv, hacc[tm]);
                    }
                }
            }
            __syncthreads();     // every wave is past its last read of sAct -> reuse it as the reduction scratch
            float* scr = sAct;
#pragma unroll
            for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((wave * MT + tm) * 16 + r) * 64 + lane] = hacc[tm][r];
            // the stream moves on while the partial tiles are reduced
            const C2WideDesc dnext = c2_desc<LD>(nxt, wave, i, h, g);
            if (nxt_wide) c2_load<LD>(bx, dnext, 0);
            dcur = dnext;
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
            const float bias = (st.bias != nullptr && n < N) ? st.bias[g * st.sW + n] : 0.f;
#pragma unroll
            for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = wave * 4 + q;
                    const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int row = row0 + m;
                    const bool ok = n < N && row < n_rows;
                    float v = red[tm][q] + bias;
                    if (st.relu) v = fmaxf(v, 0.f);
                    if (st.mask != nullptr) v = (ok && st.mask[(size_t)row * st.ldmask + n] > 0.f) ? v : 0.f;
                    if (!ok) v = 0.f;
                    if (feed_next) sAct[m * C2_LDK + n] = v;
                    if (st.out != nullptr && ok) st.out[g * st.sOut + (size_t)row * st.ldout + n] = v;
                }
            if (feed_next) {
                // the next step reads K' = N <= 32 padded to 64 columns: columns [32, 64) must be zero too
                for (int e = tid; e < 32 * TM; e += CH_THREADS) sAct[(e >> 5) * C2_LDK + 32 + (e & 31)] = 0.f;
            }
            do_copy = false;
        }
        C2_TICK()
        __syncthreads();    // sAct of the next step (or of the trailing copy) complete before anyone reads it
    }
    // a wide last step whose output goes to HBM (the backward chain's g_0): nothing is left to hide the copy behind
    if (do_copy)
        for (int piece = 0; piece < N_PIECES; ++piece) c2_copy_piece(sAct, cdst, piece);
    C2_TICK()
    if (PROF && prof_out != nullptr && tid == 0)
        for (int q = 0; q < 24; ++q) prof_out[q] = (q < n_tick) ? ticks[q] : 0;
}
#undef C2_TICK

// ---- persistent launch over up to CH_MAX_MULTI chains ------------------------------------------------------------------
// Units are 64-row tiles, numbered chain after chain.  Workgroup b (of S = gridDim.x) takes unit r*S + b in each of
// `full_rounds` rounds; the `tail_units` remaining units are done as 32-row halves when 2 * tail_units <= S (workgroup b
// takes half b & 1 of unit tail_base + b / 2), else as whole units by the first tail_units workgroups.  The second half of
// the grid runs its tail job first.
struct Chain2Multi {
    ChainArgs p[CH_MAX_MULTI];
    int unit_start[CH_MAX_MULTI + 1];   // first 64-row unit of each chain
    int n;
    int full_rounds;
    int tail_base, tail_units;
    int tail_halves;                    // 1: 32-row half tiles
    int stagger;                        // which workgroups run their tail job first: 0 none, 1 the second half of the grid,
                                        // 2 odd XCD-local index, 3 every second arrival on its CU (ticket from cu_tickets)
    unsigned int* cu_tickets;           // [C2_CU_SLOTS] monotonically increasing arrival counters (stagger 3)
    int xcd_contig;                     // 1 (grid % 8 == 0): consecutive LOGICAL workgroups run on one XCD.  Batched chains
                                        // (nb networks of a population): the two or four tiles of one network then share an
                                        // L2 and its weights are fetched from HBM once instead of once per tile
    long long* prof;                    // probes only: [gridDim.x][2][24] phase stamps
};
constexpr int C2_CU_SLOTS = 8192;

__device__ __forceinline__ int c2_find_chain(const Chain2Multi& m, int unit) {
    int q = 0;
    while (q + 1 < m.n && unit >= m.unit_start[q + 1]) ++q;
    return q;
}

// identity of the CU this workgroup runs on (XCC, shader engine / array, CU): only used to pair up co-resident workgroups
__device__ __forceinline__ unsigned c2_cu_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
    return ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
}

template <int SCHED, bool PROF, bool NMAJOR = false>
__device__ __forceinline__ void mlp_chain2_persistent(const Chain2Multi& m, float* sAct) {
    const int S = (int)gridDim.x;
    // (workgroup x runs on XCD x % 8)
    const int b = (m.xcd_contig && (S & 7) == 0) ? ((int)blockIdx.x & 7) * (S >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    bool tail_first = false;
    if (m.stagger == 1) tail_first = b >= (S >> 1);
    else if (m.stagger == 2) tail_first = ((b >> 3) & 1) != 0;
    else if (m.stagger == 3) {
        // the two workgroups of a CU take opposite job orders whatever the dispatcher's placement: every second arrival on a
        // CU (a ticket from a counter that only ever increments -- no reset between launches) runs its tail job first
        __shared__ unsigned s_ticket;
        if (threadIdx.x == 0) s_ticket = atomicAdd(m.cu_tickets + (c2_cu_key() & (C2_CU_SLOTS - 1)), 1u);
        __syncthreads();
        tail_first = (s_ticket & 1u) != 0;
    }
    // (in_mode 3: workgroup 0 reports the device-side row count to the host, whatever jobs it goes on to take -- ChainArgs::count_mirror)
    if (b == 0 && threadIdx.x == 0)
        for (int q = 0; q < m.n; ++q)
            if (m.p[q].rows_dev != nullptr && m.p[q].count_mirror != nullptr)
                *m.p[q].count_mirror = ((unsigned long long)m.p[q].count_tag << 32) | (unsigned int)*m.p[q].rows_dev;
    // jobs of this workgroup: `full_rounds` whole units and at most one tail job (every condition below is workgroup-uniform)
    for (int j = 0; j <= m.full_rounds; ++j) {
        const bool is_tail = tail_first ? (j == 0) : (j == m.full_rounds);
        int unit, half = -1;
        if (is_tail) {
            if (m.tail_halves) {
                if (b >= 2 * m.tail_units) continue;
                unit = m.tail_base + (b >> 1);
                half = b & 1;
            } else {
                if (b >= m.tail_units) continue;
                unit = m.tail_base + b;
            }
        } else {
            unit = (tail_first ? j - 1 : j) * S + b;
        }
        const int q = c2_find_chain(m, unit);
        int lu = unit - m.unit_start[q], g = 0;                  // batched chain: unit -> (network, row tile)
        if (m.p[q].nb > 1) { const int upn = (m.p[q].rows + 63) >> 6; g = lu / upn; lu -= g * upn; }
        const int row0 = lu * 64 + (half > 0 ? 32 : 0);
        // in_mode 3 (the lazily evaluated target rows of a step that selected MANY pairs): the row count is on the device; units
        // beyond it have nothing to do (workgroup-uniform), workgroup 0 reports the count to the host
        int nrd = -1;
        if (m.p[q].rows_dev != nullptr) {
            nrd = min(m.p[q].rows, *m.p[q].rows_dev);
            if (row0 >= nrd) continue;
        }
        const int rows_q = nrd >= 0 ? nrd : m.p[q].rows;
        long long* pout = (PROF && m.prof != nullptr) ? m.prof + ((size_t)b * 2 + (j > 0 ? 1 : 0)) * 24 : nullptr;
        if (NMAJOR) {        // (every chain of the launch streams its wide steps N-major: ChainArgs::fast == 2)
            if (half >= 0) {
                if (row0 < rows_q) mlp_chain2_body<32, SCHED, 2, PROF>(m.p[q], row0, sAct, pout, g, nrd);
            } else {
                mlp_chain2_body<64, SCHED, 2, PROF>(m.p[q], row0, sAct, pout, g, nrd);
            }
        } else if (m.p[q].fast) {
            if (half >= 0) {
                if (row0 < rows_q) mlp_chain2_body<32, SCHED, 1, PROF>(m.p[q], row0, sAct, pout, g, nrd);
            } else {
                mlp_chain2_body<64, SCHED, 1, PROF>(m.p[q], row0, sAct, pout, g, nrd);
            }
        } else {
            if (half >= 0) {
                if (row0 < rows_q) mlp_chain2_body<32, SCHED, 0, PROF>(m.p[q], row0, sAct, pout, g, nrd);
            } else {
                mlp_chain2_body<64, SCHED, 0, PROF>(m.p[q], row0, sAct, pout, g, nrd);
            }
        }
        __syncthreads();     // the tile buffer is free for the next job
    }
}

template <int SCHED>
__global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain2_kernel(Chain2Multi m) {
    // (+8: the last group's look-ahead operand read of the last row runs 4 floats past the tile; the values are unused)
    __shared__ __attribute__((aligned(16))) float sAct[C2_TM * C2_LDK + 8];
    mlp_chain2_persistent<SCHED, false>(m, sAct);
}

// the same schedule with the N-major weight stream (forward passes that read the nn.Linear matrices as they are: morl_ac.hip)
static __global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain2_n_kernel(Chain2Multi m) {
    __shared__ __attribute__((aligned(16))) float sAct[C2_TM * C2_LDK + 8];
    mlp_chain2_persistent<1, false, true>(m, sAct);
}

// ---- K-major shadow copies of the weights ----------------------------------------------------------------------------------
// job mode 0: W_l [N][K] (nn.Linear layout) -> Wt_l [Kpad][ldn], Kpad = round_up(K, 64), zero outside [K][N]: the operand the
//             forward chain streams (and, read N-major, the narrow-step operand of the backward chain)
// job mode 1: W_l [K = out][ld = in] -> the same rows followed by zero rows up to Kpad: the backward chain's operand when
//             `out` is not a multiple of 64 (the Q head), so that its weight stream never relies on the range check
// 32 x 32 tiles through LDS: global reads and writes are both coalesced (the round-1 kernel read with stride K: 13 MB
// fetched to move 1.7 MB).  blockIdx.y = 0: params -> wt; 1: params2 -> wt2.
constexpr int SH_T = 32;     // tile edge: 32 x 32 tiles -> ~480 workgroups for both flagship networks, one load + one store round each

struct ShadowJob {
    long long src_off, dst_off;
    int rows_src, cols_src;     // source matrix [rows_src][cols_src] (row-major, dense)
    int dst_rows, dst_ld;       // destination [dst_rows][dst_ld]
    int mode;
    int k4;                     // destination in the K4 layout of c2_load_fast (dst_ld == 256): (k, n) -> ((k >> 2) * 256 + n) * 4 + (k & 3)
    int tiles_c;                // tiles along the destination's column axis
    int tile_start;
};
struct ShadowArgs {
    ShadowJob job[2 * MORL_MAX_LAYERS];
    int n, tiles;
};

// tiles [first_tile, ...) of network `net` (0: params -> wt, 1: params2 -> wt2), strided by `tile_stride`
__device__ __forceinline__ void shadow_tiles_body(const float* __restrict__ params, float* __restrict__ wt, const ShadowArgs& a,
                                                  int first_tile, int tile_stride, float* tile) {
    const int tid = (int)threadIdx.x, fast = tid & (SH_T - 1), slow = tid / SH_T;     // slow in [0, 8)
    for (int t = first_tile; t < a.tiles; t += tile_stride) {
        int q = 0;
        while (q + 1 < a.n && t >= a.job[q + 1].tile_start) ++q;
        const ShadowJob& j = a.job[q];
        const int lt = t - j.tile_start;
        const int r0 = (lt / j.tiles_c) * SH_T, c0 = (lt % j.tiles_c) * SH_T;   // destination tile origin
        const float* src = params + j.src_off;
        float* dst = wt + j.dst_off;
        if (j.mode == 0) {
            // destination (k, n) <- source (n, k): read source rows n0 + nn along k, write destination rows k0 + kk along n
            float v[SH_T / 8];
#pragma unroll
            for (int it = 0; it < SH_T / 8; ++it) {
                const int n = c0 + slow + 8 * it, k = r0 + fast;
                v[it] = (n < j.rows_src && k < j.cols_src) ? src[(size_t)n * j.cols_src + k] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < SH_T / 8; ++it) tile[(slow + 8 * it) * (SH_T + 1) + fast] = v[it];
            __syncthreads();
            if (j.k4) {
                // K4 layout: thread (k-quad = slow, n = fast) writes the four k of its quad as one 16-byte piece; a half-wave
                // covers 512 contiguous bytes (dst_rows is a multiple of 64, the destination 16-byte aligned)
                const int n = c0 + fast, k = r0 + 4 * slow;
                if (n < j.dst_ld) {
                    const float* tq = tile + fast * (SH_T + 1) + 4 * slow;
                    *reinterpret_cast<float4*>(dst + (((size_t)(k >> 2) * j.dst_ld + n) << 2)) = make_float4(tq[0], tq[1], tq[2], tq[3]);
                }
            } else {
#pragma unroll
                for (int it = 0; it < SH_T / 8; ++it) {
                    const int kk = slow + 8 * it, k = r0 + kk, n = c0 + fast;
                    if (k < j.dst_rows && n < j.dst_ld) dst[(size_t)k * j.dst_ld + n] = tile[fast * (SH_T + 1) + kk];
                }
            }
            __syncthreads();
        } else if (j.k4) {
            // rows r0 + 4 * slow .. + 3 of column c0 + fast: four coalesced row reads, one 16-byte piece of the K4 copy
            const int r = r0 + 4 * slow, c = c0 + fast;
            if (c < j.dst_ld) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (r + u < j.rows_src && c < j.cols_src) ? src[(size_t)(r + u) * j.cols_src + c] : 0.f;
                *reinterpret_cast<float4*>(dst + (((size_t)(r >> 2) * j.dst_ld + c) << 2)) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int it = 0; it < SH_T / 8; ++it) {
                const int r = r0 + slow + 8 * it, c = c0 + fast;
                if (r < j.dst_rows && c < j.dst_ld)
                    dst[(size_t)r * j.dst_ld + c] = (r < j.rows_src && c < j.cols_src) ? src[(size_t)r * j.cols_src + c] : 0.f;
            }
        }
    }
}

static __global__ __launch_bounds__(256) void shadow_weights_kernel(const float* __restrict__ params, float* __restrict__ wt,
                                                                    const float* __restrict__ params2, float* __restrict__ wt2,
                                                                    ShadowArgs a) {
    __shared__ float tile[SH_T * (SH_T + 1)];
    if (blockIdx.y == 1) shadow_tiles_body(params2, wt2, a, (int)blockIdx.x, (int)gridDim.x, tile);
    else shadow_tiles_body(params, wt, a, (int)blockIdx.x, (int)gridDim.x, tile);
}

}  // namespace morl
