// Layer-fused MLP engine, SMALL-ROW variant (gfx950, wave64): 16-row tiles on v_mfma_f32_16x16x4_f32.
//
// Same contract and argument structures as mlp_chain2.h (ChainArgs / ChainStep of mlp_chain.h, activations M-major in LDS,
// weights streamed L2 -> registers, deferred LDS -> HBM saves), for launches whose row count cannot fill the chip with
// 32 / 64-row tiles: the 128-row batches of the actor-critic learners (one 32-row tile carries a whole 256 x 256 layer on ONE
// CU: 7 us of MFMA per layer, 4-8 busy CUs per pass) and the 2 048-row shards of an 8-GPU strong-scaled Envelope step.
// A 16-row tile halves the per-workgroup MFMA time again and doubles the workgroups; a tile's LDS footprint is 16.6 KB and its
// waves need ~110 registers, so four tiles share a CU.
//
//   MFMA 16x16x4: lane l supplies A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15]; D register r of lane l is
//   row (l >> 4) * 4 + r, column l & 15.  Exact fp32, k-ordered fma chain like the 32x32x2 form.
//   * A operand: lane (row, kq) reads sAct[row][16c + 4kq .. +3] as ONE ds_read_b128 per 16-deep group c; MFMA step t of the
//     group multiplies the contraction indices {16c + 4kq + t : kq = 0..3}.
//   * B operand: the MFMA column slot j of a wave's four column tiles is mapped to the physical columns 64w + 4j + ct, so one
//     buffer_load_dwordx4 per lane and step feeds all four tiles, and the epilogue writes 16-byte column quads.
//   * narrow steps (N <= 32): the four waves split the contraction (wave w: k in [64w, 64w + 64)), operand read N-major, two
//     column tiles of 16, partial tiles summed through LDS in wave order.
// The ReLU sign bits of a tile are one 16-bit word per work-item (bit ct * 4 + r) in the same buffers mlp_chain2 uses (same
// bytes per row); a forward / backward pair must use the same tile size -- the host decides from the row count alone.
#pragma once
#include <cstdlib>

#include "mlp_chain2.h"

namespace morl {

constexpr int C16_TM = 16;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct C16BSet {
    float4 v[8];       // one 32-deep chunk: v[4g + t] = B[k0 + 16g + 4kq + t][64w + 4j .. +3]  (narrow: see c16_load_narrow)
};

struct C16Desc {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane_off;      // bytes: (4kq * ldb + 64w + 4j) * 4, or CH_OOB
    int stride;        // bytes per k-row
};

// live == false: a zero-sized descriptor -- loads through it return zeros without touching memory (the stream's look-ahead past the
// last wide step of a run, issued unconditionally: a load under a branch costs the MFMA loop its look-ahead, see mlp_chain16_body)
__device__ __forceinline__ C16Desc c16_desc(const ChainStep& st, int wave, int j, int kq, int g, bool live = true) {
    C16Desc d;
    const int col = wave * 64 + 4 * j;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(st.Bmat + g * st.sW), 0, live ? (st.kpad > st.K ? st.kpad : st.K) * st.ldb * 4 : 0, 0x00020000);
    d.lane_off = (col < st.ldb) ? (4 * kq * st.ldb + col) * 4 : CH_OOB;
    d.stride = st.ldb * 4;
    return d;
}

// rows >= K lie beyond the resource -> 0 (K padding, dummy prefetches)
__device__ __forceinline__ void c16_load_wide(C16BSet& s, const C16Desc& d, int k0) {
    const int base = d.lane_off + k0 * d.stride;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int off = base + (16 * (q >> 2) + (q & 3)) * d.stride;
        s.v[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, off, 0, 0));
    }
}

// ChainArgs::fast == 1 (the contexts whose wide steps all have 256 columns): the operand is in the K4 layout of mlp_chain2.h's
// c2_load_fast -- element (k, n) at ((k >> 2) * 256 + n) * 4 + (k & 3).  The four k of a lane's group are 16 contiguous bytes
// per column: four loads per group as before, one per COLUMN instead of one per k, the register set filled transposed.
__device__ __forceinline__ C16Desc c16_desc_k4(const ChainStep& st, int wave, int j, int kq, int g, bool live = true) {
    C16Desc d;
    const int col = wave * 64 + 4 * j;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(st.Bmat + g * st.sW), 0, live ? (st.kpad > st.K ? st.kpad : st.K) * 256 * 4 : 0, 0x00020000);
    d.lane_off = (kq * 256 + col) * 16;
    d.stride = 256 * 16;              // bytes per k4-row
    return d;
}

__device__ __forceinline__ void c16_load_k4(C16BSet& s, const C16Desc& d, int k0) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
        float4 q[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            q[ct] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, d.lane_off + ct * 16,
                                                                                      ((k0 >> 2) + 4 * gq) * d.stride, 0));
        s.v[4 * gq + 0] = make_float4(q[0].x, q[1].x, q[2].x, q[3].x);
        s.v[4 * gq + 1] = make_float4(q[0].y, q[1].y, q[2].y, q[3].y);
        s.v[4 * gq + 2] = make_float4(q[0].z, q[1].z, q[2].z, q[3].z);
        s.v[4 * gq + 3] = make_float4(q[0].w, q[1].w, q[2].w, q[3].w);
    }
}

template <bool K4>
__device__ __forceinline__ C16Desc c16_desc_t(const ChainStep& st, int wave, int j, int kq, int g, bool live = true) {
    return K4 ? c16_desc_k4(st, wave, j, kq, g, live) : c16_desc(st, wave, j, kq, g, live);
}
template <bool K4>
__device__ __forceinline__ void c16_load_t(C16BSet& s, const C16Desc& d, int k0) {
    if (K4) c16_load_k4(s, d, k0); else c16_load_wide(s, d, k0);
}
// one 16-deep group (half a set: v[4 gq .. 4 gq + 3]) of the chunk at k0
template <bool K4>
__device__ __forceinline__ void c16_load_half(C16BSet& s, const C16Desc& d, int k0, int gq) {
    if (K4) {
        float4 q[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            q[ct] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, d.lane_off + ct * 16,
                                                                                      ((k0 >> 2) + 4 * gq) * d.stride, 0));
        s.v[4 * gq + 0] = make_float4(q[0].x, q[1].x, q[2].x, q[3].x);
        s.v[4 * gq + 1] = make_float4(q[0].y, q[1].y, q[2].y, q[3].y);
        s.v[4 * gq + 2] = make_float4(q[0].z, q[1].z, q[2].z, q[3].z);
        s.v[4 * gq + 3] = make_float4(q[0].w, q[1].w, q[2].w, q[3].w);
    } else {
        const int base = d.lane_off + k0 * d.stride;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            s.v[4 * gq + t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d.rsrc, base + (16 * gq + t) * d.stride, 0, 0));
    }
}

// narrow step: wave w contracts k in [64w, 64w + 64) = 4 groups of 16; lane (j = column within the tile, kq) loads
// Bt[n][64w + 16c + 4kq .. +3] for the two column tiles n = j and n = 16 + j: v[2c + ct]
// (live == false: a zero-sized descriptor, eight loads that touch no memory -- see c16_desc)
__device__ __forceinline__ void c16_load_narrow(C16BSet& s, const ChainStep& st, int wave, int j, int kq, int g, bool live = true) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(st.Bt + g * st.sW), 0, live ? st.N * st.ldbt * 4 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = wave * 64 + 16 * c + 4 * kq;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int n = 16 * ct + j;
            const int off = (n < st.N && k < st.K) ? (n * st.ldbt + k) * 4 : CH_OOB;     // K is a multiple of 4
            s.v[2 * c + ct] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        }
    }
}

__device__ __forceinline__ float c16_elem(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

// Input rows computed by the chain's own input stage (ChainArgs::in_mode == 4) instead of read from a matrix an extra launch wrote:
// hook(g, row, v) fills the row's first 16 entries (and keeps whatever copy other kernels need).  The default has none.
struct C16NoHook {
    static constexpr bool active = false;
    __device__ __forceinline__ void operator()(int, int, float (&)[16]) const {}
};

// ---- the post-op stages of LayerNorm / Dropout networks (ChainPost of mlp_chain.h) on the tile's rows in LDS ------------------
// ac_kernels.h's ac_uniform (same bits: the keep masks of the per-layer path), repeated here because this header is also compiled
// into the Envelope translation unit, which does not see ac_kernels.h
__device__ __forceinline__ float c16_uniform(unsigned long long seed, unsigned long long idx) {
    unsigned int x = (unsigned int)idx * 0x9E3779B9u + dropout_seed_mix(seed);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

constexpr int C16_POSTJ = CH_MAXW / 64;      // columns per lane: lane + 64 j

// A wave's four rows are processed SIDE BY SIDE (every load, hash and shuffle of the four rows issued before the first is consumed):
// the per-row work is a chain of latencies (LDS / L2 round trips, six dependent cross-lane shuffles per sum), and a tile has four
// waves, not the hundreds a launch of its own hides them behind -- row after row the stage cost 10+ us per layer on MI355X, the
// whole pass more than the launches it replaced.  Each row's arithmetic, and its order, is ac_post_fwd_body's / ac_post_bwd_kernel's.
__device__ __forceinline__ void c16_wave_sum4(float (&v)[4]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = __shfl_xor(v[q], off);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += o[q];
    }
}

// forward: wave w, rows 4w .. 4w+3 -- ac_post_fwd_body's arithmetic on sAct[m][0 .. N)
__device__ __forceinline__ void c16_post_fwd(const ChainPostSet& ps, const ChainPost& a, float* sAct, int row0, int n_rows, int N, int g) {
    const int lane = lane_id(), wave = wave_id();
    const float* __restrict__ gam = a.gamma ? a.gamma + (long long)g * ps.pstride : nullptr;
    float gm[C16_POSTJ], bt[C16_POSTJ];
#pragma unroll
    for (int j = 0; j < C16_POSTJ; ++j) {
        const int c = lane + 64 * j;
        gm[j] = (gam && c < N) ? gam[c] : 0.f;
        bt[j] = (gam && c < N) ? gam[N + c] : 0.f;
    }
    float v[4][C16_POSTJ], sum[4];
    bool live[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = wave * 4 + q, row = row0 + m;
        live[q] = row < n_rows;                                        // (wave-uniform)
        const float* zr = sAct + m * C2_LDK;
        sum[q] = 0.f;
        // element index of (row, column 0) in the keep-mask / RNG stream: ((g * cap + row) * N + c)
        const unsigned long long e0 = ((unsigned long long)g * ps.cap + row) * (unsigned long long)N;
#pragma unroll
        for (int j = 0; j < C16_POSTJ; ++j) {
            const int c = lane + 64 * j;
            float x = 0.f;
            bool kept = false;
            if (live[q] && c < N) {
                x = zr[c];
                if (a.drop) {
                    bool keep;
                    if (a.ext_mask) keep = a.ext_mask[(long long)g * ps.ext_gstride + (long long)row * N + c] != 0;
                    else keep = c16_uniform(a.seed, e0 + c) >= ps.drop_p;
                    kept = keep;
                    x = keep ? x * ps.inv_keep : 0.f;
                }
                sum[q] += x;
            }
            if (a.drop && live[q] && 64 * j < N) {           // (wave-uniform)
                const unsigned long long bits = __ballot(kept);
                if (lane == 0) a.mask[((long long)g * ps.cap + row) * ((N + 63) >> 6) + j] = bits;
            }
            v[q][j] = x;
        }
    }
    if (!gam) {                                                        // Dropout only (its backward needs the keep mask and h, nothing else)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* zr = sAct + (wave * 4 + q) * C2_LDK;
#pragma unroll
            for (int j = 0; j < C16_POSTJ; ++j) {
                const int c = lane + 64 * j;
                if (live[q] && c < N) zr[c] = fmaxf(v[q][j], 0.f);
            }
        }
        return;
    }
    c16_wave_sum4(sum);
    float mean[4], sq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mean[q] = sum[q] / (float)N;
        sq[q] = 0.f;
#pragma unroll
        for (int j = 0; j < C16_POSTJ; ++j) {
            const int c = lane + 64 * j;
            if (c < N) { const float d = v[q][j] - mean[q]; sq[q] += d * d; }
        }
    }
    c16_wave_sum4(sq);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = wave * 4 + q, row = row0 + m;
        if (!live[q]) continue;
        const float var = sq[q] / (float)N;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (lane == 0) a.rstd[(long long)g * ps.cap + row] = rstd;
        float* zr = sAct + m * C2_LDK;
        float* __restrict__ xh_out = a.xhat + (long long)g * a.gstride + (long long)row * a.ld;
#pragma unroll
        for (int j = 0; j < C16_POSTJ; ++j) {
            const int c = lane + 64 * j;
            if (c < N) {
                const float xh = (v[q][j] - mean[q]) * rstd;
                xh_out[c] = xh;
                zr[c] = fmaxf(xh * gm[j] + bt[j], 0.f);
            }
        }
    }
}

// backward: ac_post_bwd_kernel's arithmetic on sAct[m][0 .. N) = dLoss/dh -> dLoss/dz; dLoss/dh also goes to a.dh_out
__device__ __forceinline__ void c16_post_bwd(const ChainPostSet& ps, const ChainPost& a, float* sAct, int row0, int n_rows, int N, int g) {
    const int lane = lane_id(), wave = wave_id();
    const float* __restrict__ gam = a.gamma ? a.gamma + (long long)g * ps.pstride : nullptr;
    float gm[C16_POSTJ];
#pragma unroll
    for (int j = 0; j < C16_POSTJ; ++j) {
        const int c = lane + 64 * j;
        gm[j] = (gam && c < N) ? gam[c] : 0.f;
    }
    float dxh[4][C16_POSTJ], xh[4][C16_POSTJ], s1[4], s2[4], rs[4];
    unsigned keepbits[4];
    bool live[4];
    // every global load of the four rows first (h, xhat, keep flags, rstd), then the arithmetic
    float hv[4][C16_POSTJ];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = row0 + wave * 4 + q;
        live[q] = row < n_rows;
        const long long base = (long long)g * a.gstride + (long long)(live[q] ? row : row0) * a.ld;
        keepbits[q] = 0u;
        rs[q] = (gam && live[q]) ? a.rstd[(long long)g * ps.cap + row] : 1.f;
#pragma unroll
        for (int j = 0; j < C16_POSTJ; ++j) {
            const int c = lane + 64 * j;
            const bool ok = live[q] && c < N;
            hv[q][j] = ok ? a.h[base + c] : 0.f;
            xh[q][j] = (ok && gam) ? a.xhat[base + c] : 0.f;
            if (ok && a.drop && ((a.mask[((long long)g * ps.cap + row) * ((N + 63) >> 6) + j] >> lane) & 1ull)) keepbits[q] |= 1u << j;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = wave * 4 + q, row = row0 + m;
        const float* dr = sAct + m * C2_LDK;
        const long long base = (long long)g * a.gstride + (long long)row * a.ld;
        s1[q] = 0.f; s2[q] = 0.f;
#pragma unroll
        for (int j = 0; j < C16_POSTJ; ++j) {
            const int c = lane + 64 * j;
            float t = 0.f;
            if (live[q] && c < N) {
                const float d = dr[c];
                if (a.dh_out) a.dh_out[base + c] = d;
                t = (hv[q][j] > 0.f) ? d : 0.f;                        // ReLU
                if (gam) {
                    t *= gm[j];
                    s1[q] += t;
                    s2[q] += t * xh[q][j];
                }
            }
            dxh[q][j] = t;
        }
    }
    if (gam) { c16_wave_sum4(s1); c16_wave_sum4(s2); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!live[q]) continue;
        float* dr = sAct + (wave * 4 + q) * C2_LDK;
        const float m1 = s1[q] / (float)N, m2 = s2[q] / (float)N;
#pragma unroll
        for (int j = 0; j < C16_POSTJ; ++j) {
            const int c = lane + 64 * j;
            if (c < N) {
                float t = dxh[q][j];
                if (gam) t = rs[q] * (t - m1 - xh[q][j] * m2);
                if (a.drop) t = ((keepbits[q] >> j) & 1u) ? t * ps.inv_keep : 0.f;
                dr[c] = t;
            }
        }
    }
}

// one 16-row tile (rows [row0, row0 + 16) of network g) through the whole chain; K4: the weight layout of ChainArgs::fast == 1
// POST: 0 plain; 1 / 2: the hidden steps of a LayerNorm / Dropout network carry their post-op (forward / backward: `ps`)
template <bool K4, class Hook = C16NoHook, int POST = 0>
__device__ __forceinline__ void mlp_chain16_body(const ChainArgs& p, int row0, float* sAct, int g, const Hook& hook = Hook(),
                                                const ChainPostSet* ps = nullptr) {
    constexpr int N_PIECES = 4;                       // 16 rows x 64 quads / 256 threads
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int n_rows = p.rows_dev ? min(p.rows, *p.rows_dev) : p.rows;        // (device-side count: in_mode 3)
    const int kq = lane >> 4, j = lane & 15;
    const int col0 = wave * 64 + 4 * j;               // first of this lane's four physical output columns (wide steps)

#ifdef C16_PROF
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tp_ = clock64();
    const long long tstart_ = tp_;
#define C16_T(q) { const long long t_ = clock64(); pt[q] += t_ - tp_; tp_ = t_; }
#else
#define C16_T(q)
#endif
    // bx / by: the two sets of the wide steps' weight stream; bn: a narrow step's operand -- a set of its own, because what is
    // loaded into a set behind one step and used in the next is, to the compiler, loaded in front of EVERY kind of next step:
    // with the narrow operand in bx each pair of a wide step waited for all loads in flight before its last four MFMAs
    C16BSet bx, by, bn;
    const bool first_wide = p.step[0].N > 32;
    bool narrow_ready = !first_wide;          // the narrow step's operand is already in bn
    C16Desc dcur = c16_desc_t<K4>(p.step[0], wave, j, kq, g);

    // ---- input tile -> sAct[m][k], zero-padded to the columns the first step multiplies --------------------------------
    {
        const bool cat_mode = p.in_mode == 0 || p.in_mode == 3;
        const int K0 = cat_mode ? (p.D + p.R) : p.K0;
        const int K0pad = first_wide ? min(CH_MAXW, (K0 + 63) & ~63) : CH_MAXW;
        const int m = tid & 15, q = tid >> 4;          // 16 threads per row, 16 columns each
        const int row = row0 + m;
        const bool row_ok = row < n_rows;
        int b = row, w = row;
        if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; w = row - b * p.W; }
            else if (p.row_order == 1) { w = row / p.B; b = row - w * p.B; }
        } else if (p.in_mode == 3) {
            const int flat = row_ok ? p.pairs[row] : 0;
            b = flat / p.W; w = flat - b * p.W;
        }
        const float* src_a = cat_mode ? p.obs + (size_t)b * p.D
                                              : p.src + (p.nb > 1 ? (g / p.src_div) * p.sSrc : 0) + (size_t)row * p.ldsrc;
        const float* src_w = p.weights + (size_t)w * p.R;
        const int kb = q * 16;
        bool hooked = false;
        float hv[16];
        if constexpr (Hook::active) {
            if (p.in_mode == 4) {                      // (K0 <= 16: the row's entries are the first column block's)
                hooked = true;
#pragma unroll
                for (int u = 0; u < 16; ++u) hv[u] = 0.f;
                if (q == 0 && row_ok) hook(g, row, hv);
            }
        }
        // (the input rows' loads stand in front of the weight stream's first sets: the wait for them then leaves the sets in flight)
        float v[16];
        const bool in_live = kb < K0pad;
        if (in_live) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = kb + u;
                float x = 0.f;
                if (row_ok && k < K0) {
                    if (cat_mode) x = (k < p.D) ? src_a[k] : src_w[k - p.D];
                    else if (hooked) x = (q == 0) ? hv[u] : 0.f;
                    else x = src_a[k];
                }
                v[u] = x;
            }
        }
        if (first_wide) { c16_load_t<K4>(bx, dcur, 0); c16_load_t<K4>(by, dcur, CH_BK); }
        else c16_load_narrow(bn, p.step[0], wave, j, kq, g);
        if (in_live) {
#pragma unroll
            for (int u = 0; u < 16; u += 4)
                *reinterpret_cast<float4*>(sAct + m * C2_LDK + kb + u) = make_float4(v[u], v[u + 1], v[u + 2], v[u + 3]);
            if (p.x0_out != nullptr && row_ok) {
#pragma unroll
                for (int u = 0; u < 16; u += 4)
                    if (kb + u < p.ldx0)
                        *reinterpret_cast<float4*>(p.x0_out + (size_t)row * p.ldx0 + kb + u) =
                            make_float4(v[u], v[u + 1], v[u + 2], v[u + 3]);
            }
        }
    }
    __syncthreads();
    C16_T(0)

    bool do_copy = false;
    C2CopyDst cdst = c2_copy_dst(sAct, 0, 0, 0, 0, tid);

    for (int s = 0; s < p.n_steps; ++s) {
        const ChainStep& st = p.step[s];
        const int K = st.K, N = st.N;
        const bool feed_next = (s + 1 < p.n_steps);
        const ChainStep& nxt = p.step[feed_next ? s + 1 : s];
        const bool nxt_wide = nxt.N > 32;

        if (N > 32) {
            // ======================= matrix-core path =======================================================
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ct][r] = 0.f;
            const int n_pairs = (K + 63) >> 6;          // K is treated as padded to a multiple of 64 with zero rows
            // (both sets hold this step's first two chunks: the prologue, the wide step or the narrow step in front loaded them)
            // what the epilogue needs from memory is requested before the contraction, not behind its barrier (phase stamps of a
            // -DC16_PROF build: 1 500 of a step's 3 300 epilogue cycles were these two loads' latency)
            // -- as range-checked buffer loads, NOT under `if (column exists)` / `if (pointer)`: a load under a branch in front of the
            // MFMA loop is, to the compiler's wait-count pass, a load that may or may not stand between the weight stream's sets and
            // their use, and the loop's waits came out as "all but two" (round 5)
            float bias[4];
            {
                const float* bp = st.bias != nullptr ? st.bias + g * st.sW : nullptr;
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, bp != nullptr ? N * 4 : 0, 0x00020000);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bias[ct] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (col0 + ct) * 4, 0, 0));
            }
            // 16-bit words: same bytes per row as the 64-bit words of the 32 / 64-row tilings
            const size_t bits_idx = ((size_t)g * st.sBits) * 4 + (size_t)(row0 >> 4) * CH_THREADS + tid;
            unsigned int bits_w = 0u, bits_r;
            {
                const unsigned short* wp = reinterpret_cast<const unsigned short*>(st.bits_in) + (st.bits_in != nullptr ? bits_idx - tid : 0);
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, st.bits_in != nullptr ? CH_THREADS * 2 : 0, 0x00020000);
                bits_r = (unsigned int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rw, tid * 2, 0, 0);
            }
            // a narrow step behind this one: its operand, requested HERE and not behind the loop -- its registers are free for the
            // loop's temporaries otherwise, and the next wide step's loop then opens by waiting for loads younger than its own sets
            narrow_ready = feed_next && !nxt_wide;
            c16_load_narrow(bn, nxt, wave, j, kq, g, narrow_ready);
            const float* pa = sAct + j * C2_LDK + 4 * kq;
            // the stream behind this step: the next step's operand if that step is wide, nothing otherwise (a zero-sized descriptor).
            // Inside the loop every reload is unconditional, from a descriptor picked with scalar selects, and nothing is fetched
            // from the argument block: with the narrow step's operand loaded under a branch in here (round 4) the compiler's
            // wait-count pass made the second set's MFMAs wait for the loads just issued into the first -- no look-ahead at all,
            // 12 650 cycles for a 256 x 256 step of 8 192 MFMA cycles (profiles/r04_chain16_phases.txt).
            const bool nxt_live = feed_next && nxt_wide;
            const C16Desc dnext = c16_desc_t<K4>(nxt, wave, j, kq, g, nxt_live);
            float4 an, ac = *reinterpret_cast<const float4*>(pa);
            int piece = 0;
            for (int pr = 0; pr < n_pairs; ++pr) {
                const int k0 = pr * 64;
                const bool more = pr + 1 < n_pairs;
                C16Desc dx;
                dx.rsrc = more ? dcur.rsrc : dnext.rsrc;
                dx.lane_off = more ? dcur.lane_off : dnext.lane_off;
                dx.stride = more ? dcur.stride : dnext.stride;
// one group = 16 contraction indices = 4 MFMA steps x 4 column tiles; the A quad of the NEXT group is read first
#define C16_GROUP(SET, G, KNEXT)                                                      \
    {                                                                                 \
        an = *reinterpret_cast<const float4*>(pa + (KNEXT));                          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                               \
            const float av = c16_elem(ac, t);                                         \
            const float4 bv = SET.v[4 * (G) + t];                                     \
            acc[0] = mfma16(av, bv.x, acc[0]);                                        \
            acc[1] = mfma16(av, bv.y, acc[1]);                                        \
            acc[2] = mfma16(av, bv.z, acc[2]);                                        \
            acc[3] = mfma16(av, bv.w, acc[3]);                                        \
        }                                                                             \
        ac = an;                                                                      \
        C2_SGB(0x100, 1);                                                             \
        C2_SGB(0x008, 16);                                                            \
    }
                // Each 16-deep half of a set is reloaded as soon as its group's MFMAs are issued -- three groups (1 500 MFMA cycles)
                // ahead of its next use, where whole sets reloaded behind their second group were two ahead -- and pinned where it
                // stands: left alone the scheduler sinks the loads behind the pair's last MFMAs and the next iteration opens by
                // waiting for all sixteen.
                const int kx = more ? k0 + 64 : 0, ky = more ? k0 + 96 : CH_BK;
                C16_GROUP(bx, 0, k0 + 16)
                __builtin_amdgcn_sched_barrier(0);
                c16_load_half<K4>(bx, dx, kx, 0);
                __builtin_amdgcn_sched_barrier(0);
                C16_GROUP(bx, 1, k0 + 32)
                __builtin_amdgcn_sched_barrier(0);
                c16_load_half<K4>(bx, dx, kx, 1);
                __builtin_amdgcn_sched_barrier(0);
                C16_GROUP(by, 0, k0 + 48)
                __builtin_amdgcn_sched_barrier(0);
                c16_load_half<K4>(by, dx, ky, 0);
                __builtin_amdgcn_sched_barrier(0);
                // (the last group's look-ahead read stays inside the buffer: column k0 + 64 + 15 <= 271 -> see the kernel's array)
                C16_GROUP(by, 1, k0 + 64)
                __builtin_amdgcn_sched_barrier(0);
                c16_load_half<K4>(by, dx, ky, 1);
                __builtin_amdgcn_sched_barrier(0);
#undef C16_GROUP
                if (do_copy && piece < N_PIECES) { c2_copy_piece(sAct, cdst, piece); ++piece; }
            }
            if (do_copy)
                for (; piece < N_PIECES; ++piece) c2_copy_piece(sAct, cdst, piece);
            dcur = dnext;
            C16_T(K > 64 ? 2 : 1)
            __syncthreads();     // every wave is past its last read of sAct
            C16_T(3)

            // ---- epilogue ---------------------------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    float x = acc[ct][r] + bias[ct];
                    if (st.relu) x = fmaxf(x, 0.f);
                    x = (col0 + ct < N) ? x : 0.f;
                    if (st.bits_out != nullptr && x > 0.f) bits_w |= 1u << (ct * 4 + r);
                    if (st.bits_in != nullptr) x = ((bits_r >> (ct * 4 + r)) & 1u) ? x : 0.f;
                    v[ct] = x;
                }
                if (feed_next || st.out != nullptr)
                    *reinterpret_cast<float4*>(sAct + (kq * 4 + r) * C2_LDK + col0) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (st.bits_out != nullptr) reinterpret_cast<unsigned short*>(st.bits_out)[bits_idx] = (unsigned short)bits_w;
            do_copy = st.out != nullptr;
            if (do_copy) cdst = c2_copy_dst(st.out + g * st.sOut, st.ldout, N, n_rows, row0, tid);
        } else {
            // ======================= narrow step: split-K over the four waves ==================================
            f32x4 hacc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) hacc[ct][r] = 0.f;
            if (!narrow_ready) c16_load_narrow(bn, st, wave, j, kq, g);       // (a narrow step behind a narrow step)
            narrow_ready = false;
            // (the bias this step's output needs: requested before the contraction and its two barriers)
            const int ct = wave >> 1;
            const int n = 16 * ct + j;
            const float bias = (st.bias != nullptr && n < N) ? st.bias[g * st.sW + n] : 0.f;
            if (do_copy)
                for (int piece = 0; piece < N_PIECES; ++piece) c2_copy_piece(sAct, cdst, piece);
            const float* pa = sAct + j * C2_LDK + wave * 64 + 4 * kq;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 a4 = *reinterpret_cast<const float4*>(pa + 16 * c);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float av = c16_elem(a4, t);        // columns >= K of sAct: finite stale values times zero weights
                    hacc[0] = mfma16(av, c16_elem(bn.v[2 * c], t), hacc[0]);
                    hacc[1] = mfma16(av, c16_elem(bn.v[2 * c + 1], t), hacc[1]);
                }
            }
            __syncthreads();     // every wave is past its last read of sAct -> reuse it as the reduction scratch
            float* scr = sAct;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[((wave * 2 + ct) * 4 + r) * 64 + lane] = hacc[ct][r];
            const C16Desc dnext = c16_desc_t<K4>(nxt, wave, j, kq, g, feed_next && nxt_wide);
            if (feed_next && nxt_wide) { c16_load_t<K4>(bx, dnext, 0); c16_load_t<K4>(by, dnext, CH_BK); }
            dcur = dnext;
            __syncthreads();
            // thread (wave, lane): column tile ct = wave >> 1, registers 2 * (wave & 1) + {0, 1} -- sums the four partials in wave order
            float red[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * (wave & 1) + q;
                float v = scr[((0 * 2 + ct) * 4 + r) * 64 + lane];
                v += scr[((1 * 2 + ct) * 4 + r) * 64 + lane];
                v += scr[((2 * 2 + ct) * 4 + r) * 64 + lane];
                v += scr[((3 * 2 + ct) * 4 + r) * 64 + lane];
                red[q] = v;
            }
            if (feed_next) __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * (wave & 1) + q;
                const int m = kq * 4 + r;
                const int row = row0 + m;
                const bool ok = n < N && row < n_rows;
                float v = red[q] + bias;
                if (st.relu) v = fmaxf(v, 0.f);
                if (st.mask != nullptr) v = (ok && st.mask[(size_t)row * st.ldmask + n] > 0.f) ? v : 0.f;
                if (!ok) v = 0.f;
                if (feed_next) sAct[m * C2_LDK + n] = v;
                if (st.out != nullptr && ok) st.out[g * st.sOut + (size_t)row * st.ldout + n] = v;
            }
            if (feed_next)      // the next step reads K' = N <= 32 padded to 64 columns: columns [32, 64) must be zero too
                for (int e = tid; e < 32 * C16_TM; e += CH_THREADS) sAct[(e >> 5) * C2_LDK + 32 + (e & 31)] = 0.f;
            do_copy = false;
            C16_T(5)
        }
        __syncthreads();
        if constexpr (POST != 0) {
            // the step's output rows are complete in LDS: their post-op, then the same barrier again (the next step's operand reads,
            // and the deferred copy to st.out, see what the post-op left)
            if (ps->st[s].active && N > 32) {
                if (POST == 1) c16_post_fwd(*ps, ps->st[s], sAct, row0, n_rows, N, g);
                else c16_post_bwd(*ps, ps->st[s], sAct, row0, n_rows, N, g);
                __syncthreads();
            }
        }
        C16_T(4)
    }
    if (do_copy)
        for (int piece = 0; piece < N_PIECES; ++piece) c2_copy_piece(sAct, cdst, piece);
    C16_T(6)
#ifdef C16_PROF
    if (p.prof != nullptr && lane == 0) {
        long long* o = p.prof + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 7; ++i) o[i] = pt[i];
        o[7] = tp_ - tstart_;
    }
#endif
#undef C16_T
}

// One workgroup per 16-row tile; tiles are numbered chain after chain, network after network.
struct Chain16Multi {
    ChainArgs p[CH_MAX_MULTI];
    int tile_start[CH_MAX_MULTI + 1];
    int n;
};

static __global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain16_kernel(Chain16Multi m) {
    // (+16: the last group's look-ahead operand read of the last row runs up to 12 floats past the tile; the values are unused)
    __shared__ __attribute__((aligned(16))) float sAct[C16_TM * C2_LDK + 16];
    const int b = (int)blockIdx.x;
    int q = 0;
    while (q + 1 < m.n && b >= m.tile_start[q + 1]) ++q;
    // (this chain's share of the argument block into the scalar cache in one round trip: the body reads it step by step, each step's
    // descriptors behind the previous step's barrier -- a chain of first-touch misses on a launch that is a chain of latencies anyway)
    kernarg_warm_at<sizeof(ChainArgs), sizeof(Chain16Multi)>(q * (int)sizeof(ChainArgs));
    int lt = b - m.tile_start[q], g = 0;
    const int tpn = (m.p[q].rows + C16_TM - 1) / C16_TM;          // tiles per network
    if (m.p[q].nb > 1) { g = lt / tpn; lt -= g * tpn; }
    if (b == 0 && threadIdx.x == 0 && m.p[q].rows_dev != nullptr && m.p[q].count_mirror != nullptr)      // (see ChainArgs::count_mirror)
        *m.p[q].count_mirror = ((unsigned long long)m.p[q].count_tag << 32) | (unsigned int)*m.p[q].rows_dev;
    if (m.p[q].rows_dev != nullptr && lt * C16_TM >= *m.p[q].rows_dev) return;      // (workgroup-uniform)
    if (m.p[q].fast == 1) mlp_chain16_body<true>(m.p[q], lt * C16_TM, sAct, g);
    else mlp_chain16_body<false>(m.p[q], lt * C16_TM, sAct, g);
}

// the same launch for LayerNorm / Dropout networks: chain q's hidden steps carry their post-op (one ChainPostSet per chain; at most
// two chains: paired forward passes)
struct Chain16PostMulti {
    ChainArgs p[2];
    ChainPostSet ps[2];
    int tile_start[3];
    int n;
};

template <int POST>
static __global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain16_post_kernel(Chain16PostMulti m) {
    __shared__ __attribute__((aligned(16))) float sAct[C16_TM * C2_LDK + 16];
    const int b = (int)blockIdx.x;
    const int q = (m.n > 1 && b >= m.tile_start[1]) ? 1 : 0;
    kernarg_warm_at<sizeof(ChainArgs), sizeof(Chain16PostMulti)>(q * (int)sizeof(ChainArgs));
    kernarg_warm_at<sizeof(ChainPostSet), sizeof(Chain16PostMulti)>(2 * (int)sizeof(ChainArgs) + q * (int)sizeof(ChainPostSet));
    int lt = b - m.tile_start[q], g = 0;
    const int tpn = (m.p[q].rows + C16_TM - 1) / C16_TM;          // tiles per network
    if (m.p[q].nb > 1) { g = lt / tpn; lt -= g * tpn; }
    if (m.p[q].fast == 1) mlp_chain16_body<true, C16NoHook, POST>(m.p[q], lt * C16_TM, sAct, g, C16NoHook(), &m.ps[q]);
    else mlp_chain16_body<false, C16NoHook, POST>(m.p[q], lt * C16_TM, sAct, g, C16NoHook(), &m.ps[q]);
}

// Host side: which launches take the 16-row tiles, and the tile table.  The choice must be the same for a forward pass and
// the backward pass that consumes its sign bits, so it depends only on the rows (x networks) of a chain -- equal for both.
// chains of up to this many rows take the 16-row tiles.  The Envelope step (five-layer nets): 256 x 24 = 6 144 rows 0.215 ms on
// 16-row tiles against 0.242 ms on 64 / 32-row ones, 8 192 rows 0.240 vs 0.245, 12 288 rows equal.  The actor-critic learners keep
// 4 096 (AC_C16_MAX_ROWS in morl_ac.hip): beyond it their populations run the N-major weight stream of the large-tile chain.
constexpr int C16_MAX_ROWS = 8192;

inline bool chain16_wanted(const ChainArgs* chains, int n, long long max_rows = C16_MAX_ROWS) {
    long long most = 0;
    for (int q = 0; q < n; ++q) {
        const long long r = (long long)chains[q].rows * (chains[q].nb > 1 ? chains[q].nb : 1);
        most = r > most ? r : most;
    }
    return most <= max_rows;
}

inline int chain16_fill(Chain16Multi& m, const ChainArgs* chains, int n) {
    m.n = n;
    int tiles = 0;
    for (int q = 0; q < n; ++q) {
        m.p[q] = chains[q];
        m.tile_start[q] = tiles;
        tiles += (chains[q].nb > 1 ? chains[q].nb : 1) * ((chains[q].rows + C16_TM - 1) / C16_TM);
    }
    for (int q = n; q <= CH_MAX_MULTI; ++q) m.tile_start[q] = tiles;
    return tiles;
}

}  // namespace morl
