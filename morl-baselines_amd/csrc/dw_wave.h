// Weight-gradient GEMMs of one backward pass as independent WAVE-level tiles (gfx950, wave64).
//
//   dW_l[o][i] = sum_rows g_l[row][o] * h_l[row][i]        db_l[o] = sum_rows g_l[row][o]
//
// Both operands are row-major with the contraction index (the batch row) as the slow axis, which is exactly the
// v_mfma_f32_32x32x2_f32 operand shape: lane (i, h) needs A[m][k+h] = g[k+h][m] and B[k+h][n] = h[k+h][n] -- 32
// consecutive floats of one row per half-wave.  So nothing has to be transposed or staged: every wave streams its
// operands HBM/L2 -> registers with buffer_load_dwordx2 (the MFMA row / column slot s is mapped to physical index
// 2s + {0,1}, so one 8-byte load feeds two tiles), owns a 64 x 64 tile of one layer's dW for one slice of the batch
// rows, and never synchronises with anybody: no LDS, no barriers.  One wave per SIMD by construction (the kernel is
// allowed the whole register file): three operand sets of 32 rows rotate, two are always in flight.
// The buffer resource of a wave spans exactly its row slice, so the hardware range check returns 0 for the rows of the
// (3-chunk padded) tail -- no branches, exact s_waitcnt counts.
// Split-K partial tiles go to slabs[split][P] in the flat parameter layout; grad_reduce_kernel sums them in order.
// Roofline: fp32 MFMA, 2*rows*out*in flop per layer; algorithmic bytes: rows*(out+in)*4 per 64x64 tile column/row
// re-read (L2 / MALL resident: g_l and h_l were just written by the backward / forward chain).
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

constexpr int DW_TILE = 64;
constexpr int DW_CHUNK = 32;   // batch rows per operand set

struct DwProblem {
    const float* G;   // [rows][ldg]  dLoss/dz_l
    const float* H;   // [rows][ldh]  layer input (x0 or saved activation)
    float* C;         // slab 0 of dW_l, row-major [M][ldc]
    float* bias;      // slab 0 of db_l [M]
    int M, N;         // out, in
    int ldg, ldh, ldc;
    int tiles_n;
};

struct DwArgs {
    DwProblem p[MORL_MAX_LAYERS];
    int tile_start[MORL_MAX_LAYERS + 1];
    int n;
    int rows;
    int k_per_split;           // multiple of 2
    long long slab_stride;     // floats between split slabs
};

struct DwSet {
    float2 a[16], b[16];
};

__device__ __forceinline__ void dw_load(DwSet& s, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int offa,
                                        int offb, int stride_a, int stride_b, int chunk) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row2 = chunk * DW_CHUNK + 2 * j;      // + h is folded into offa / offb
        s.a[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(ra, offa + row2 * stride_a, 0, 0));
        s.b[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rb, offb + row2 * stride_b, 0, 0));
    }
}

__global__ __launch_bounds__(64, 1) void dw_wave_kernel(DwArgs g) {
    const int lane = lane_id();
    const int h = lane >> 5, i = lane & 31;
    const int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int q = 0;
    while (q + 1 < g.n && id >= g.tile_start[q + 1]) ++q;
    const DwProblem& p = g.p[q];
    const int local = id - g.tile_start[q];
    const int m0 = (local / p.tiles_n) * DW_TILE, n0 = (local % p.tiles_n) * DW_TILE;
    const int split = (int)blockIdx.y;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.rows, kbeg + g.k_per_split);
    const int nrows = max(0, kend - kbeg);

    // operand streams: this wave's row slice only (range check = slice bounds)
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.G + (size_t)kbeg * p.ldg), 0, nrows * p.ldg * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.H + (size_t)kbeg * p.ldh), 0, nrows * p.ldh * 4, 0x00020000);
    const int ca = m0 + 2 * i, cb = n0 + 2 * i;                       // this lane's column pair in g / h
    // a pair straddling the row end would read into the next row: ld is even and >= the padded width, so
    // "first column < ld" is enough; columns in [M, ldg) / [N, ldh) are zero padding written by the producers
    const int offa = (ca < p.ldg) ? (h * p.ldg + ca) * 4 : 0x40000000;
    const int offb = (cb < p.ldh) ? (h * p.ldh + cb) * 4 : 0x40000000;
    const int stride_a = p.ldg * 4, stride_b = p.ldh * 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum0 = 0.f, bsum1 = 0.f;                                   // column sums of g (bias gradient)

    const int n_chunks = (nrows + DW_CHUNK - 1) / DW_CHUNK;
    const int n_iter = (n_chunks + 2) / 3;                            // chunks padded to a multiple of 3 (zeros)
    DwSet sx, sy, sz;
    dw_load(sx, ra, rb, offa, offb, stride_a, stride_b, 0);
    dw_load(sy, ra, rb, offa, offb, stride_a, stride_b, 1);
#define DW_COMPUTE(S)                                                        \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                         \
        acc[0][0] = mfma32(S.a[j].x, S.b[j].x, acc[0][0]);                   \
        acc[0][1] = mfma32(S.a[j].x, S.b[j].y, acc[0][1]);                   \
        acc[1][0] = mfma32(S.a[j].y, S.b[j].x, acc[1][0]);                   \
        acc[1][1] = mfma32(S.a[j].y, S.b[j].y, acc[1][1]);                   \
        bsum0 += S.a[j].x;                                                   \
        bsum1 += S.a[j].y;                                                   \
    }
    for (int it = 0; it < n_iter; ++it) {
        const int c = it * 3;
        dw_load(sz, ra, rb, offa, offb, stride_a, stride_b, c + 2);
        DW_COMPUTE(sx)
        dw_load(sx, ra, rb, offa, offb, stride_a, stride_b, c + 3);
        DW_COMPUTE(sy)
        dw_load(sy, ra, rb, offa, offb, stride_a, stride_b, c + 4);
        DW_COMPUTE(sz)
    }
#undef DW_COMPUTE

    // ---- epilogue: C[m][n], m = m0 + 2*slot_m + tm, n = n0 + 2*i + tn -------------------------------------------
    float* __restrict__ C = p.C + (size_t)split * g.slab_stride;
    const int nn = n0 + 2 * i;
    const bool pair_ok = (p.ldc & 1) == 0;                            // 8-byte stores need an even row stride
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + tm;
            if (m < p.M && nn < p.N) {
                float* o = C + (size_t)m * p.ldc + nn;
                if (nn + 1 < p.N) {
                    if (pair_ok) *reinterpret_cast<float2*>(o) = make_float2(acc[tm][0][r], acc[tm][1][r]);
                    else { o[0] = acc[tm][0][r]; o[1] = acc[tm][1][r]; }
                } else {
                    o[0] = acc[tm][0][r];
                }
            }
        }
    if (n0 == 0 && p.bias != nullptr) {
        // lanes (i, 0) and (i, 1) hold the even / odd batch rows of columns 2i, 2i+1: fixed-order combine
        const float o0 = __shfl_xor(bsum0, 32), o1 = __shfl_xor(bsum1, 32);
        if (h == 0) {
            float* bo = p.bias + (size_t)split * g.slab_stride;
            const int m = m0 + 2 * i;
            if (m < p.M) bo[m] = bsum0 + o0;
            if (m + 1 < p.M) bo[m + 1] = bsum1 + o1;
        }
    }
}

}  // namespace morl
