// Weight gradients of one backward pass on the bf16 matrix cores with fp32-class accuracy (gfx950, wave64): one launch, split-K
// over the batch rows, every fp32 product evaluated as six products of three-way bf16 splits (see mlp_chain_bf.h for the arithmetic),
//
//   dW_l[o][i] = sum_rows g_l[row][o] * h_l[row][i]        db_l[o] = sum_rows g_l[row][o]        (the dW / db half of loss.backward(),
//                                                                                                 envelope.py:323)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 inside a row slice, slices summed in a fixed order by grad_reduce_ranges_kernel
// (dw_tiles.h) exactly like the fp32 engine's.  Both operands are fp32 in HBM (what the forward / backward chains saved); a
// workgroup splits its slices ON THE FLY:
//   * 512 work-items = 8 waves, one workgroup per CU, SPECIALISED: waves 4-7 are producers (loads, split, fragment writes, db:
//     vector ALU only), waves 0-3 consumers (fragment reads + MFMAs only) -- one of each per SIMD, so the split of chunk c + 1 runs on
//     the vector ALU while chunk c is on the matrix pipe.  (The first version gave every wave both jobs and a run-time tile count:
//     a branch around every MFMA and the split serialised in front of them -- 143 us against the fp32 engine's 69.)
//     A job = (problem, group of output rows, row slice); per 32-row chunk the producers turn a 32 x (16 TG) block of g and a
//     32 x (16 TH) block of h into MFMA fragments: a work-item takes up to three (column pair, row octet) items, loads 8 x 8 bytes each
//     (branch-free buffer loads: rows beyond the slice and columns beyond the matrix read as zeros), splits the 16 values and
//     writes each column's eight rows as one 16-byte fragment lane per split part -- fragment block = (operand tile, part):
//     1 KB, lane (column, octet) at 16 (column + 16 octet), so the MFMA phase's ds_read_b128 is 1 KB contiguous, conflict-free.
//     db rides along: the g items keep running column sums (combined over the four octets in a fixed order at the end).
//   * TG + TH <= 24 tiles = 72 KB per chunk, double-buffered (144 KB): chunk c is multiplied while chunk c + 1 is split and written
//     and chunk c + 2 is in flight from memory; ONE barrier per chunk (96 MFMAs per wave).
//   * a wave owns up to 2 x 8 output tiles of 16 x 16 (64 accumulator registers); which ones depends on the problem's shape (Layout):
//     256 x 256 layers: two groups of 128 output rows, wave (w >> 1, w & 1) -> rows 32 (w >> 1).., columns 128 (w & 1)..;
//     a narrow input (the first layer, in <= 128): all 256 output rows in one group, wave w -> rows 32 w.., all columns;
//     a narrow output (the Q head, out <= 32): wave w -> all rows, columns 32 w...
//     Row slices are sized by the MFMAs a chunk costs the busiest wave, so every job of the launch carries the same matrix-core work.
// Roofline: 6 x (2 * rows * out * in) flop per layer on the bf16 pipe; HBM bytes = both operands once or twice (an operand block is
// shared by the row groups / column halves of its layer) + the split-K slabs.
#pragma once
#include "dw_tiles.h"
#include "mlp_chain_bf.h"

namespace morl {

constexpr int DWB_THREADS = 512;
constexpr int DWB_BK = 32;                       // rows per chunk = one MFMA k-step
constexpr int DWB_MAX_TILES = 24;                // operand tiles (g + h) per chunk: 72 KB
constexpr int DWB_BUF_BYTES = DWB_MAX_TILES * 3 * BF_BLOCK;
constexpr int DWB_LDS_BYTES = 2 * DWB_BUF_BYTES;

struct DwbProblem {
    const float* G;   // [rows][ldg]  dLoss/dz_l          (output index contiguous)
    const float* H;   // [rows][ldh]  layer input          (input index contiguous)
    float* C;         // slab 0 of dW_l, row-major [M][ldc]
    float* colsum;    // slab 0 of db_l [M]
    int M, N;         // out, in
    int ldg, ldh, ldc;
    int gcols, hcols; // loadable columns of G / H: multiples of 4, pad columns are zeros
    int layout;       // 0: groups of 128 output rows x all (<= 256) columns; 1: all (<= 256) output rows x <= 128 columns; 2: <= 32 rows x <= 256 columns
    int groups;       // output-row groups (layout 0: ceil(M / 128), else 1)
    int tg, th;       // operand tiles of a chunk: g (16 output rows each) and h (16 input columns each)
    int k_per_split, splits;
    int job_start;    // job = job_start + split * groups + group
};

struct DwbArgs {
    DwbProblem p[MORL_MAX_LAYERS];
    int n, rows, jobs;
    long long slab_stride;     // floats between split slabs
    SumTreeUpdate per;         // per.tree != NULL: one extra workgroup (block `jobs`) applies the step's PER priority update (as dw_tiles.h)
};

// chunk boundary: this wave's fragment writes have landed in LDS (lgkmcnt(0)), then the workgroup barrier -- NOT __syncthreads(),
// whose release fence also waits for the vector-memory counter, i.e. for the operand loads of the chunk after next that are
// meant to stay in flight across the barrier
#define DWB_BARRIER() do { __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8)); __builtin_amdgcn_s_barrier(); } while (0)

// the 8 x 8 bytes of one split item: rows k0 + 8 octet + e of TWO consecutive columns
struct DwbStage { float2 v[8]; };

// one split item of a producer work-item: (column pair cp, row octet oct) of the g block or of the h block, or idle.  A chunk has
// 32 (TG + TH) <= 768 items = at most three per producer work-item (slot k: items [256 k, 256 k + 256)); the g items come first and
// 32 TG is a multiple of 64, so every (slot, wave) belongs to ONE operand and its buffer descriptor is wave-uniform (chosen per LANE
// it made hipcc wrap every load into a waterfall loop).  Column PAIRS, not quads: three items per work-item for the 24-tile
// chunk of a 256 x 256 layer -- every producer wave carries the same work (with quads waves 4, 5 had two items and 6, 7 one) --
// and the eight lanes of a ds_write_b128 group then cover eight different bank quads (quads: 2-way conflicts on every write).
struct DwbItem {
    __amdgpu_buffer_rsrc_t rsrc;
    int base_off, ld, dst_off;
    bool live, is_g;
};
__device__ __forceinline__ DwbItem dwb_item(const DwbProblem& g, int m0, int slot, int pwave, int lane, int kend) {
    DwbItem it;
    const int first = (slot * 4 + pwave) * 64;                 // first item of this (slot, wave)
    const int n_g = 32 * g.tg, n_h = 32 * g.th;
    it.is_g = first < n_g;                                     // (wave-uniform)
    const int local = first + lane - (it.is_g ? 0 : n_g);      // item index inside the operand
    const int npairs = 8 * (it.is_g ? g.tg : g.th);            // column pairs of the operand block
    const int cp = local % npairs, oct = local / npairs;
    it.live = local < (it.is_g ? n_g : n_h);
    const int col_local = 2 * cp;
    const int col = (it.is_g ? m0 : 0) + col_local;
    it.ld = it.is_g ? g.ldg : g.ldh;
    const bool col_ok = it.live && col < (it.is_g ? g.gcols : g.hcols);     // (column counts are even: pairs are in or out as a whole)
    // rows [0, kend) of the operand: everything beyond this split's slice reads as zero
    it.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(it.is_g ? g.G : g.H), 0, kend * it.ld * 4, 0x00020000);
    it.base_off = col_ok ? (8 * oct * it.ld + col) * 4 : DW2_OOB;
    const int tile = (it.is_g ? 0 : g.tg) + (col_local >> 4);
    it.dst_off = ((tile * 3) * 64 + (col_local & 15) + 16 * oct) * 16;
    return it;
}

__device__ __forceinline__ void dwb_load(DwbStage& s, const DwbItem& it, int k0) {
    // (the row goes through the VECTOR offset: the hardware's range check does not see the scalar one)
#pragma unroll
    for (int e = 0; e < 8; ++e)
        s.v[e] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(it.rsrc, it.base_off + (k0 + e) * it.ld * 4, 0, 0));
}

// split the item's two columns (8 rows each) and write them as fragment lanes: column j -> lane (col % 16) + 16 octet of its tile
__device__ __forceinline__ void dwb_split_store(const DwbStage& s, unsigned char* lane_dst) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = (j == 0) ? s.v[e].x : s.v[e].y;
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bf_split2(c[2 * u], c[2 * u + 1], hi[u], mid[u], lo[u]);
        *reinterpret_cast<bf_u32x4*>(lane_dst + j * 16) = bf_u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<bf_u32x4*>(lane_dst + BF_BLOCK + j * 16) = bf_u32x4{mid[0], mid[1], mid[2], mid[3]};
        *reinterpret_cast<bf_u32x4*>(lane_dst + 2 * BF_BLOCK + j * 16) = bf_u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
}

// PRODUCER waves (4 .. 7, 256 work-items): per chunk each work-item splits up to three items.
// CONSUMER waves (0 .. 3): N_OT x N_IT output tiles each, every MFMA unconditional (tile counts are compile-time; operand tiles
// beyond the matrix are zeros the producers wrote), the N_OT accumulators of an input tile alternating.
template <int N_OT, int N_IT>
__device__ __forceinline__ void dwb_job(const DwbProblem& g, int group, int split, int rows, long long slab_stride, unsigned char* lds) {
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kbeg = split * g.k_per_split;
    const int kend = min(rows, kbeg + g.k_per_split);
    const int m0 = (g.layout == 0) ? group * 128 : 0;          // first output row of this job's g block
    const bool producer = wave >= 4;

    if (producer) {
        const int pwave = wave - 4;
        DwbItem it[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) it[k] = dwb_item(g, m0, k, pwave, lane, kend);
        float csum[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // running column sums of this work-item's g columns (db)
        // two register stages: chunk c + 1 waits in one to be split while chunk c + 2 is in flight into the other -- a chunk is ~1.5 us
        // of matrix-core time, a loaded HBM round trip about as long: with ONE stage (the first producer / consumer version) every
        // chunk waited for its operands and the launch ran at the memory LATENCY (96 us)
        DwbStage st[2][3];
        auto put = [&](int sidx, unsigned char* buf) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (it[k].live) {
                    if (it[k].is_g) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { csum[k][0] += st[sidx][k].v[e].x; csum[k][1] += st[sidx][k].v[e].y; }
                    }
                    dwb_split_store(st[sidx][k], buf + it[k].dst_off);
                }
        };
        auto load = [&](int sidx, int k0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dwb_load(st[sidx][k], it[k], k0);
        };
        // prologue: chunk 0 split into buffer 0, chunks 1 and 2 in flight
        load(0, kbeg);
        load(1, kbeg + DWB_BK);
        put(0, lds);
        load(0, kbeg + 2 * DWB_BK);
        DWB_BARRIER();
        // iteration c: chunk c is being multiplied out of buffer c & 1; chunk c + 1 (stage (c + 1) & 1) goes to the other buffer,
        // chunk c + 3 into flight behind it (into the stage just emptied); chunk c + 2 stays in flight across the barrier
        for (int k0 = kbeg; k0 < kend; k0 += 2 * DWB_BK) {
            if (k0 + DWB_BK < kend) put(1, lds + DWB_BUF_BYTES);
            load(1, k0 + 3 * DWB_BK);
            DWB_BARRIER();
            if (k0 + DWB_BK < kend) {
                if (k0 + 2 * DWB_BK < kend) put(0, lds);
                load(0, k0 + 4 * DWB_BK);
                DWB_BARRIER();
            }
        }
        // db: the four octets of a g column, in octet order (the operand buffers are free: the loop ended with a barrier)
        float* scr = reinterpret_cast<float*>(lds);            // [4 octets][256 columns]
        if (g.colsum != nullptr) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (it[k].live && it[k].is_g) {
                    const int local = (k * 4 + pwave) * 64 + lane, npairs = 8 * g.tg;
                    const int cp = local % npairs, oct = local / npairs;
                    scr[oct * 256 + 2 * cp] = csum[k][0];
                    scr[oct * 256 + 2 * cp + 1] = csum[k][1];
                }
        }
        __syncthreads();
        const int ptid = tid - 256;
        if (g.colsum != nullptr && ptid < 16 * g.tg && m0 + ptid < g.M)
            g.colsum[(size_t)split * slab_stride + m0 + ptid] = ((scr[ptid] + scr[256 + ptid]) + scr[512 + ptid]) + scr[768 + ptid];
        return;
    }

    // ---- consumer: output tiles rows 16 (ot0 + a), columns 16 (it0 + b) ------------------------------------------------------------
    int ot0, it0;
    if (g.layout == 0) { ot0 = 4 * (wave >> 1); it0 = 8 * (wave & 1); }          // 128 x 256: 2 x 2 waves of 64 x 128
    else if (g.layout == 1) { ot0 = 4 * wave; it0 = 0; }                          // 256 x 64: 4 x 1 waves of 64 x 64
    else { ot0 = 0; it0 = 4 * wave; }                                            // 32 x 256: 1 x 4 waves of 32 x 64
    f32x4 acc[N_OT][N_IT];
#pragma unroll
    for (int a = 0; a < N_OT; ++a)
#pragma unroll
        for (int b = 0; b < N_IT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    DWB_BARRIER();                                             // (the producers' prologue)
    const unsigned char* frag_lane = lds + lane * 16;
    constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};
    for (int k0 = kbeg, c = 0; k0 < kend; k0 += DWB_BK, ++c) {
        const unsigned char* cur = frag_lane + (c & 1) * DWB_BUF_BYTES;
        const unsigned char* hb = cur + (g.tg + it0) * 3 * BF_BLOCK;
        // A fragments (the wave's g tiles): once per chunk; B fragments (h tiles): one tile ahead
        bf_u32x4 fa[N_OT][3], fb[2][3];
#pragma unroll
        for (int a = 0; a < N_OT; ++a) bf_frag_load(fa[a], cur + (ot0 + a) * 3 * BF_BLOCK, 0);
        bf_frag_load(fb[0], hb, 0);
        BF_PIN();
#pragma unroll
        for (int b = 0; b < N_IT; ++b) {
            // D[o][i] += g^T[o][k] h[k][i]: A = the g fragment (lane = output row), B = the h fragment (lane = input column)
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                if (b + 1 < N_IT && p < 3) fb[(b + 1) & 1][2 - p] = *reinterpret_cast<const bf_u32x4*>(hb + (3 * (b + 1) + 2 - p) * BF_BLOCK);
#pragma unroll
                for (int a = 0; a < N_OT; ++a) acc[a][b] = bf_mfma(fa[a][pw[p]], fb[b & 1][px[p]], acc[a][b]);
                BF_PIN();
            }
        }
        DWB_BARRIER();
    }

    // ---- epilogue: slab tiles; D register r of lane (i, q) of tile (a, b) is row 16 (ot0 + a) + 4 q + r, column 16 (it0 + b) + i --------
    float* __restrict__ C = g.C + (size_t)split * slab_stride;
    const int li = lane & 15, q = lane >> 4;
#pragma unroll
    for (int a = 0; a < N_OT; ++a)
#pragma unroll
        for (int b = 0; b < N_IT; ++b) {
            const int colc = 16 * (it0 + b) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * (ot0 + a) + 4 * q + r;
                if (row < g.M && colc < g.N) C[(size_t)row * g.ldc + colc] = acc[a][b][r];
            }
        }
    __syncthreads();                                           // (pairs with the producers' db barrier)
}

__global__ __launch_bounds__(DWB_THREADS) void dw_bf_kernel(DwbArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[DWB_LDS_BYTES];
    static_assert(DWB_LDS_BYTES >= ST_LDS_BYTES, "the tree update borrows the operand buffers as scratch");
    if ((int)blockIdx.x >= a.jobs) {
        // (the tree update is written for ST_THREADS work-items; the surplus of this launch's 512 idles)
        if ((int)blockIdx.x == a.jobs && a.per.tree != nullptr) sumtree_update_body(a.per, lds);
        return;
    }
    // consecutive jobs = the output-row groups of one row slice, which read the same slice of h_l: keep them on one XCD's L2 (the
    // dispatcher places block b on XCD b % 8)
    const int job = xcd_remap((int)blockIdx.x, a.jobs);
    int q = 0;
    while (q + 1 < a.n && job >= a.p[q + 1].job_start) ++q;
    const DwbProblem& g = a.p[q];
    const int local = job - g.job_start;
    if (g.layout == 0) dwb_job<4, 8>(g, local % g.groups, local / g.groups, a.rows, a.slab_stride, lds);
    else if (g.layout == 1) dwb_job<4, 4>(g, 0, local, a.rows, a.slab_stride, lds);
    else dwb_job<2, 4>(g, 0, local, a.rows, a.slab_stride, lds);
}

}  // namespace morl
