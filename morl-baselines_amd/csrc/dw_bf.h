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
    long long* prof;           // development probe only (tools/probes/dwb_probe.hip): [jobs][DWB_PROF_SLOTS] phase cycle sums
};
constexpr int DWB_PROF_SLOTS = 16;

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
    const float* base;         // the operand matrix (wave-uniform, like ld / kend / live / is_g)
    int kend;
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
    it.live = first < n_g + n_h;                               // (wave-uniform too: both item counts are multiples of 64)
    const int col_local = 2 * cp;
    const int col = (it.is_g ? m0 : 0) + col_local;
    it.ld = it.is_g ? g.ldg : g.ldh;
    const bool col_ok = it.live && col < (it.is_g ? g.gcols : g.hcols);     // (column counts are even: pairs are in or out as a whole)
    // rows [0, kend) of the operand: everything beyond this split's slice reads as zero
    it.base = it.is_g ? g.G : g.H;
    it.kend = kend;
    it.base_off = col_ok ? (8 * oct * it.ld + col) * 4 : DW2_OOB;
    const int tile = (it.is_g ? 0 : g.tg) + (col_local >> 4);
    it.dst_off = ((tile * 3) * 64 + (col_local & 15) + 16 * oct) * 16;
    return it;
}

// The chunk's first row goes into the DESCRIPTOR (base + k0 rows, range = what is left of the slice: scalar arithmetic, and rows
// beyond the slice still read as zeros), the row inside the chunk through eight constant VECTOR offsets (the hardware's range
// check does not see the scalar offset).  Recomputing vector addresses per chunk cost a producer wave 24 vector instructions
// -- and hipcc built them in registers that loads still in flight were going to write, waiting for those loads first.
__device__ __forceinline__ void dwb_load(DwbStage& s, const DwbItem& it, const int (&voff)[8], int k0) {
    const int left = max(it.kend - k0, 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(it.base + (size_t)k0 * it.ld), 0, left * it.ld * 4, 0x00020000);
#pragma unroll
    for (int e = 0; e < 8; ++e) s.v[e] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[e], 0, 0));
}

// split the item's two columns (8 rows each) and write them as fragment lanes: column j -> lane (col % 16) + 16 octet of its tile.
// STAGE-WISE over the item's eight (row pair, column) chains: convert all, subtract all, convert all ... -- written chain by chain
// (bf_split2 per pair) hipcc emitted the 8 x 9 dependent instructions as ONE serial chain through a single register pair, and a producer
// wave is alone on its SIMD's vector ALU: nothing hid the ~9-cycle dependent-issue latency (1 700 cycles per item, the launch's
// critical path: tools/probes/dwb_probe.hip).  The scheduling fences keep the stages apart.
__device__ __forceinline__ void dwb_split_store(const DwbStage& s, unsigned char* lane_dst) {
    float a[8], b[8];                      // chain i = 4 j + u: rows 2 u, 2 u + 1 of column j
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[4 * j + u] = (j == 0) ? s.v[2 * u].x : s.v[2 * u].y;
            b[4 * j + u] = (j == 0) ? s.v[2 * u + 1].x : s.v[2 * u + 1].y;
        }
    unsigned hi[8], mid[8], lo[8];
    float ua[8], ub[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = bf_pack2(a[i], b[i]);
    BF_PIN();
#pragma unroll
    for (int i = 0; i < 8; ++i) { ua[i] = bf_low(hi[i]); ub[i] = bf_high(hi[i]); }
    BF_PIN();
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = bf_sub(a[i], ua[i]); b[i] = bf_sub(b[i], ub[i]); }
    BF_PIN();
#pragma unroll
    for (int i = 0; i < 8; ++i) mid[i] = bf_pack2(a[i], b[i]);
    BF_PIN();
#pragma unroll
    for (int i = 0; i < 8; ++i) { ua[i] = bf_low(mid[i]); ub[i] = bf_high(mid[i]); }
    BF_PIN();
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = bf_sub(a[i], ua[i]); b[i] = bf_sub(b[i], ub[i]); }
    BF_PIN();
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = bf_pack2(a[i], b[i]);
    BF_PIN();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        *reinterpret_cast<bf_u32x4*>(lane_dst + j * 16) = bf_u32x4{hi[4 * j], hi[4 * j + 1], hi[4 * j + 2], hi[4 * j + 3]};
        *reinterpret_cast<bf_u32x4*>(lane_dst + BF_BLOCK + j * 16) = bf_u32x4{mid[4 * j], mid[4 * j + 1], mid[4 * j + 2], mid[4 * j + 3]};
        *reinterpret_cast<bf_u32x4*>(lane_dst + 2 * BF_BLOCK + j * 16) = bf_u32x4{lo[4 * j], lo[4 * j + 1], lo[4 * j + 2], lo[4 * j + 3]};
    }
}

// PRODUCER waves (4 .. 7, 256 work-items): per chunk each work-item splits up to three items.
// CONSUMER waves (0 .. 3): N_OT x N_IT output tiles each, every MFMA unconditional (tile counts are compile-time; operand tiles
// beyond the matrix are zeros the producers wrote), the N_OT accumulators of an input tile alternating.
template <int N_OT, int N_IT, bool PROF = false>
__device__ __forceinline__ void dwb_job(const DwbProblem& g, int group, int split, int rows, long long slab_stride, unsigned char* lds,
                                        long long* prof = nullptr) {
    const int tid = (int)threadIdx.x, lane = tid & 63;
    long long pt[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, wall0 = 0;
    if (PROF) { tprev = clock64(); wall0 = wall_clock64(); }
#define DWB_TICK(q) if (PROF) { const long long t_ = clock64(); pt[q] += t_ - tprev; tprev = t_; }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kbeg = split * g.k_per_split;
    const int kend = min(rows, kbeg + g.k_per_split);
    const int m0 = (g.layout == 0) ? group * 128 : 0;          // first output row of this job's g block
    const bool producer = wave >= 4;

    if (producer) {
        const int pwave = wave & 3;
        DwbItem it[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) it[k] = dwb_item(g, m0, k, pwave, lane, kend);
        float csum[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // running column sums of this work-item's g columns (db)
        // two register stages: chunk c + 1 waits in one to be split while chunk c + 2 is in flight into the other -- a chunk is ~1.5 us
        // of matrix-core time, a loaded HBM round trip about as long: with ONE stage (the first producer / consumer version) every
        // chunk waited for its operands and the launch ran at the memory LATENCY (96 us)
        DwbStage st[2][3];
        int voff[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) voff[k][e] = it[k].base_off + e * it[k].ld * 4;
        auto put = [&](int sidx, unsigned char* buf) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (it[k].live) {
                    if (it[k].is_g) {
                        const DwbStage& q = st[sidx][k];      // (pairwise: three dependent additions per column instead of eight)
                        csum[k][0] = bf_add(csum[k][0], bf_add(bf_add(bf_add(q.v[0].x, q.v[1].x), bf_add(q.v[2].x, q.v[3].x)),
                                                               bf_add(bf_add(q.v[4].x, q.v[5].x), bf_add(q.v[6].x, q.v[7].x))));
                        csum[k][1] = bf_add(csum[k][1], bf_add(bf_add(bf_add(q.v[0].y, q.v[1].y), bf_add(q.v[2].y, q.v[3].y)),
                                                               bf_add(bf_add(q.v[4].y, q.v[5].y), bf_add(q.v[6].y, q.v[7].y))));
                    }
                    dwb_split_store(st[sidx][k], buf + it[k].dst_off);
                }
        };
        auto load = [&](int sidx, int k0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dwb_load(st[sidx][k], it[k], voff[k], k0);
        };
        // prologue: chunk 0 split into buffer 0, chunks 1 and 2 in flight
        load(0, kbeg);
        load(1, kbeg + DWB_BK);
        put(0, lds);
        load(0, kbeg + 2 * DWB_BK);
        DWB_BARRIER();
        DWB_TICK(0)
        // iteration c: chunk c is being multiplied out of buffer c & 1; chunk c + 1 (stage (c + 1) & 1) goes to the other buffer,
        // chunk c + 3 into flight behind it (into the stage just emptied); chunk c + 2 stays in flight across the barrier
        for (int k0 = kbeg; k0 < kend; k0 += 2 * DWB_BK) {
            if (PROF) { BF_VMCNT(24); DWB_TICK(1) }   // vmcnt(24): this stage's operands are here
            if (k0 + DWB_BK < kend) put(1, lds + DWB_BUF_BYTES);
            DWB_TICK(2)
            load(1, k0 + 3 * DWB_BK);
            DWB_TICK(3)
            DWB_BARRIER();
            DWB_TICK(4)
            if (k0 + DWB_BK < kend) {
                if (PROF) { BF_VMCNT(24); DWB_TICK(1) }
                if (k0 + 2 * DWB_BK < kend) put(0, lds);
                DWB_TICK(2)
                load(0, k0 + 4 * DWB_BK);
                DWB_TICK(3)
                DWB_BARRIER();
                DWB_TICK(4)
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
        const int ptid = tid & 255;
        if (g.colsum != nullptr && ptid < 16 * g.tg && m0 + ptid < g.M)
            g.colsum[(size_t)split * slab_stride + m0 + ptid] = ((scr[ptid] + scr[256 + ptid]) + scr[512 + ptid]) + scr[768 + ptid];
        DWB_TICK(5)
        if (PROF && prof != nullptr && (tid & 255) == 0) {
            for (int q = 0; q < 6; ++q) prof[8 + q] = pt[q];
        }
        return;
    }

    // ---- consumer: output tiles rows 16 (ot0 + a), columns 16 (it0 + b) ------------------------------------------------------------
    int ot0, it0;
    const int cwave = wave & 3;
    if (g.layout == 0) { ot0 = 4 * (cwave >> 1); it0 = 8 * (cwave & 1); }          // 128 x 256: 2 x 2 waves of 64 x 128
    else if (g.layout == 1) { ot0 = 4 * cwave; it0 = 0; }                          // 256 x 64: 4 x 1 waves of 64 x 64
    else { ot0 = 0; it0 = 4 * cwave; }                                            // 32 x 256: 1 x 4 waves of 32 x 64
    f32x4 acc[N_OT][N_IT];
#pragma unroll
    for (int a = 0; a < N_OT; ++a)
#pragma unroll
        for (int b = 0; b < N_IT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    DWB_BARRIER();                                             // (the producers' prologue)
    DWB_TICK(0)
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
        if (PROF) { __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8)); DWB_TICK(1) }
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
        DWB_TICK(2)
        DWB_BARRIER();
        DWB_TICK(3)
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
    DWB_TICK(4)
    if (PROF && prof != nullptr && (tid & 255) == 0) {
        for (int q = 0; q < 5; ++q) prof[q] = pt[q];
        prof[6] = wall0; prof[7] = wall_clock64();
    }
#undef DWB_TICK
}

template <bool PROF>
__device__ __forceinline__ void dw_bf_body(const DwbArgs& a) {
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
    long long* prof = (PROF && a.prof != nullptr) ? a.prof + (size_t)job * DWB_PROF_SLOTS : nullptr;
    if (g.layout == 0) dwb_job<4, 8, PROF>(g, local % g.groups, local / g.groups, a.rows, a.slab_stride, lds, prof);
    else if (g.layout == 1) dwb_job<4, 4, PROF>(g, 0, local, a.rows, a.slab_stride, lds, prof);
    else dwb_job<2, 4, PROF>(g, 0, local, a.rows, a.slab_stride, lds, prof);
}
__global__ __launch_bounds__(DWB_THREADS) void dw_bf_kernel(DwbArgs a) { dw_bf_body<false>(a); }

}  // namespace morl
