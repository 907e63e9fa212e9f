// Weight gradients of one backward pass on the bf16 matrix cores with fp32-class accuracy (gfx950, wave64): one launch, split-K
// over the batch rows, every fp32 product evaluated as six products of three-way bf16 splits (see mlp_chain_bf.h for the arithmetic),
//
//   dW_l[o][i] = sum_rows g_l[row][o] * h_l[row][i]        db_l[o] = sum_rows g_l[row][o]        (the dW / db half of loss.backward(),
//                                                                                                 envelope.py:323)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 inside a row slice, slices summed in a fixed order by grad_reduce_ranges_kernel
// (dw_tiles.h) exactly like the fp32 engine's.  Both operands are fp32 in HBM (what the forward / backward chains saved); a
// workgroup splits its slices ON THE FLY:
//   * 768 work-items = 12 waves, one workgroup per CU, SPECIALISED: waves 0-3 are consumers (fragment reads + MFMAs only, one per
//     SIMD), waves 4-11 producers (loads, split, fragment writes, db: vector ALU only, TWO per SIMD).  What sets the shape: beside a
//     wave that keeps the SIMD's matrix pipe busy, ANOTHER wave issues one plain vector instruction per ~9.7 cycles, however many
//     such waves there are (tools/probes/valu_mfma_probe.hip, profiles/r04_valu_issue_probe.txt) -- the split is ~7 vector
//     instructions per element, so ONE producer wave per SIMD needed 5 300 cycles per chunk against the consumers' 3 100
//     (profiles/r04_dw_bf_probe.txt: 55 us per launch); two producer waves each carrying a third of that work stay under the
//     consumers' time.  Three waves per SIMD leave a wave 168 registers: the workgroup's output tile is 128 x 128 (64 accumulator
//     registers per consumer), not 128 x 256.  Measured per 32-row chunk of a 256 x 256 layer's job: consumers 1 970 cycles (96 MFMAs =
//     1 536 + 400 of exposed fragment reads), the older producer wave of a SIMD 1 810, the younger 2 330 (it loses the issue
//     arbitration; raising its priority moves the loss to the other two, zero-sum), + ~300 for the barrier: the launch is ISSUE
//     bound -- three waves put ~460 instructions per chunk through one SIMD.
//     (History: every wave doing both jobs with run-time tile counts, 143 us; one stage of operand prefetch, 96; 8 waves, 55; this, 50.)
//   * a job = (problem, group of <= 128 output rows, group of <= 128 input columns, row slice); per 32-row chunk the producers
//     turn a 32 x (16 TG) block of g and a 32 x (16 TH) block of h (TG + TH <= 16 operand tiles) into MFMA fragments: ONE (column
//     pair, row octet) item per producer work-item -- 8 x 8 bytes loaded (branch-free buffer loads: rows beyond the slice and
//     columns beyond the matrix read as zeros), 16 values split, each column's eight rows written as one 16-byte fragment lane per
//     split part -- fragment block = (operand tile, part): 1 KB, lane (column, octet) at 16 (column + 16 octet), so the MFMA
//     phase's ds_read_b128 is 1 KB contiguous.  db rides along: the g items keep running column sums.
//   * 48 KB per chunk, double-buffered (96 KB): chunk c is multiplied while chunk c + 1 is split and written and chunks c + 2, c + 3
//     are in flight from memory (two register stages); ONE barrier per chunk.
//   * three shapes (compile-time tile counts, every MFMA unconditional):
//       0  out > 32, in > 64:  TG = 8, TH = 8, consumers 2 x 2, each 4 x 4 tiles (96 MFMAs per chunk)
//       1  in <= 64 (the first layer):  TG = 8, TH = 4, consumers 4 x 1, each 2 x 4 tiles
//       2  out <= 32 (the Q head):  TG = 2, TH = 8, consumers 1 x 4, each 2 x 2 tiles
//     Jobs are balanced by their CHUNK count: whatever the shape, a chunk is not shorter than the producers' load -> split -> write
//     turn-around.
// Roofline: 6 x (2 * rows * out * in) flop per layer on the bf16 pipe; HBM bytes = the operand blocks (an operand block is shared by
// the jobs of its row slice: L2) + the split-K slabs.
#pragma once
#include "dw_tiles.h"
#include "mlp_chain_bf.h"

namespace morl {

constexpr int DWB_THREADS = 768;
constexpr int DWB_CONSUMERS = 256;               // work-items of the consumer waves (0 .. 3)
constexpr int DWB_PRODUCERS = DWB_THREADS - DWB_CONSUMERS;
constexpr int DWB_BK = 32;                       // rows per chunk = one MFMA k-step
constexpr int DWB_MAX_TILES = 16;                // operand tiles (g + h) per chunk: 48 KB = one item per producer work-item
constexpr int DWB_BUF_BYTES = DWB_MAX_TILES * 3 * BF_BLOCK;
constexpr int DWB_LDS_BYTES = 2 * DWB_BUF_BYTES;
static_assert(32 * DWB_MAX_TILES == DWB_PRODUCERS, "a chunk is one (column pair, row octet) item per producer work-item");

// operand tiles of a chunk per shape: g (16 output rows each), h (16 input columns each)
constexpr int DWB_TG[3] = {8, 8, 2}, DWB_TH[3] = {8, 4, 8};

struct DwbProblem {
    const float* G;   // [rows][ldg]  dLoss/dz_l          (output index contiguous)
    const float* H;   // [rows][ldh]  layer input          (input index contiguous)
    float* C;         // slab 0 of dW_l, row-major [M][ldc]
    float* colsum;    // slab 0 of db_l [M]
    int M, N;         // out, in
    int ldg, ldh, ldc;
    int gcols, hcols; // loadable columns of G / H: even, pad columns are zeros
    int shape;        // 0 / 1 / 2, see above
    int mgroups;      // groups of 16 TG output rows: ceil(M / (16 TG))
    int ngroups;      // groups of 16 TH input columns: ceil(N / (16 TH))
    int k_per_split, splits;
    int job_start;    // job = job_start + (split * mgroups + mgroup) * ngroups + ngroup
};

struct DwbArgs {
    DwbProblem p[MORL_MAX_LAYERS];
    int n, rows, jobs;
    long long slab_stride;     // floats between split slabs
    SumTreeUpdate per;         // per.tree != NULL: one extra workgroup (block `jobs`) applies the step's PER priority update (as dw_tiles.h)
    long long* prof;           // development probe only (tools/probes/dwb_probe.hip): [jobs][DWB_PROF_SLOTS] phase cycle sums
};
constexpr int DWB_PROF_SLOTS = 50;        // per wave w: [4 w .. 4 w + 3] = prologue, work, barrier wait, epilogue (cycle sums); 48, 49: wall clock

// chunk boundary: this wave's fragment writes have landed in LDS (lgkmcnt(0)), then the workgroup barrier -- NOT __syncthreads(),
// whose release fence also waits for the vector-memory counter, i.e. for the operand loads of the chunks after next that are
// meant to stay in flight across the barrier
#define DWB_BARRIER() do { __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8)); __builtin_amdgcn_s_barrier(); } while (0)

// the 8 x 8 bytes of a split item: rows k0 + 8 octet + e of TWO consecutive columns
struct DwbStage { float2 v[8]; };

// The split item of a producer work-item: (column pair, row octet) of the g block or of the h block, or idle.  The g items come
// first and 32 TG is a multiple of 64, so a producer WAVE belongs to one operand (or idles as a whole): its buffer descriptor is
// wave-uniform (chosen per LANE it made hipcc wrap every load into a waterfall loop) and `is_g` / `live` are scalar branches.
// Column PAIRS: the eight lanes of a ds_write_b128 group cover eight different 16-byte slots (quads: 2-way conflicts on every write).
struct DwbItem {
    const float* base;         // the operand matrix (wave-uniform, like ld / kend / live / is_g)
    int kend, ld;
    int dst_off;
    bool live, is_g;
};
template <int TG, int TH>
__device__ __forceinline__ DwbItem dwb_item(const DwbProblem& g, int m0, int n0, int pwave, int lane, int kend, int (&voff)[8]) {
    static_assert((32 * TG) % 64 == 0 && (32 * TH) % 64 == 0 && TG + TH <= DWB_MAX_TILES, "operand blocks are whole waves of items");
    DwbItem it;
    constexpr int n_g = 32 * TG, n_h = 32 * TH;
    const int first = pwave * 64;                              // first item of this wave
    it.is_g = first < n_g;
    it.live = first < n_g + n_h;
    const int local = first + lane - (it.is_g ? 0 : n_g);      // item index inside the operand
    const int npairs = 8 * (it.is_g ? TG : TH);                // column pairs of the operand block
    const int cp = local % npairs, oct = local / npairs;
    const int col_local = 2 * cp;
    const int col = (it.is_g ? m0 : n0) + col_local;
    it.ld = it.is_g ? g.ldg : g.ldh;
    const bool col_ok = it.live && col < (it.is_g ? g.gcols : g.hcols);     // (column counts are even: pairs are in or out as a whole)
    it.base = it.is_g ? g.G : g.H;
    it.kend = kend;
    // the row inside the chunk goes through eight constant VECTOR offsets (the hardware's range check does not see the scalar
    // offset); the chunk's first row through the descriptor, see dwb_load
    const int base_off = col_ok ? (8 * oct * it.ld + col) * 4 : DW2_OOB;
#pragma unroll
    for (int e = 0; e < 8; ++e) voff[e] = base_off + e * it.ld * 4;
    const int tile = (it.is_g ? 0 : TG) + (col_local >> 4);
    it.dst_off = ((tile * 3) * 64 + (col_local & 15) + 16 * oct) * 16;
    return it;
}

// The chunk's first row goes into the DESCRIPTOR (base + k0 rows, range = what is left of the slice: scalar arithmetic, and rows
// beyond the slice still read as zeros).  Recomputing vector addresses per chunk cost a producer wave 8 vector instructions per
// item -- and hipcc built them in registers that loads still in flight were going to write, waiting for those loads first.
__device__ __forceinline__ void dwb_load(DwbStage& s, const DwbItem& it, const int (&voff)[8], int k0) {
    const int left = max(it.kend - k0, 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(it.base + (size_t)k0 * it.ld), 0, left * it.ld * 4, 0x00020000);
#pragma unroll
    for (int e = 0; e < 8; ++e) s.v[e] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[e], 0, 0));
}

// split the item's two columns (8 rows each) and write them as fragment lanes: column j -> lane (col % 16) + 16 octet of its tile.
// STAGE-WISE over the item's eight (row pair, column) chains -- convert all, unpack all, subtract all ... -- with scheduling fences
// between the stages (chain by chain, hipcc emitted one serial dependency chain through a single register pair), and with scalar
// subtractions (bf_sub: v_pk_add_f32 is 2.5 x a plain instruction beside a busy matrix pipe and wants register pairs).
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

// One job.  SHAPE fixes TG / TH and the consumers' arrangement: WN waves side by side, each N_OT x N_IT output tiles.
template <int SHAPE, bool PROF = false>
__device__ __forceinline__ void dwb_job(const DwbProblem& g, int mgroup, int ngroup, int split, int rows, long long slab_stride,
                                        unsigned char* lds, long long* prof = nullptr) {
    constexpr int TG = DWB_TG[SHAPE], TH = DWB_TH[SHAPE];
    constexpr int WN = SHAPE == 0 ? 2 : SHAPE == 1 ? 1 : 4, WM = 4 / WN;
    constexpr int N_OT = TG / WM, N_IT = TH / WN;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long pt[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, wall0 = 0;
    if (PROF) { tprev = clock64(); wall0 = wall_clock64(); }
#define DWB_TICK(q) if (PROF) { const long long t_ = clock64(); pt[q] += t_ - tprev; tprev = t_; }
    const int kbeg = split * g.k_per_split;
    const int kend = min(rows, kbeg + g.k_per_split);
    const int m0 = mgroup * 16 * TG;                           // first output row of this job's g block
    const int n0 = ngroup * 16 * TH;                           // first input column of its h block

    if (wave >= 4) {
        // ---- producer ------------------------------------------------------------------------------------------------------------
        const int pwave = wave - 4;
        int voff[8];
        const DwbItem it = dwb_item<TG, TH>(g, m0, n0, pwave, lane, kend, voff);
        float csum0 = 0.f, csum1 = 0.f;                         // running column sums of this work-item's g columns (db)
        // two register stages: chunk c + 1 waits in one to be split while chunk c + 2 is in flight into the other (a chunk is ~1 us
        // of matrix-core time, a loaded HBM round trip longer)
        DwbStage st[2];
        auto put = [&](int sidx, unsigned char* buf) {
            if (it.live) {
                if (it.is_g) {
                    const DwbStage& q = st[sidx];               // (pairwise: three dependent additions per column instead of eight)
                    csum0 = bf_add(csum0, bf_add(bf_add(bf_add(q.v[0].x, q.v[1].x), bf_add(q.v[2].x, q.v[3].x)),
                                                 bf_add(bf_add(q.v[4].x, q.v[5].x), bf_add(q.v[6].x, q.v[7].x))));
                    csum1 = bf_add(csum1, bf_add(bf_add(bf_add(q.v[0].y, q.v[1].y), bf_add(q.v[2].y, q.v[3].y)),
                                                 bf_add(bf_add(q.v[4].y, q.v[5].y), bf_add(q.v[6].y, q.v[7].y))));
                }
                dwb_split_store(st[sidx], buf + it.dst_off);
            }
        };
        // (unconditional: a wave without items has out-of-range offsets, a chunk beyond the slice a zero-sized descriptor -- neither
        // touches memory.  Under `if (live)` / inside the odd-chunk branch the loads were, to the compiler's wait-count pass, loads
        // that may or may not have been issued, and every split waited for ALL loads in flight: one chunk of look-ahead, not two.)
        auto load = [&](int sidx, int k0) { dwb_load(st[sidx], it, voff, k0); };
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
            if (PROF) { BF_VMCNT(8); DWB_TICK(1) }              // (this stage's operands are here)
            if (k0 + DWB_BK < kend) put(1, lds + DWB_BUF_BYTES);
            DWB_TICK(2)
            load(1, k0 + 3 * DWB_BK);
            DWB_TICK(3)
            DWB_BARRIER();
            DWB_TICK(4)
            const bool second = k0 + DWB_BK < kend;             // (an odd chunk count: the last iteration has no second half)
            if (PROF && second) { BF_VMCNT(8); DWB_TICK(1) }
            if (second && k0 + 2 * DWB_BK < kend) put(0, lds);
            DWB_TICK(2)
            load(0, k0 + 4 * DWB_BK);
            DWB_TICK(3)
            if (second) DWB_BARRIER();
            DWB_TICK(4)
        }
        // db: the four octets of a g column, in octet order (the operand buffers are free: the loop ended with a barrier)
        float* scr = reinterpret_cast<float*>(lds);            // [4 octets][16 TG columns]
        const bool want_db = g.colsum != nullptr && ngroup == 0;
        if (want_db && it.live && it.is_g) {
            const int local = pwave * 64 + lane, cp = local % (8 * TG), oct = local / (8 * TG);
            scr[oct * 16 * TG + 2 * cp] = csum0;
            scr[oct * 16 * TG + 2 * cp + 1] = csum1;
        }
        __syncthreads();
        const int ptid = tid - DWB_CONSUMERS;
        if (want_db && ptid < 16 * TG && m0 + ptid < g.M)
            g.colsum[(size_t)split * slab_stride + m0 + ptid] =
                ((scr[ptid] + scr[16 * TG + ptid]) + scr[32 * TG + ptid]) + scr[48 * TG + ptid];
        DWB_TICK(5)
        if (PROF && prof != nullptr && lane == 0) {
            prof[4 * wave] = pt[0]; prof[4 * wave + 1] = pt[1] + pt[2] + pt[3]; prof[4 * wave + 2] = pt[4]; prof[4 * wave + 3] = pt[5];
        }
        return;
    }

    // ---- consumer: output tiles rows 16 (ot0 + a), columns 16 (it0 + b) of the job's block -----------------------------------------
    const int ot0 = N_OT * (wave / WN), it0 = N_IT * (wave % WN);
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
        const unsigned char* hb = cur + (TG + it0) * 3 * BF_BLOCK;
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
            const int colc = n0 + 16 * (it0 + b) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * (ot0 + a) + 4 * q + r;
                if (row < g.M && colc < g.N) C[(size_t)row * g.ldc + colc] = acc[a][b][r];
            }
        }
    __syncthreads();                                           // (pairs with the producers' db barrier)
    DWB_TICK(4)
    if (PROF && prof != nullptr && lane == 0) {
        prof[4 * wave] = pt[0]; prof[4 * wave + 1] = pt[1] + pt[2]; prof[4 * wave + 2] = pt[3]; prof[4 * wave + 3] = pt[4];
        if (wave == 0) { prof[48] = wall0; prof[49] = wall_clock64(); }
    }
#undef DWB_TICK
}

template <bool PROF>
__device__ __forceinline__ void dw_bf_body(const DwbArgs& a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[DWB_LDS_BYTES];
    static_assert(DWB_LDS_BYTES >= ST_LDS_BYTES, "the tree update borrows the operand buffers as scratch");
    if ((int)blockIdx.x >= a.jobs) {
        if ((int)blockIdx.x == a.jobs && a.per.tree != nullptr) sumtree_update_body(a.per, lds);
        return;
    }
    // consecutive jobs = the output / input groups of one row slice, which read the same rows of g_l and h_l: keep them on one XCD's
    // L2 (the dispatcher places block b on XCD b % 8)
    const int job = xcd_remap((int)blockIdx.x, a.jobs);
    int q = 0;
    while (q + 1 < a.n && job >= a.p[q + 1].job_start) ++q;
    const DwbProblem& g = a.p[q];
    const int local = job - g.job_start;
    const int ng = local % g.ngroups, mg = (local / g.ngroups) % g.mgroups, split = local / (g.ngroups * g.mgroups);
    long long* prof = (PROF && a.prof != nullptr) ? a.prof + (size_t)job * DWB_PROF_SLOTS : nullptr;
    if (g.shape == 0) dwb_job<0, PROF>(g, mg, ng, split, a.rows, a.slab_stride, lds, prof);
    else if (g.shape == 1) dwb_job<1, PROF>(g, mg, ng, split, a.rows, a.slab_stride, lds, prof);
    else dwb_job<2, PROF>(g, mg, ng, split, a.rows, a.slab_stride, lds, prof);
}
__global__ __launch_bounds__(DWB_THREADS) void dw_bf_kernel(DwbArgs a) {
    kernarg_warm<sizeof(DwbArgs)>();       // (a block finds its problem by walking the table: dependent fetches of the argument block)
    dw_bf_body<false>(a);
}

}  // namespace morl
