// Weight-gradient GEMMs of one backward pass, second generation (gfx950, wave64): one launch, split-K over the batch rows.
//
//   dW_l[o][i] = sum_rows g_l[row][o] * h_l[row][i]        db_l[o] = sum_rows g_l[row][o]
//
// What changed against gemm_grouped_tn_db_kernel (round 1: 0.46 of the fp32-MFMA peak, a quarter of it spent on padding):
//   * three WAVE LAYOUTS instead of one 128 x 128 tile for everything, so that the narrow problems stop multiplying
//     zeros: 2x2 waves of 64x64 (the 256 x 256 layers), 4x1 waves of 32x64 (in-dim <= 64: the first layer's 256 x 36),
//     1x4 waves of 32x32 (out-dim <= 32: the Q head's 18 x 256).  A layout with fewer MFMAs per contraction step gets
//     proportionally LONGER row slices, so every workgroup of the launch carries the same matrix-core work and the grid is
//     one balanced round of ~2 workgroups per CU.
//   * operand reads are 8-byte: the MFMA row slot i of a wave's two row tiles is mapped to the physical rows 2i, 2i+1
//     (same for columns), so one ds_read_b64 feeds both tiles -- half the LDS instructions per MFMA -- and the epilogue
//     stores 8-byte column pairs.
//   * the reads of contraction step j+1 are pinned ahead of the MFMAs of step j with sched_group_barrier (hipcc otherwise
//     sinks each ds_read next to its first use and waits on it: the exposed LDS latency was a tenth of the loop).
// Everything else as before: both operands are row-major with the contraction index as the slow axis, i.e. straight 16-byte
// copies into a K-major LDS image; double-buffered LDS, one barrier per 32-row chunk, next chunk's global loads in flight
// under the MFMAs; split-K partials go to slabs[split][P] in the flat parameter layout (fixed order => deterministic).
// Roofline: fp32 MFMA, 2 * rows * out * in flop per layer.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"
#include "replay_kernels.h"

namespace morl {


constexpr int DW2_BK = 32;
constexpr int DW2_THREADS = 256;
constexpr int DW2_LD = 132;        // floats per k-row of an LDS operand image (128 + 4: 16-byte rows, conflict-free b64 reads)

struct Dw2Problem {
    const float* G;   // [rows][ldg]  dLoss/dz_l          (A operand: out index contiguous)
    const float* H;   // [rows][ldh]  layer input          (B operand: in index contiguous)
    float* C;         // slab 0 of dW_l, row-major [M][ldc]
    float* colsum;    // slab 0 of db_l [M]
    int M, N;         // out, in
    int ldg, ldh, ldc;
    int layout;       // 0: 128 x 128 (2x2 waves of 64x64), 1: 128 x 64 (4x1 waves of 32x64), 2: 32 x 128 (1x4 waves of 32x32)
    int tiles_m, tiles_n;
    int k_per_split, splits;
    int job_start;    // first job (= workgroup) of this problem: job = job_start + split * tiles + tile
    int gcols, hcols;           // loadable columns of G / H: multiples of 4, pad columns are zeros
    int c_vec2;                 // 8-byte slab stores legal
};

struct Dw2Args {
    Dw2Problem p[MORL_MAX_LAYERS];
    int n, rows, jobs;
    long long slab_stride;     // floats between split slabs
    SumTreeUpdate per;         // per.tree != NULL: one extra workgroup (block `jobs`) applies the step's PER priority update
                               // (prioritized_buffer.py:187-195) -- it only depends on the TD kernel's priorities, nothing here
                               // depends on it, and the launch has a free slot: 16 us of serial tail disappear
    long long* prof;           // development probe only (tools/probes/dw_probe.hip): [jobs][6] cycle sums of the phases
    int stagger;               // s_sleep units (64 cycles) the second half of the grid waits before starting: the two workgroups of a
                               // CU (blocks b and b + grid/2) run identical chunk loops and would otherwise hit their barriers together
};

// global -> registers: a 32 x WIDTH chunk of a row-major operand (rows = contraction index), 16 bytes per load
template <int WIDTH>
struct Dw2Stage {
    float4 v[(DW2_BK * WIDTH / 4 + DW2_THREADS - 1) / DW2_THREADS];
};

// Branch-free: the operand is a buffer resource spanning rows [0, kend) of the matrix, so the rows of a ragged last chunk are
// zeroed by the hardware range check, and a lane whose columns do not exist carries an out-of-range offset.  (A branchy
// "vector or scalar tail" load made hipcc merge the two paths with moves of the loaded registers, i.e. wait for the loads
// right where they were issued -- the global latency of every chunk was exposed.)  Requires 16-byte aligned rows and a
// column count that is a multiple of 4 (pad columns of the matrices are zeros); the host falls back to the round-1 engine
// otherwise.
constexpr int DW2_OOB = 0x40000000;

template <int WIDTH>
__device__ __forceinline__ void dw2_load(Dw2Stage<WIDTH>& s, __amdgpu_buffer_rsrc_t rsrc, int ld, int col0, int cols, int k0) {
    constexpr int QPR = WIDTH / 4;                                   // float4 per k-row
    constexpr int N = (DW2_BK * QPR + DW2_THREADS - 1) / DW2_THREADS;
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int f = tid + q * DW2_THREADS;
        const int k = k0 + f / QPR, c = col0 + (f % QPR) * 4;
        const int off = (f < DW2_BK * QPR && c < cols) ? (k * ld + c) * 4 : DW2_OOB;
        const c2_u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        s.v[q] = __builtin_bit_cast(float4, raw);
    }
}

template <int WIDTH>
__device__ __forceinline__ void dw2_store(const Dw2Stage<WIDTH>& s, float* __restrict__ sm) {
    constexpr int QPR = WIDTH / 4;
    constexpr int N = (DW2_BK * QPR + DW2_THREADS - 1) / DW2_THREADS;
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int f = tid + q * DW2_THREADS;
        if (f < DW2_BK * QPR) *reinterpret_cast<float4*>(sm + (f / QPR) * DW2_LD + (f % QPR) * 4) = s.v[q];
    }
}

// piece q of the same store (the pieces of a stage are spread over the MFMA loop of the chunk before)
template <int WIDTH>
__device__ __forceinline__ void dw2_store_piece(const Dw2Stage<WIDTH>& s, float* __restrict__ sm, int q) {
    constexpr int QPR = WIDTH / 4;
    constexpr int N = (DW2_BK * QPR + DW2_THREADS - 1) / DW2_THREADS;
    if (q < N) {
        const int f = (int)threadIdx.x + q * DW2_THREADS;
        if (f < DW2_BK * QPR) *reinterpret_cast<float4*>(sm + (f / QPR) * DW2_LD + (f % QPR) * 4) = s.v[q];
    }
}

// LAYOUT -> tile shape and the waves' sub-tiles (TMW x TNW tiles of 32 x 32 per wave)
template <int LAYOUT> struct Dw2Shape;
template <> struct Dw2Shape<0> { static constexpr int BM = 128, BN = 128, TMW = 2, TNW = 2; };
template <> struct Dw2Shape<1> { static constexpr int BM = 128, BN = 64, TMW = 1, TNW = 2; };
template <> struct Dw2Shape<2> { static constexpr int BM = 32, BN = 128, TMW = 1, TNW = 1; };

template <int LAYOUT, bool PROF = false>
__device__ __forceinline__ void dw2_tile(const Dw2Problem& g, int tile_m, int tile_n, int split, int rows, long long slab_stride,
                                         float* sAbase, float* sBbase, long long* prof = nullptr) {
    long long pt[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    if (PROF) tprev = clock64();
#define DW2_TICK(q) if (PROF) { const long long t_ = clock64(); pt[q] += t_ - tprev; tprev = t_; }
    using S = Dw2Shape<LAYOUT>;
    constexpr int BM = S::BM, BN = S::BN, TMW = S::TMW, TNW = S::TNW;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * g.k_per_split;
    const int kend = min(rows, kbeg + g.k_per_split);
    const int lane = lane_id(), wave = wave_id();
    const int h = lane >> 5, i = lane & 31;
    // first row / column of this wave's sub-tile inside the workgroup tile
    const int wrow = (LAYOUT == 0) ? (wave >> 1) * 64 : (LAYOUT == 1) ? wave * 32 : 0;
    const int wcol = (LAYOUT == 0) ? (wave & 1) * 64 : (LAYOUT == 1) ? 0 : wave * 32;

    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b = 0; b < TNW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float colsum = 0.f;
    const bool do_colsum = (g.colsum != nullptr) && (tile_n == 0);

    // three-stage pipeline: chunk c is multiplied out of LDS while chunk c+1 waits in registers to be stored and chunk c+2 is
    // in flight from memory -- the loads of a chunk have a whole chunk of MFMAs plus a barrier to land (with one chunk of
    // look-ahead every workgroup waited ~2 000 cycles per chunk for them: 16 MB in flight chip-wide, ~3 us of loaded latency)
    Dw2Stage<BM> ra0, ra1;
    Dw2Stage<BN> rb0, rb1;
    constexpr int SBUF = DW2_BK * DW2_LD;      // floats per LDS buffer
    // rows [0, kend) of the operands: everything beyond this split's slice reads as zero
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)g.G, 0, kend * g.ldg * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)g.H, 0, kend * g.ldh * 4, 0x00020000);
    if (kbeg < kend) {
        dw2_load<BM>(ra0, rg, g.ldg, m0, g.gcols, kbeg);
        dw2_load<BN>(rb0, rh, g.ldh, n0, g.hcols, kbeg);
        dw2_load<BM>(ra1, rg, g.ldg, m0, g.gcols, kbeg + DW2_BK);
        dw2_load<BN>(rb1, rh, g.ldh, n0, g.hcols, kbeg + DW2_BK);
        dw2_store<BM>(ra0, sAbase);
        dw2_store<BN>(rb0, sBbase);
    }
    __syncthreads();
    DW2_TICK(4)                        // prologue: the first two chunks' loads, the first chunk staged
    float av[2][2], bv[2][2];          // [parity of the pair][tile]
// one 32-row chunk: BUF = LDS buffer holding it, LOADS = register stage that takes chunk + 2, STORES = stage holding chunk + 1
#define DW2_CHUNK(BUF, LA, LB, SA, SB, K0)                                                                           \
    {                                                                                                                \
        DW2_TICK(3)                                                                                                  \
        dw2_load<BM>(LA, rg, g.ldg, m0, g.gcols, (K0) + 2 * DW2_BK);                                                 \
        dw2_load<BN>(LB, rh, g.ldh, n0, g.hcols, (K0) + 2 * DW2_BK);                                                 \
        const float* sAc = sAbase + (BUF) * SBUF;                                                                    \
        const float* pa = sAc + h * DW2_LD + wrow + TMW * i;                                                         \
        const float* pb = sBbase + (BUF) * SBUF + h * DW2_LD + wcol + TNW * i;                                       \
        const bool st_ = (K0) + DW2_BK < kend;      /* chunk + 1 exists: its stage goes to the other buffer, piece by piece, */ \
        float* sAo = sAbase + ((BUF) ^ 1) * SBUF;   /* under this chunk's MFMAs (nobody reads that buffer now)              */ \
        float* sBo = sBbase + ((BUF) ^ 1) * SBUF;                                                                    \
        DW2_READ(0, 0)                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                           \
        _Pragma("unroll") for (int kk = 0; kk < DW2_BK; kk += 2) {                                                   \
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;                                                            \
            const int kn = (kk + 2 < DW2_BK) ? kk + 2 : kk;                                                          \
            DW2_READ(nxt, kn)                                                                                        \
            _Pragma("unroll") for (int a = 0; a < TMW; ++a)                                                          \
                _Pragma("unroll") for (int b = 0; b < TNW; ++b) acc[a][b] = mfma32(av[cur][a], bv[cur][b], acc[a][b]); \
            if (st_ && (kk & 2) == 0) {             /* one 16-byte piece of each operand every other contraction step */ \
                dw2_store_piece<BM>(SA, sAo, kk >> 2);                                                               \
                dw2_store_piece<BN>(SB, sBo, kk >> 2);                                                               \
            }                                                                                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, TMW * TNW, 0);                                               \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        DW2_TICK(0)                                                                                                  \
        if (do_colsum) {                                                                                             \
            constexpr int PARTS = DW2_THREADS / BM, KPP = DW2_BK / PARTS;                                            \
            const int crow = (int)threadIdx.x % BM, part = (int)threadIdx.x / BM;                                    \
            _Pragma("unroll") for (int kk = 0; kk < KPP; ++kk) colsum += sAc[(part * KPP + kk) * DW2_LD + crow];     \
        }                                                                                                            \
        DW2_TICK(1)                                                                                                  \
        __syncthreads();                                                                                             \
        DW2_TICK(2)                                                                                                  \
    }
#define DW2_READ(SLOT, KK)                                                                              \
    {                                                                                                   \
        if (TMW == 2) { const float2 t = *reinterpret_cast<const float2*>(pa + (KK) * DW2_LD); av[SLOT][0] = t.x; av[SLOT][1] = t.y; } \
        else av[SLOT][0] = pa[(KK) * DW2_LD];                                                           \
        if (TNW == 2) { const float2 t = *reinterpret_cast<const float2*>(pb + (KK) * DW2_LD); bv[SLOT][0] = t.x; bv[SLOT][1] = t.y; } \
        else bv[SLOT][0] = pb[(KK) * DW2_LD];                                                           \
    }
    // (bias sums: all 256 threads share the row sums of a chunk -- thread -> row tid % BM, k-slice tid / BM -- so that no wave
    // lags behind the others at the chunk's barrier)
    for (int k0 = kbeg; k0 < kend; k0 += 2 * DW2_BK) {
        DW2_CHUNK(0, ra0, rb0, ra1, rb1, k0)                         // chunk in buffer 0; chunk + 1 (stage 1) -> buffer 1
        if (k0 + DW2_BK < kend) DW2_CHUNK(1, ra1, rb1, ra0, rb0, k0 + DW2_BK)
    }
#undef DW2_READ
#undef DW2_CHUNK

    // epilogue: D register r of lane (ci, h) of tile (a, b) is physical row wrow + TMW * ((r & 3) + 8 * (r >> 2) + 4h) + a,
    // physical column wcol + TNW * ci + b
    float* __restrict__ C = g.C + (size_t)split * slab_stride;
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wrow + TMW * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
            if (row >= g.M) continue;
            const int col = n0 + wcol + TNW * i;
            float* dst = C + (size_t)row * g.ldc + col;
            if (TNW == 2) {
                if (g.c_vec2 == 2 && col + 1 < g.N) {
                    // non-temporal: the slab is written once and read once by the reduction; keep it from filling the L2s with
                    // dirty lines that the end of the launch has to flush (A/B: MORL_DW_NT)
                    typedef float dw2_v2f __attribute__((ext_vector_type(2)));
                    dw2_v2f v2;
                    v2.x = acc[a][0][r]; v2.y = acc[a][TNW - 1][r];
                    __builtin_nontemporal_store(v2, reinterpret_cast<dw2_v2f*>(dst));
                } else if (g.c_vec2 && col + 1 < g.N) *reinterpret_cast<float2*>(dst) = make_float2(acc[a][0][r], acc[a][TNW - 1][r]);
                else {
                    if (col < g.N) dst[0] = acc[a][0][r];
                    if (col + 1 < g.N) dst[1] = acc[a][TNW - 1][r];
                }
            } else if (col < g.N) dst[0] = acc[a][0][r];
        }
    if (do_colsum) {
        // combine the k-slices in slice order (the operand buffers are free: the loop ended with a barrier)
        constexpr int PARTS = DW2_THREADS / BM;
        float* scr = sAbase;
        scr[threadIdx.x] = colsum;
        __syncthreads();
        if ((int)threadIdx.x < BM && m0 + (int)threadIdx.x < g.M) {
            float t = scr[threadIdx.x];
#pragma unroll
            for (int q = 1; q < PARTS; ++q) t += scr[q * BM + (int)threadIdx.x];
            g.colsum[(size_t)split * slab_stride + m0 + (int)threadIdx.x] = t;
        }
    }
    DW2_TICK(5)                        // epilogue: slab tile and bias sums to HBM
    if (PROF && prof != nullptr && threadIdx.x == 0)
        for (int q = 0; q < 6; ++q) prof[q] = pt[q];
#undef DW2_TICK
}

template <bool PROF>
__device__ __forceinline__ void dw_tiles_body(const Dw2Args& a) {
    __shared__ __attribute__((aligned(16))) float sA[2][DW2_BK * DW2_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][DW2_BK * DW2_LD];
    // consecutive jobs = the tiles of one row slice, which share that slice of g_l / h_l: keep them on one XCD's L2
    static_assert(sizeof(sA) >= ST_LDS_BYTES, "the tree update borrows the A-operand buffers as scratch");
    if ((int)blockIdx.x >= a.jobs) {
        if ((int)blockIdx.x == a.jobs && a.per.tree != nullptr) sumtree_update_body(a.per, &sA[0][0]);
        return;
    }
    const int job = xcd_remap((int)blockIdx.x, a.jobs);
    if (a.stagger > 0 && (int)blockIdx.x >= (a.jobs >> 1))
        for (int k = 0; k < a.stagger; k += 64) __builtin_amdgcn_s_sleep(64);
    int q = 0;
    while (q + 1 < a.n && job >= a.p[q + 1].job_start) ++q;
    const Dw2Problem& g = a.p[q];
    const int local = job - g.job_start;
    const int tiles = g.tiles_m * g.tiles_n;
    const int split = local / tiles, tile = local % tiles;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    if (g.layout == 0) dw2_tile<0, PROF>(g, tm, tn, split, a.rows, a.slab_stride, &sA[0][0], &sB[0][0], PROF && a.prof ? a.prof + 6 * job : nullptr);
    else if (g.layout == 1) dw2_tile<1, PROF>(g, tm, tn, split, a.rows, a.slab_stride, &sA[0][0], &sB[0][0], PROF && a.prof ? a.prof + 6 * job : nullptr);
    else dw2_tile<2, PROF>(g, tm, tn, split, a.rows, a.slab_stride, &sA[0][0], &sB[0][0], PROF && a.prof ? a.prof + 6 * job : nullptr);
}

__global__ __launch_bounds__(DW2_THREADS, 2) void dw_tiles_kernel(Dw2Args a) { dw_tiles_body<false>(a); }

// ----------------------------------------------------------------------------------------------------------------------------
// grads[p] = sum_{s < splits(p)} slabs[s][p] for slabs whose split count differs per parameter range (the layouts above), plus
// the sum-of-squares partials and the loss exactly like grad_reduce_kernel (optim_kernels.h).
// ----------------------------------------------------------------------------------------------------------------------------
struct Dw2Ranges {
    long long end[2 * MORL_MAX_LAYERS];   // parameter ranges [end[r-1], end[r]) in flat order (W_0, b_0, W_1, ...)
    int splits[2 * MORL_MAX_LAYERS];
    int n;
};


// same contract as grad_reduce_kernel; `rg` gives the split count of every parameter range
static __global__ __launch_bounds__(256) void grad_reduce_ranges_kernel(const float* __restrict__ slabs, Dw2Ranges rg,
                                                                        long long slab_stride, float* __restrict__ grads,
                                                                        long long P, double* __restrict__ sumsq_part,
                                                                        const double* __restrict__ loss_part, int n_loss,
                                                                        double inv_mse, double inv_aux, float lambda,
                                                                        float* __restrict__ loss_out) {
    __shared__ double s_red[256 / 64];
    double ss = 0.0;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        int r = 0;
        while (r + 1 < rg.n && p >= rg.end[r]) ++r;
        const int splits = rg.splits[r];
        // 8 independent loads in flight per step (the adds stay in slab order -> same bits run to run)
        float g = slabs[p];
        int s = 1;
        for (; s + 8 <= splits; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(s + u) * slab_stride + p];
#pragma unroll
            for (int u = 0; u < 8; ++u) g += v[u];
        }
        for (; s < splits; ++s) g += slabs[(size_t)s * slab_stride + p];
        grads[p] = g;
        ss += (double)g * (double)g;
    }
    ss = wave_sum(ss);
    if (lane_id() == 0) s_red[wave_id()] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 256 / 64; ++w) t += s_red[w];
        sumsq_part[blockIdx.x] = t;
    }
    if (blockIdx.x == 0 && loss_out != nullptr && wave_id() == 0) {
        double a = 0.0, c = 0.0;
        for (int e = lane_id(); e < n_loss; e += kWave) { a += loss_part[2 * e]; c += loss_part[2 * e + 1]; }
        a = wave_sum(a);
        c = wave_sum(c);
        if (lane_id() == 0) {
            const float mse = (float)(a * inv_mse);
            float loss = mse;
            if (lambda > 0.f) loss = (1.0f - lambda) * mse + lambda * (float)(c * inv_aux);
            *loss_out = loss;
        }
    }
}

}  // namespace morl
