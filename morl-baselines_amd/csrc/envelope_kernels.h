// Envelope-Q specific kernels: network-input assembly, the envelope arg-max + TD epilogue.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

// ----------------------------------------------------------------------------------------------
// X0[row][0:D] = obs[b][:], X0[row][D:D+R] = weights[k][:], zero padding up to ldx.
// row_order 2: row r = (obs[r], weights[r]);  0: row = b*W + k (next-state slab order); 1: row = k*B + b (reference TD-row order,
// envelope.py:284-291).  Replaces th.cat((obs, w)) of QNet.forward (envelope.py:75) together with the
// repeat / repeat_interleave tiling (envelope.py:284-291, 416-418) -- the tiled batch is never built.
// HBM-bound: writes rows*ldx*4 bytes, reads are L2 hits.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_input_kernel(const float* __restrict__ obs,
                                                          const float* __restrict__ weights, float* __restrict__ x0,
                                                          int B, int W, int D, int R, int ldx, int row_order) {
    const long long total = (long long)B * W * ldx;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(e / ldx), c = (int)(e % ldx);
        int b, k;
        if (row_order == 0) { b = row / W; k = row % W; }
        else if (row_order == 1) { k = row / B; b = row % B; }
        else { b = row; k = row; }                                  // 2: paired rows, weights [B][R] (W = 1)
        float v = 0.f;
        if (c < D) v = obs[(size_t)b * D + c];
        else if (c < D + R) v = weights[(size_t)k * R + (c - D)];
        x0[e] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// Envelope arg-max + TD target + loss gradient, one workgroup per transition b (4 waves).
//
//   scal(i; j, a) = w_i . Qo[b][j][a][:]   -- products and sums separately rounded, objective order
//                                            (bit-identical to th.einsum("br,bwar->bwa"), envelope.py:422)
//   (j*, a*) = first arg-max over the flattened (j, a) index  (== max over a then arg-max over j with
//              torch's first-max tie-break, envelope.py:424-426)
//   target[i,b,:] = Qt[b][j*][a*][:]                                      (envelope.py:429-439)
//   tq = r_b + ((1 - done_b) * gamma) * target                             (envelope.py:298)
//   td = Q[i,b,action_b,:] - tq ; dQ = dLoss/dQ for MSE (+ homotopy term)  (envelope.py:300-313)
//
// The Qo[b] slab (W*A*R floats) and the weight vectors are staged in LDS once and shared by the W rows of this
// transition.  Lane <-> TD row i; the waves of the workgroup split the (j, a) candidates and read each candidate's R values
// as LDS broadcasts, so the arg-max loop has no cross-lane traffic at all; the per-wave partial winners of a row are merged
// through LDS in candidate order, lowest index winning ties.  The shuffle form north_star names (lanes <-> candidates, a wave
// butterfly carrying (value, index)) was built and A/B-measured on MI355X in round 3 (profiles/r03_argmax_ab.json): 16.6 us against
// 13.0 us for this form at W = 64 x A = 6, 52.5 against 51.5 us at the weak-scaled W = 512 -- with 64 rows per transition the lanes
// are better spent on rows; bit-identical indices, so the slower form was removed in round 4.
// diag_only restricts j to i (DDQN target, envelope.py:442-463).
// HBM-bound and tiny: reads 2*B*W*A*R*4 bytes once.
// ----------------------------------------------------------------------------------------------
constexpr int ENV_MAX_SLAB = 9216;   // floats of LDS for one Qo[b] / Qt[b] slab (W*A*R)
constexpr int ENV_MAX_WR = 1536;     // floats of LDS for the weight vectors (W*R)

struct EnvelopeTdArgs {
    const float* qo;        // [B][W][A][R]
    const float* qt;        // [B][W][A][R]
    const float* weights;   // [W][R]
    const float* row_weights;  // optional [B][R]: generic mode, ONE scalarisation vector per row b (W_i = 1);
                               // reproduces Envelope.envelope_target(obs, w, sampled_w) for arbitrary rows
    const float* q_main;    // [W*B][ldq]  Q_online(s_b, w_i), row i*B+b ; may be NULL (reduce only)
    const int32_t* actions; // [B]
    const float* rewards;   // [B][R]
    const float* dones;     // [B]
    float* target;          // [W*B][R] or NULL
    int32_t* pref;          // [W*B] or NULL
    int32_t* ac;            // [W*B] or NULL
    float* dq;              // [W*B][ldq] gradient wrt Q (zero outside the taken action) or NULL
    double* loss_part;      // [B][2]  sum td^2, sum (wQ - wTQ)^2 over this transition's W rows, or NULL
    float* priority;        // [B] |td . w| of the i = 0 row, or NULL
    float* priority_clear;  // [B] zeroed instead (a shard that does not own weight 0: the ranks' priorities are summed), or NULL
    int B, W, A, R, ldq;
    int i_groups;           // the W rows of a transition are split over this many workgroups (grid = B * i_groups)
    int WI;                 // number of scalarisation vectors (TD rows per transition) in `weights`; 0 -> W.  A rank of a
                            // weight-sharded job owns WI = W/G of them while the slabs still hold all W candidates
    int i_offset;           // global index of weights[0] (only matters for diag_only: candidate j == global i)
    int diag_only;
    float gamma;
    float c_mse;            // (1 - lambda) * 2 / (W*B*R)
    float c_aux;            // lambda * 2 / (W*B)
    int fma_scal;           // 1: scalarise with an fma chain  fma(w_r, q_r, ...fma(w_1, q_1, w_0 * q_0))  -- what torch's unbatched
                            // einsum("r,bar->ba") of Envelope.max_action (envelope.py:389-402) evaluates to -- instead of
                            // separately rounded products and sums (the batched einsum of the TD target)
    // Lazy target evaluation (morl_envelope_update's default on the layer-fused engines): a TD row only ever reads the target
    // network at ITS arg-max (j*, a*), and the rows of a transition agree on a handful of j* (1 546 distinct (b, j*) pairs of
    // 16 384 at the flagship shape), so the target network is evaluated AFTER the arg-max, on the distinct pairs only.
    //   phase 1  arg-max only: best_io[row] = flattened (j*, a*); the distinct (b, j*) of the workgroup's rows get compact target
    //            rows and every TD row learns its own (qt, q_main unused)
    //   phase 2  TD only:      best_io read back; the target vector of TD row `row` is qt + (row_slot[row] * A + a*) * R
    //   phase 0  both in one launch from full slabs (the weight-sharded step, the per-layer engine, parity outputs)
    int phase;
    int32_t* best_io;       // [rows] internal row order (see bmajor)
    // phase 1: the rows of a workgroup find their distinct j* through LDS (the lowest TD row selecting a weight owns it) and one
    // lane takes that many compact rows from the step's counter -- ONE returning global atomic per workgroup pass.  (A version
    // that also deduplicated through global memory, an atomicExch on a per-pair epoch tag, cost the launch 12.3 us instead of
    // 5.7: two dependent device-scope round trips per workgroup.)  The ORDER of the compact rows varies from run to run; the
    // values do not -- every row of the 16-row chain is computed independently of its position.  Workgroups that share a
    // transition (i_groups > 1) may each list the same pair: evaluated twice, same value.  count[epoch & 1] is this step's
    // counter, the other one is zeroed for the next step; nothing else needs clearing
    int32_t* pairs_out;     // [rows] compact row -> b * W + j
    int32_t* row_slot;      // [rows] TD row (internal order) -> compact row (written by phase 1, read by phase 2)
    int32_t* count;         // [2]
    int epoch;              // >= 1
    float* zero_ptr;        // optional: zero_ptr[k] = 0 for k in [0, zero_n) outside [keep_lo, keep_hi) -- the batch-sharded step's
    int zero_n, keep_lo, keep_hi;   // "the other ranks' priorities are zeros" (one memset launch less per rank step)
    int bmajor;             // internal row order of q_main / dq: 0 = row i * B + b (reference order, envelope.py:284-291),
                            // 1 = row b * WI + i (what the layer-fused engines use: the rows of a transition are contiguous, so a
                            // backward tile needs one or two slabs, chain_td.h).  target / pref / ac stay in reference order
    int part_floats;        // 0: qo / qt are [B][W][A][R].  > 0: all-gathered layout, the slab of transition b is made of
    long long part_stride;  //    W*A*R / part_floats pieces of part_floats floats, piece g at  g * part_stride + b * part_floats
};

// LDS-resident throughout: phase 1 stages Qo[b], Qt[b], the weight vectors and the taken-action Q entries of this
// transition with coalesced / parallel loads; phase 3 writes all outputs in bulk.
// Phase 2 mapping: lane <-> scalarisation vector i (TD row), wave q of the block's NW (4..16) <-> a 1/NW slice of the
// (j, a) candidates (NW grows with the candidate count: a weight-sharded job reduces over all gathered W * A).
// Every lane walks its wave's candidates in index order reading the candidate's R values as LDS *broadcasts* (all lanes
// read the same address: one conflict-free access), so the arg-max needs no cross-lane traffic at all; the NW
// per-wave partial winners of a row are merged through LDS in candidate order (first maximum wins, like th.max /
// th.argmax).  The kernel is latency-bound, not bandwidth-bound: what matters is that no lane waits on a dependent
// shuffle or global load inside the candidate loop.
constexpr int ENV_MAX_WAVES = 16;

// candidates [q_lo, q_hi) of one TD row in index order (q0: the row's first candidate): scal = w . Q, products and sums rounded
// separately in objective order (FMA: the fma chain of Envelope.max_action's unbatched einsum); first maximum wins
template <int RT, bool FMA>
__device__ __forceinline__ void env_scan(const float* q0, const float (&wi)[MORL_MAX_OBJ], int q_lo, int q_hi, int c_off,
                                         float& best, int& best_c) {
    int c = q_lo;
    for (; c + 4 <= q_hi; c += 4) {                    // 4 candidates per step: their LDS reads overlap
        float sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* q = q0 + (size_t)(c + u) * RT;   // envelope: wave-uniform address -> LDS broadcast
            float s = __fmul_rn(wi[0], q[0]);
#pragma unroll
            for (int r = 1; r < RT; ++r) s = FMA ? fmaf(q[r], wi[r], s) : __fadd_rn(s, __fmul_rn(wi[r], q[r]));
            sv[u] = s;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (sv[u] > best || best_c == 0x7fffffff) { best = sv[u]; best_c = c_off + c + u; }
    }
    for (; c < q_hi; ++c) {
        const float* q = q0 + (size_t)c * RT;
        float s = __fmul_rn(wi[0], q[0]);
#pragma unroll
        for (int r = 1; r < RT; ++r) s = FMA ? fmaf(q[r], wi[r], s) : __fadd_rn(s, __fmul_rn(wi[r], q[r]));
        if (s > best || best_c == 0x7fffffff) { best = s; best_c = c_off + c; }
    }
}

// The arg-max stage (phase 1 of envelope_td_kernel) of ONE transition for a caller whose workgroup already holds the transition's
// online slab in LDS -- the no-grad pass of the split-bf16 forward chain when its row tile IS the transition's W rows
// (mlp_chain_bf.h): the launch of envelope_td_kernel<1> (10 us of latencies between the forward passes and the target rows)
// becomes ~1 500 cycles at the end of workgroups that are resident anyway.  Same mapping -- lane <-> TD row i (W <= 64), the
// caller's NW waves <-> slices of the (j, a) candidates --, same env_scan, same merge order, same compaction: best_io / row_slot
// are bit-identical to the separate launch's, pairs_out up to the order of the compact rows (which varies from run to run there
// too).  Every work-item of the workgroup calls it; W scalarisation vectors, i_groups = 1, the slab in [W][A][R] order.  A row tile
// may hold several whole transitions (64 rows = two transitions of 32 weights ...): the workgroup's waves then form `sub`-groups of NW
// waves, one per transition, each with its own LDS regions; all of them run this code side by side (the barriers are the workgroup's).
struct EnvArgmaxLds {
    float* qo;      // [W * A * R]   filled by the caller (no barrier needed before the call)
    float* w;       // [W * R]
    float* pv;      // [NW][64]
    int* pc;        // [NW][64]
    int* mark;      // [W]
    int* slot;      // [W]
    int* best;      // [W]
};
// (the arguments as scalars, not as the EnvelopeTdArgs they come from: handed on as a struct out of another kernel's argument
// block, hipcc copied that whole block to scratch memory -- 1.2 KB per work-item, both chain kernels 1.8 x slower)
struct EnvArgmaxArgs {
    const float* weights;
    int32_t* best_io;
    int32_t* pairs_out;
    int32_t* row_slot;
    int32_t* count;
    int epoch, B, W, A, R, diag_only, i_offset, fma_scal, bmajor;
};
template <int NW>
__device__ __forceinline__ void envelope_argmax_tile(const float* weights, int32_t* best_io, int32_t* pairs_out, int32_t* row_slot,
                                                     int32_t* count, int epoch, int nB, int W, int A, int R, int diag_only,
                                                     int i_offset, int fma_scal, int bmajor, int b, const EnvArgmaxLds L, int sub = 0) {
    const int tid = (int)threadIdx.x - sub * 64 * NW, lane = tid & 63, wave = tid >> 6;     // (inside this transition's wave group)
    for (int e = tid; e < W; e += 64 * NW) L.mark[e] = 0x7fffffff;
    for (int e = tid; e < W * R; e += 64 * NW) L.w[e] = weights[e];
    __syncthreads();
    const int i = lane;
    const bool live = i < W && b < nB;          // (a tile's last transitions may lie beyond the batch: they only keep the barriers company)
    float wi[MORL_MAX_OBJ];
#pragma unroll
    for (int r = 0; r < MORL_MAX_OBJ; ++r) wi[r] = (live && r < R) ? L.w[i * R + r] : 0.f;
    const int n_c = diag_only ? A : W * A;
    const int c_off = (diag_only && live) ? (i + i_offset) * A : 0;
    const int q_lo = (int)(((long long)n_c * wave) / NW), q_hi = (int)(((long long)n_c * (wave + 1)) / NW);
    float best = -INFINITY;
    int best_c = 0x7fffffff;
    const float* qb = L.qo + (size_t)c_off * R;
    switch (fma_scal ? R + MORL_MAX_OBJ : R) {
#define MORL_TD_CASE(r) case r: env_scan<r, false>(qb, wi, q_lo, q_hi, c_off, best, best_c); break; \
                        case r + MORL_MAX_OBJ: env_scan<r, true>(qb, wi, q_lo, q_hi, c_off, best, best_c); break;
        MORL_TD_CASE(1) MORL_TD_CASE(2) MORL_TD_CASE(3) MORL_TD_CASE(4)
        MORL_TD_CASE(5) MORL_TD_CASE(6) MORL_TD_CASE(7) MORL_TD_CASE(8)
#undef MORL_TD_CASE
        default: break;
    }
    if (live) { L.pv[wave * 64 + i] = best; L.pc[wave * 64 + i] = best_c; }
    __syncthreads();
    int jsel = 0;
    if (wave == 0 && live) {
        // merge the slices in candidate order: strictly greater replaces, so the first maximum wins
        float bv = L.pv[lane];
        int bc = L.pc[lane];
        for (int q = 1; q < NW; ++q) {
            const float v = L.pv[q * 64 + lane];
            const int cc = L.pc[q * 64 + lane];
            if (cc != 0x7fffffff && (bc == 0x7fffffff || v > bv)) { bv = v; bc = cc; }
        }
        L.best[i] = bc;
        best_io[bmajor ? (size_t)b * W + i : (size_t)i * nB + b] = bc;
        // distinct j* of the transition: the lowest TD row selecting a weight owns it, the owners take consecutive compact rows
        jsel = bc / A;
        atomicMin(&L.mark[jsel], i);
    }
    __syncthreads();
    if (wave == 0) {
        const bool first = live && L.mark[jsel] == i;
        const unsigned long long won = __ballot(first);
        if (won != 0ull) {
            const int leader = __ffsll(won) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(count + (epoch & 1), __popcll(won));
            base = __shfl(base, leader);
            if (first) {
                const int k = base + __popcll(won & ((1ull << lane) - 1ull));
                pairs_out[k] = b * W + jsel;
                L.slot[jsel] = k;
            }
        }
    }
    __syncthreads();
    if (wave == 0 && live) row_slot[bmajor ? (size_t)b * W + i : (size_t)i * nB + b] = L.slot[jsel];
    if (b == 0 && tid == 0) count[(epoch + 1) & 1] = 0;
}

// PHASE: EnvelopeTdArgs::phase as a compile-time constant -- the arg-max-only and TD-only launches of a lazily evaluated step do
// not carry each other's code, registers and LDS (nor does the one-launch form: with the three in one body it ran 10.7 us instead
// of 9.4 at the flagship shape).  (Also measured: the LDS sized per launch instead of for the largest slab -- 9 KB instead of
// 119 KB, every workgroup resident at once instead of one per CU in two rounds -- was 1 us SLOWER per launch; the kernel is
// bound by each workgroup's own chain of latencies, not by residency.)
template <int PHASE>
__global__ __launch_bounds__(64 * ENV_MAX_WAVES) void envelope_td_kernel(EnvelopeTdArgs p) {
    __shared__ float s_qo[PHASE != 2 ? ENV_MAX_SLAB : 4];
    __shared__ float s_qt[PHASE == 0 ? ENV_MAX_SLAB : 4];
    __shared__ float s_w[ENV_MAX_WR];
    __shared__ float s_qm[ENV_MAX_WR];    // Q_online(s_b, w_i)[action_b][r]
    __shared__ float s_tgt[ENV_MAX_WR];   // selected target vectors
    __shared__ float s_g[ENV_MAX_WR];     // dLoss/dQ of the taken action
    __shared__ int s_best[ENV_MAX_WR];    // flattened (j*, a*)
    __shared__ int s_mark[PHASE == 1 ? ENV_MAX_WR : 4];        // [W] (W * R <= ENV_MAX_WR): lowest TD row of this workgroup that selected weight j
    __shared__ int s_slot[PHASE == 1 ? ENV_MAX_WR : 4];        // [W]: compact target row of (b, j)
    __shared__ float s_pv[ENV_MAX_WAVES][kWave];   // per-wave partial maxima of the 64 rows in flight ...
    __shared__ int s_pc[ENV_MAX_WAVES][kWave];     // ... and their candidate indices
    __shared__ double s_red[4][2];
    const int nw = (int)(blockDim.x >> 6);
    const int ig_n = p.i_groups > 0 ? p.i_groups : 1;
    const int b = (int)blockIdx.x / ig_n, ig = (int)blockIdx.x % ig_n;
    const int lane = lane_id(), wave = wave_id();
    const int W = p.W, A = p.A, R = p.R;
    const int slab = W * A * R;
    const bool generic = p.row_weights != nullptr;
    const int nI = generic ? 1 : (p.WI > 0 ? p.WI : W);   // scalarisation vectors (TD rows) of this transition
    const int per_g = (nI + ig_n - 1) / ig_n;
    const int i_lo = min(nI, ig * per_g), i_hi = min(nI, i_lo + per_g);
    const bool train = p.q_main != nullptr;
    const int act = train ? p.actions[b] : 0;
    const int pf = p.part_floats > 0 ? p.part_floats : slab;
    if (PHASE == 1)
        for (int e = (int)threadIdx.x; e < W; e += (int)blockDim.x) s_mark[e] = 0x7fffffff;
    if (PHASE != 2)
        for (int e = (int)threadIdx.x; e < slab; e += (int)blockDim.x) {
            const int g = e / pf;
            const size_t off = (size_t)g * (size_t)p.part_stride + (size_t)b * pf + (size_t)(e - g * pf);
            s_qo[e] = p.qo[off];
            if (PHASE == 0) s_qt[e] = p.qt[off];
        }
    if (PHASE == 2)       // (best (j*, a*), compact target row) -> target vector: two dependent loads, one TD row per thread,
        for (int i = i_lo + (int)threadIdx.x; i < i_hi; i += (int)blockDim.x) {      // under the staging loads below
            const size_t irow = p.bmajor ? (size_t)b * nI + i : (size_t)i * p.B + b;
            const int bc = p.best_io[irow];
            const float* qt = p.qt + ((size_t)p.row_slot[irow] * A + (bc % A)) * R;
            s_best[i] = bc;
            for (int r = 0; r < R; ++r) s_tgt[i * R + r] = qt[r];
        }
    for (int e = (int)threadIdx.x; e < nI * R; e += (int)blockDim.x) {
        s_w[e] = generic ? p.row_weights[(size_t)b * R + e] : p.weights[e];
        if (train) {
            const size_t qrow = p.bmajor ? (size_t)b * nI + (e / R) : (size_t)(e / R) * p.B + b;
            s_qm[e] = p.q_main[qrow * p.ldq + act * R + (e % R)];
        }
    }
    __syncthreads();

    const float not_done_gamma = train ? __fmul_rn(__fsub_rn(1.0f, p.dones[b]), p.gamma) : 0.f;  // (1 - d) * gamma
    float rew[MORL_MAX_OBJ];
#pragma unroll
    for (int r = 0; r < MORL_MAX_OBJ; ++r) rew[r] = (train && r < R) ? p.rewards[(size_t)b * R + r] : 0.f;
    double acc_mse = 0.0, acc_aux = 0.0;

    // rows of this workgroup in batches of 64 (one per lane)
    for (int ib = i_lo; ib < i_hi; ib += kWave) {
        const int i = ib + lane;
        const bool live = i < i_hi;
        float wi[MORL_MAX_OBJ];
#pragma unroll
        for (int r = 0; r < MORL_MAX_OBJ; ++r) wi[r] = (live && r < R) ? s_w[i * R + r] : 0.f;
        const int n_c = p.diag_only ? A : W * A;
        if (PHASE == 2) {
            // the arg-max was taken by an earlier launch (phase 1)
        } else {
        // candidate range of this wave: a quarter of (j, a) in index order (DDQN: of the A actions of slab j = i)
        const int c_off = (p.diag_only && live) ? (i + p.i_offset) * A : 0;     // per-lane base in DDQN mode
        const int q_lo = (int)(((long long)n_c * wave) / nw), q_hi = (int)(((long long)n_c * (wave + 1)) / nw);
        float best = -INFINITY;
        int best_c = 0x7fffffff;
        // one straight-line instantiation per objective count and scalarisation form (a run-time R made the inner loop an
        // eight-way predicated one: most of this kernel's time)
        const float* qb = s_qo + (size_t)c_off * R;
        switch (p.fma_scal ? R + MORL_MAX_OBJ : R) {
#define MORL_TD_CASE(r) case r: env_scan<r, false>(qb, wi, q_lo, q_hi, c_off, best, best_c); break; \
                        case r + MORL_MAX_OBJ: env_scan<r, true>(qb, wi, q_lo, q_hi, c_off, best, best_c); break;
            MORL_TD_CASE(1) MORL_TD_CASE(2) MORL_TD_CASE(3) MORL_TD_CASE(4)
            MORL_TD_CASE(5) MORL_TD_CASE(6) MORL_TD_CASE(7) MORL_TD_CASE(8)
#undef MORL_TD_CASE
            default: break;
        }
        if (live) { s_pv[wave][i - ib] = best; s_pc[wave][i - ib] = best_c; }
        __syncthreads();
        }
        if (wave == 0 && live) {
            const size_t irow = p.bmajor ? (size_t)b * nI + i : (size_t)i * p.B + b;     // internal row of (i, b)
            int bc;
            if (PHASE == 2) {
                bc = s_best[i];
            } else {
                // merge the slices in candidate order: strictly greater replaces, so the first maximum wins
                float bv = s_pv[0][lane];
                bc = s_pc[0][lane];
                for (int q = 1; q < nw; ++q) {
                    const float v = s_pv[q][lane];
                    const int cc = s_pc[q][lane];
                    if (cc != 0x7fffffff && (bc == 0x7fffffff || v > bv)) { bv = v; bc = cc; }
                }
            }
            s_best[i] = bc;
            if (PHASE == 1) {
                p.best_io[irow] = bc;
            } else {
            const float* qt = (PHASE == 2) ? s_tgt + (size_t)i * R : s_qt + (size_t)bc * R;
            float td[MORL_MAX_OBJ];
            float wq = 0.f, wtq = 0.f;
#pragma unroll
            for (int r = 0; r < MORL_MAX_OBJ; ++r) {
                td[r] = 0.f;
                if (r < R) {
                    s_tgt[i * R + r] = qt[r];
                    if (train) {
                        const float tq = __fadd_rn(rew[r], __fmul_rn(not_done_gamma, qt[r]));
                        const float qv = s_qm[i * R + r];
                        td[r] = __fsub_rn(qv, tq);
                        wq = (r == 0) ? __fmul_rn(qv, wi[0]) : __fadd_rn(wq, __fmul_rn(qv, wi[r]));
                        wtq = (r == 0) ? __fmul_rn(tq, wi[0]) : __fadd_rn(wtq, __fmul_rn(tq, wi[r]));
                    }
                }
            }
            if (train) {
                const float daux = __fsub_rn(wq, wtq);
                double m = 0.0;
#pragma unroll
                for (int r = 0; r < MORL_MAX_OBJ; ++r)
                    if (r < R) {
                        s_g[i * R + r] = p.c_mse * td[r] + p.c_aux * daux * wi[r];
                        m += (double)td[r] * (double)td[r];
                    }
                acc_mse += m;
                acc_aux += (double)daux * (double)daux;
                if (i == 0 && p.priority) {
                    float pr = __fmul_rn(td[0], wi[0]);
#pragma unroll
                    for (int r = 1; r < MORL_MAX_OBJ; ++r)
                        if (r < R) pr = __fadd_rn(pr, __fmul_rn(td[r], wi[r]));
                    p.priority[b] = fabsf(pr);
                }
            }
            }
        }
        if (PHASE == 1) {
            // distinct j* of this pass: the lowest TD row selecting a weight owns it (s_mark), the owners take consecutive compact rows
            const int jsel = (wave == 0 && live) ? s_best[i] / A : 0;
            if (wave == 0 && live) atomicMin(&s_mark[jsel], i);
            __syncthreads();
            if (wave == 0) {
                const bool first = live && s_mark[jsel] == i;
                const unsigned long long won = __ballot(first);
                if (won != 0ull) {
                    const int leader = __ffsll(won) - 1;
                    int base = 0;
                    if (lane == leader) base = atomicAdd(p.count + (p.epoch & 1), __popcll(won));
                    base = __shfl(base, leader);
                    if (first) {
                        const int k = base + __popcll(won & ((1ull << lane) - 1ull));
                        p.pairs_out[k] = b * W + jsel;
                        s_slot[jsel] = k;
                    }
                }
            }
            __syncthreads();
            if (wave == 0 && live) p.row_slot[p.bmajor ? (size_t)b * nI + i : (size_t)i * p.B + b] = s_slot[jsel];
        }
        __syncthreads();
    }
    if (PHASE == 1) {                  // (uniform: the arg-max launch writes nothing else)
        if (blockIdx.x == 0 && threadIdx.x == 0) p.count[(p.epoch + 1) & 1] = 0;
        return;
    }
    // loss partials: only wave 0 accumulated; butterfly sum over its lanes (fixed order)
    if (wave == 0) {
        acc_mse = wave_sum(acc_mse);
        acc_aux = wave_sum(acc_aux);
        if (lane == 0) { s_red[0][0] = acc_mse; s_red[0][1] = acc_aux; }
    }
    __syncthreads();

    // phase 3: bulk outputs.  Output row of (i, b) is i*B + b (generic mode: b).
    const int nB = p.B;
    const int nMine = i_hi - i_lo;
    for (int e = (int)threadIdx.x; e < nMine * R; e += (int)blockDim.x) {
        const int i = i_lo + e / R;
        const size_t row = generic ? (size_t)b : (size_t)i * nB + b;
        if (p.target) p.target[row * R + (e % R)] = s_tgt[i * R + (e % R)];
    }
    for (int i = i_lo + (int)threadIdx.x; i < i_hi; i += (int)blockDim.x) {
        const size_t row = generic ? (size_t)b : (size_t)i * nB + b;
        if (p.pref) p.pref[row] = s_best[i] / A;
        if (p.ac) p.ac[row] = s_best[i] % A;
    }
    if (train && p.dq) {
        for (int e = (int)threadIdx.x; e < nMine * p.ldq; e += (int)blockDim.x) {
            const int i = i_lo + e / p.ldq, c = e % p.ldq;
            const int r = c - act * R;
            const size_t drow = p.bmajor ? (size_t)b * nI + i : (size_t)i * nB + b;
            p.dq[drow * p.ldq + c] = (r >= 0 && r < R) ? s_g[i * R + r] : 0.f;
        }
    }
    if (p.priority_clear && ig == 0 && threadIdx.x == 0) p.priority_clear[b] = 0.f;
    if (p.zero_ptr)
        for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < p.zero_n; k += (int)(gridDim.x * blockDim.x))
            if (k < p.keep_lo || k >= p.keep_hi) p.zero_ptr[k] = 0.f;
    if (p.loss_part && threadIdx.x == 0) {
        p.loss_part[(size_t)blockIdx.x * 2 + 0] = s_red[0][0];
        p.loss_part[(size_t)blockIdx.x * 2 + 1] = s_red[0][1];
    }
}

}  // namespace morl
