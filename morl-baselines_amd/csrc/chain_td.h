// The envelope arg-max + TD target + dLoss/dQ of a gradient step (envelope.py:298-313, :422-439) as the INPUT STAGE of the
// backward chain (mlp_chain2.h, ChainArgs::in_mode == 2): the workgroup that carries a tile of TD rows through the backward
// layers computes those rows' dLoss/dQ itself instead of reading them from a separate launch (envelope_td_kernel: 11-13 us of the
// 0.36 ms step for 2.4 MB of input, latency-bound, plus a launch boundary in front of the backward pass).  Same arithmetic, same
// roundings, same first-maximum tie-break as envelope_kernels.h -- results are bit-identical to the separate kernel's
// (tests/test_chain_tilings.py runs the parity tests under both).
// MEASURED (MI355X, round 3, profiles/r03_td_fused_ab.json): break-even -- the stage adds 8.6 us to the backward launch (2.4 fixed,
// ~2.5 candidate scan, 3 target gather / TD, 1.2 row write-out: both co-resident workgroups of a CU sit in it at the same time, so
// nothing hides it) and 2 us to the loss reduction, and removes a 10 us kernel + one launch boundary: 0.3630 vs 0.3614 ms per
// step.  It is therefore OFF by default (MORL_TD_FUSED=1 turns it on) and kept tested; what made it and the separate kernel 6 us
// faster than the first version was instantiating the candidate scan per objective count (a run-time R turned the inner loop
// into an eight-way predicated one).
//
// Tile rows are TD rows in the engines' internal order r = b * WI + i (the rows of a transition are contiguous: the training
// forward assembles them that way, row_order 0), so a tile needs the Qo slabs of ceil(TM / WI) + 1 transitions at most -- 4.6 KB
// at W = 64.  They are staged in the tile's own LDS buffer (free until the chain starts) with one round of coalesced loads;
// then lane <-> TD row, the 256 / TM slices of the workgroup split the (j, a) candidates and read each candidate's R values
// from LDS (the rows of one transition read the same address: a broadcast), and the per-slice winners of a row are merged
// through LDS in candidate order, the first maximum winning ties -- the arg-max form of envelope_td_kernel, which the A/B of
// DESIGN.md section 4 found faster than a wave butterfly.  Everything after the arg-max (target gather, TD error, homotopy
// term, loss partials, PER priority, dLoss/dQ) runs one thread per (row, objective).
#pragma once
#include "mlp_chain.h"
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

struct ChainTd {
    const float* qo;        // [B][W][A][R] next-state slabs of the online net (or the all-gathered layout, see part_floats)
    const float* qt;        // ... of the target net
    const float* weights;   // [WI][R] the scalarisation vectors of this launch's TD rows
    const float* q_main;    // [rows][ldq] Q_online(s_b, w_i), row i * B + b
    const int32_t* actions; // [B]
    const float* rewards;   // [B][R]
    const float* dones;     // [B]
    float* dq;              // [rows][ldq] dLoss/dQ, also needed in HBM (the head's weight gradient reads it)
    double* loss_part;      // [ceil(rows / 16)][2]: sum td^2, sum (wQ - wTQ)^2 per 16-row block (a tile writes all of its blocks)
    float* priority;        // [B] |td . w| of the i == 0 rows, or NULL
    float* priority_clear;  // [B] zeroed instead (a shard that does not own weight 0), or NULL
    float* target;          // [rows][R] or NULL (parity outputs)
    int32_t* pref;          // [rows] or NULL
    int32_t* ac;            // [rows] or NULL
    int B, W, A, R, ldq;    // W = candidates' weight count (all gathered weights of a sharded job)
    int WI;                 // scalarisation vectors (TD rows per transition) of this launch; row = b * WI + i
    int i_offset;           // global index of weights[0] (diag_only: candidate j == global i)
    int diag_only;          // DDQN target: only the A actions of slab j = i
    float gamma, c_mse, c_aux;
    int part_floats;        // 0: [B][W][A][R]; > 0: all-gathered layout, see EnvelopeTdArgs
    long long part_stride;
};

constexpr int CTD_MAX_ROWS = 64;     // rows of the largest tile

// LDS scratch of the stage: declared ONCE by the kernel and handed down (a __shared__ array inside this template would be
// instantiated per tile size and weight stream -- six copies, 33 KB, and the kernel would drop to one workgroup per CU)
struct ChainTdScratch {
    double l[CTD_MAX_ROWS][2];                 // per-row loss terms
    float g[CTD_MAX_ROWS][MORL_MAX_OBJ];       // dLoss/dQ of the taken action
    float pv[CH_THREADS];                      // per-(slice, row) partial maxima ...
    int pc[CH_THREADS];                        // ... and their candidate indices
    int bc[CTD_MAX_ROWS];                      // flattened (j*, a*) of every tile row
    int act[CTD_MAX_ROWS];
};

// can the slabs of every transition a tile touches be staged at once in `tile_floats` floats of LDS?  (the host falls back to
// envelope_td_kernel in front of the backward pass otherwise: the weak-scaled job's 512-weight slabs)
inline bool ctd_slab_ok(long long slab_floats, int WI, int tile_floats) {
    return slab_floats >= 1 && WI >= 1 && ((long long)(CTD_MAX_ROWS / WI) + 2) * slab_floats <= tile_floats;
}

// candidates [q_lo, q_hi) of one row in index order: scal = w . Q with every product and sum rounded separately (objective
// order), first maximum wins
template <int RT>
__device__ __forceinline__ void ctd_scan(const float* q0, const float (&wi)[MORL_MAX_OBJ], int q_lo, int q_hi, int c_off,
                                         float& best, int& best_c) {
    int cc = q_lo;
    for (; cc + 4 <= q_hi; cc += 4) {                  // 4 candidates per step: their LDS reads overlap
        float sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* q = q0 + (size_t)(cc + u) * RT;
            float sc = __fmul_rn(wi[0], q[0]);
#pragma unroll
            for (int r = 1; r < RT; ++r) sc = __fadd_rn(sc, __fmul_rn(wi[r], q[r]));
            sv[u] = sc;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (sv[u] > best || best_c == 0x7fffffff) { best = sv[u]; best_c = c_off + cc + u; }
    }
    for (; cc < q_hi; ++cc) {
        const float* q = q0 + (size_t)cc * RT;
        float sc = __fmul_rn(wi[0], q[0]);
#pragma unroll
        for (int r = 1; r < RT; ++r) sc = __fadd_rn(sc, __fmul_rn(wi[r], q[r]));
        if (sc > best || best_c == 0x7fffffff) { best = sc; best_c = c_off + cc; }
    }
}

// One tile: rows [row0, row0 + TM) of `rows`.  On return sAct[m][0 .. K0pad) holds the tile's dLoss/dQ rows (zero padded) -- the
// input of the backward chain -- and the HBM copy / loss partials / priorities / parity outputs are written.  256 threads.
template <int TM>
__device__ __forceinline__ void chain_td_stage(const ChainTd& t, int row0, int rows, float* sAct, int ld_act, int K0pad,
                                               ChainTdScratch* scr) {
    int* s_bc = scr->bc;
    int* s_act = scr->act;
    float (*s_g)[MORL_MAX_OBJ] = scr->g;
    double (*s_l)[2] = scr->l;
    float* s_pv = scr->pv;
    int* s_pc = scr->pc;
    const int tid = (int)threadIdx.x, lane = lane_id();
    const int A = t.A, R = t.R, W = t.W, B = t.B, WI = t.WI;
    const int slab = W * A * R;
    const int pf = t.part_floats > 0 ? t.part_floats : slab;
    const int n_c = t.diag_only ? A : W * A;
    const int n_rows = min(TM, rows - row0);           // live rows of this tile (>= 1)
    const int b_lo = row0 / WI, b_hi = (row0 + n_rows - 1) / WI;
    // (row, objective) work-items of the last phase: their taken actions are fetched now, under the slab loads
    constexpr int NP = (TM + 31) / 32;                 // passes of 32 rows
    int p_act[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int m = k * 32 + (tid >> 3);
        p_act[k] = t.actions[(row0 + (m < n_rows ? m : 0)) / WI];
    }
    // ---- the slabs of the tile's transitions -> LDS: every load of a thread is issued before its first LDS store -------------
    {
        const int total = (b_hi - b_lo + 1) * slab;
        if (((slab | pf) & 3) == 0 && (t.part_stride & 3) == 0 && (((uintptr_t)t.qo) & 15u) == 0) {     // 16-byte pieces
            const int total4 = total >> 2, slab4 = slab >> 2, pf4 = pf >> 2;
            for (int base = tid; base < total4; base += 4 * CH_THREADS) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * CH_THREADS;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx < total4) {
                        const int tt = idx / slab4, e = idx - tt * slab4, g = e / pf4;
                        v[u] = *reinterpret_cast<const float4*>(t.qo + (size_t)g * (size_t)t.part_stride +
                                                                ((size_t)(b_lo + tt) * pf4 + (size_t)(e - g * pf4)) * 4);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * CH_THREADS;
                    if (idx < total4) *reinterpret_cast<float4*>(sAct + (size_t)idx * 4) = v[u];
                }
            }
        } else {
            for (int base = tid; base < total; base += 8 * CH_THREADS) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * CH_THREADS;
                    v[u] = 0.f;
                    if (idx < total) {
                        const int tt = idx / slab, e = idx - tt * slab, g = e / pf;
                        v[u] = t.qo[(size_t)g * (size_t)t.part_stride + (size_t)(b_lo + tt) * pf + (size_t)(e - g * pf)];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * CH_THREADS;
                    if (idx < total) sAct[idx] = v[u];
                }
            }
        }
    }
    // what the last phase needs besides the arg-max: issued here, used after the candidate scan
    float p_qv[NP][MORL_MAX_OBJ], p_rw[NP][MORL_MAX_OBJ], p_ndg[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int m = k * 32 + (tid >> 3);
        const int row = row0 + (m < n_rows ? m : 0);
        const int b = row / WI;
        p_ndg[k] = __fmul_rn(__fsub_rn(1.0f, t.dones[b]), t.gamma);                                 // (1 - d) * gamma
#pragma unroll
        for (int r = 0; r < MORL_MAX_OBJ; ++r) {
            p_qv[k][r] = (r < R) ? t.q_main[(size_t)row * t.ldq + p_act[k] * R + r] : 0.f;
            p_rw[k][r] = (r < R) ? t.rewards[(size_t)b * R + r] : 0.f;
        }
    }
    __syncthreads();
    // ---- arg-max: lane <-> row, the 256 / TM slices split the candidates -------------------------------------------------------
    {
        constexpr int S = CH_THREADS / TM;
        const int m = tid & (TM - 1), sl = tid / TM;
        const bool live = m < n_rows;
        const int row = row0 + (live ? m : 0);
        const int b = row / WI, i = row - b * WI;
        float wi[MORL_MAX_OBJ];
#pragma unroll
        for (int r = 0; r < MORL_MAX_OBJ; ++r) wi[r] = (r < R) ? t.weights[(size_t)i * R + r] : 0.f;
        const float* q0 = sAct + (size_t)(b - b_lo) * slab + (t.diag_only ? (size_t)(i + t.i_offset) * A * R : 0);
        const int c_off = t.diag_only ? (i + t.i_offset) * A : 0;
        const int q_lo = (int)(((long long)n_c * sl) / S), q_hi = (int)(((long long)n_c * (sl + 1)) / S);
        float best = -INFINITY;
        int best_c = 0x7fffffff;
        // (one straight-line instantiation per objective count: with R a run-time bound the eight-way predicated inner loop was
        // 8.6 of the stage's 17 us at 384 candidates per row)
        switch (R) {
            case 1: ctd_scan<1>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
            case 2: ctd_scan<2>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
            case 3: ctd_scan<3>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
            case 4: ctd_scan<4>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
            case 5: ctd_scan<5>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
            case 6: ctd_scan<6>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
            case 7: ctd_scan<7>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
            default: ctd_scan<8>(q0, wi, q_lo, q_hi, c_off, best, best_c); break;
        }
        s_pv[sl * TM + m] = best;
        s_pc[sl * TM + m] = best_c;
        __syncthreads();
        if (tid < TM) {
            // merge the slices in candidate order: strictly greater replaces, so the first maximum wins
            float bv = s_pv[tid];
            int bc = s_pc[tid];
            for (int q = 1; q < S; ++q) {
                const float v = s_pv[q * TM + tid];
                const int c2 = s_pc[q * TM + tid];
                if (c2 != 0x7fffffff && (bc == 0x7fffffff || v > bv)) { bv = v; bc = c2; }
            }
            s_bc[tid] = bc;
        }
        __syncthreads();
    }
    (void)lane;
    // ---- TD error, loss terms, dLoss/dQ: one thread per (row, objective), 32 rows per pass.  Every thread walks all R objectives
    // of its row (a handful of L2 hits) instead of exchanging them with its neighbours: no cross-lane traffic -----------------------
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int m = k * 32 + (tid >> 3), r = tid & (MORL_MAX_OBJ - 1);
        const bool row_ok = m < n_rows;
        const int row = row0 + (row_ok ? m : 0);
        const int b = row / WI, i = row - b * WI;
        const bool on = row_ok && r < R;
        const int bc = s_bc[row_ok ? m : 0];
        const int act = p_act[k];
        float tgt_r = 0.f, td_r = 0.f, wi_r = 0.f, wq = 0.f, wtq = 0.f, pr = 0.f;
        double msq = 0.0;
#pragma unroll
        for (int q = 0; q < MORL_MAX_OBJ; ++q) {
            if (q < R) {
                const int e = bc * R + q;
                const int g = e / pf;
                const float tgt = t.qt[(size_t)g * (size_t)t.part_stride + (size_t)b * pf + (size_t)(e - g * pf)];
                const float wi = t.weights[(size_t)i * R + q];
                const float qv = p_qv[k][q];
                const float tq = __fadd_rn(p_rw[k][q], __fmul_rn(p_ndg[k], tgt));
                const float td = __fsub_rn(qv, tq);
                // w . Q, w . TQ and w . td: products summed in objective order, every operation rounded separately
                wq = (q == 0) ? __fmul_rn(qv, wi) : __fadd_rn(wq, __fmul_rn(qv, wi));
                wtq = (q == 0) ? __fmul_rn(tq, wi) : __fadd_rn(wtq, __fmul_rn(tq, wi));
                pr = (q == 0) ? __fmul_rn(td, wi) : __fadd_rn(pr, __fmul_rn(td, wi));
                msq += (double)td * (double)td;
                if (q == r) { tgt_r = tgt; td_r = td; wi_r = wi; }
            }
        }
        const float daux = __fsub_rn(wq, wtq);
        if (on) {
            s_g[m][r] = t.c_mse * td_r + t.c_aux * daux * wi_r;
            if (t.target) t.target[((size_t)i * B + b) * R + r] = tgt_r;      // parity outputs: reference row order i * B + b
        }
        if (row_ok && r == 0) {
            s_act[m] = act;
            s_l[m][0] = msq;
            s_l[m][1] = (double)daux * (double)daux;
            if (t.pref) t.pref[(size_t)i * B + b] = bc / A;
            if (t.ac) t.ac[(size_t)i * B + b] = bc % A;
            if (i == 0 && t.priority) t.priority[b] = fabsf(pr);
            if (i == 0 && t.priority_clear) t.priority_clear[b] = 0.f;
        }
    }
    __syncthreads();
    // ---- the tile's input rows: sAct[m][0 .. K0pad) and the HBM copy dq[row][0 .. ldq) ---------------------------------------
    for (int e = tid; e < TM * K0pad; e += CH_THREADS) {
        const int m = e / K0pad, c = e - m * K0pad;
        float v = 0.f;
        if (m < n_rows) {
            const int r = c - s_act[m] * R;
            if (r >= 0 && r < R) v = s_g[m][r];
            if (c < t.ldq) t.dq[(size_t)(row0 + m) * t.ldq + c] = v;
        }
        sAct[(size_t)m * ld_act + c] = v;
    }
    // loss partials of the tile's 16-row blocks, rows in order (fixed summation order)
    if (tid < TM / 16) {
        double a = 0.0, c2 = 0.0;
        for (int m = tid * 16; m < min(n_rows, tid * 16 + 16); ++m) { a += s_l[m][0]; c2 += s_l[m][1]; }
        t.loss_part[(size_t)(row0 / 16 + tid) * 2 + 0] = a;
        t.loss_part[(size_t)(row0 / 16 + tid) * 2 + 1] = c2;
    }
}

}  // namespace morl
