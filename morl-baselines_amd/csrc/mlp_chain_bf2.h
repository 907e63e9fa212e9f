// The two ONLINE forward passes of an Envelope step as ONE chain (round 6): a workgroup carries a 64-row tile of the next-state pass AND
// the matching 64-row tile of the training pass -- they multiply the same weights, so every weight fragment a wave reads from the ring
// feeds FOUR MFMAs (two feature tiles x two row sub-tiles) instead of two, and the workgroup stages the weight stream once for 128 rows
// instead of once per 64: half the L2 -> LDS traffic, half the DMA issue slots, half the barriers and fragment reads per MFMA of
// mlp_chain_bf_kernel's forward launch (which runs the two passes as separate workgroups, two per CU, each staging the stream for itself).
// One workgroup per CU (one wave per SIMD: 128 accumulator + 192 activation registers per lane), every workgroup alike: nothing finishes
// early.  Arithmetic, slot order, epilogues, sign bits, saves and the arg-max stage are mlp_chain_bf.h's own (its functions are used as
// they are); sub-tile 0 is the no-grad next-state tile (MODE 0, arg-max at the end), sub-tile 1 the training tile (MODE 1).
// Replaces, together: QNet.forward of envelope.py:300 and :420 (common/networks.py:10-48).
#pragma once
#include "mlp_chain_bf.h"

namespace morl {

// product steps [P0, P1) of a tile pair for BOTH sub-tiles (see bf_six_part): per step the two tiles' fragments meet the two sub-tiles'
// activations -- four MFMAs, no two consecutive ones on one accumulator
template <int P0, int P1, bool READ, int R0 = 0, int R1 = 1, int R2 = 2, int R3 = 3, int R4 = 4, int R5 = 5, int DMA0 = -1>
__device__ __forceinline__ void bf2_six_part(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const bf_u32x4 (&w0)[3], const bf_u32x4 (&w1)[3],
                                             const bf_u32x4 (&xa)[3], const bf_u32x4 (&xb)[3], bf_u32x4 (&n0)[3], bf_u32x4 (&n1)[3],
                                             const unsigned char* next_base, const BfRing* ring = nullptr) {
    constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};
    constexpr int rd[6] = {R0, R1, R2, R3, R4, R5};
#pragma unroll
    for (int p = P0; p < P1; ++p) {
        if (READ) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (rd[k] == p) {
                    const int pl = 2 - (k >> 1);
                    if ((k & 1) == 0) n0[pl] = *reinterpret_cast<const bf_u32x4*>(next_base + pl * BF_BLOCK);
                    else n1[pl] = *reinterpret_cast<const bf_u32x4*>(next_base + (3 + pl) * BF_BLOCK);
                }
        }
        a0 = bf_mfma(w0[pw[p]], xa[px[p]], a0);
        a1 = bf_mfma(w1[pw[p]], xa[px[p]], a1);
        b0 = bf_mfma(w0[pw[p]], xb[px[p]], b0);
        b1 = bf_mfma(w1[pw[p]], xb[px[p]], b1);
        if (DMA0 >= 0) {
            if (p - P0 == 0) bf_ring_piece<4, DMA0>(*ring);
            if (p - P0 == 1) bf_ring_piece<4, DMA0 < 0 ? -1 : DMA0 + 1>(*ring);
            if (p - P0 == 2) bf_ring_piece<4, DMA0 < 0 ? -1 : DMA0 + 2>(*ring);
            if (p - P0 == 3) bf_ring_piece<4, DMA0 < 0 ? -1 : DMA0 + 3>(*ring);
            if (p - P0 == 4) bf_ring_piece<4, DMA0 < 0 ? -1 : DMA0 + 4>(*ring);
            if (p - P0 == 5) bf_ring_piece<4, DMA0 < 0 ? -1 : DMA0 + 5>(*ring);
        }
        BF_PIN();
    }
}
template <bool READ, int DMA0 = -1>
__device__ __forceinline__ void bf2_six_pair(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const bf_u32x4 (&w0)[3], const bf_u32x4 (&w1)[3],
                                             const bf_u32x4 (&xa)[3], const bf_u32x4 (&xb)[3], bf_u32x4 (&n0)[3], bf_u32x4 (&n1)[3],
                                             const unsigned char* next_base, const BfRing* ring = nullptr) {
    bf2_six_part<0, 6, READ, 0, 1, 2, 3, 4, 5, DMA0>(a0, a1, b0, b1, w0, w1, xa, xb, n0, n1, next_base, ring);
}

// bf_wide_step for two sub-tiles: the same stages, entries and fragment look-ahead; acc[t][T], x[t][s]
template <int KSTEPS, int EXTRA0>
__device__ __forceinline__ void bf2_wide_step(f32x4 (&acc)[2][16], const bf_u32x4 (&x)[2][8][3], BfRing& ring, int lane) {
    bf_u32x4 fa[2][3], fb[2][3];
    BF_PIN();
    const unsigned char* base = bf_stage_enter<4, EXTRA0>(ring) + lane * 16;
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {
        fa[0][pl] = *reinterpret_cast<const bf_u32x4*>(base + pl * BF_BLOCK);
        fa[1][pl] = *reinterpret_cast<const bf_u32x4*>(base + (3 + pl) * BF_BLOCK);
        BF_PIN();
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int T = 8 * hf;
            const bool last = (s == KSTEPS - 1) && (hf == 1);
#define BF2_ACC(i) acc[0][T + (i)], acc[0][T + (i) + 1], acc[1][T + (i)], acc[1][T + (i) + 1]
            if (s == 0 && hf == 0) {
                bf2_six_pair<true, 0>(BF2_ACC(0), fa[0], fa[1], x[0][s], x[1][s], fb[0], fb[1], base + 6 * BF_BLOCK, &ring);
                bf2_six_pair<true, 6>(BF2_ACC(2), fb[0], fb[1], x[0][s], x[1][s], fa[0], fa[1], base + 12 * BF_BLOCK, &ring);
            } else {
                bf2_six_pair<true, 4>(BF2_ACC(0), fa[0], fa[1], x[0][s], x[1][s], fb[0], fb[1], base + 6 * BF_BLOCK, &ring);
                bf2_six_pair<true, 10>(BF2_ACC(2), fb[0], fb[1], x[0][s], x[1][s], fa[0], fa[1], base + 12 * BF_BLOCK, &ring);
            }
            bf2_six_pair<true>(BF2_ACC(4), fa[0], fa[1], x[0][s], x[1][s], fb[0], fb[1], base + 18 * BF_BLOCK);
            if (!last) {
                bf2_six_part<0, 2, false>(BF2_ACC(6), fb[0], fb[1], x[0][s], x[1][s], fa[0], fa[1], base);
                base = ((s == 0 && hf == 0) ? bf_stage_enter<4, EXTRA0>(ring) : bf_stage_enter<4, 0>(ring)) + lane * 16;
                BF_PIN();
                bf2_six_part<2, 6, true, 2, 2, 3, 3, 4, 5, 0>(BF2_ACC(6), fb[0], fb[1], x[0][s], x[1][s], fa[0], fa[1], base, &ring);
            } else {
                bf2_six_pair<false>(BF2_ACC(6), fb[0], fb[1], x[0][s], x[1][s], fa[0], fa[1], base);
            }
#undef BF2_ACC
        }
    }
}

// the narrow last step for two sub-tiles (see bf_head_step)
template <int NT, int EXTRA0>
__device__ __forceinline__ void bf2_head_step(f32x4 (&acc)[2][2], const bf_u32x4 (&x)[2][8][3], BfRing& ring, int lane) {
    constexpr int KS_PER_STAGE = BF_STAGE_BLOCKS / (3 * NT);
    constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int s0 = 0; s0 < 8; s0 += KS_PER_STAGE) {
        BF_PIN();
        const unsigned char* base = bf_stage_begin<4, EXTRA0>(ring) + lane * 16;
        bf_u32x4 f[KS_PER_STAGE][NT][3];
#pragma unroll
        for (int ks = 0; ks < KS_PER_STAGE; ++ks)
#pragma unroll
            for (int t = 0; t < NT; ++t) bf_frag_load(f[ks][t], base, ks * NT + t);
        BF_PIN();
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int ks = 0; ks < KS_PER_STAGE; ++ks)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[0][t] = bf_mfma(f[ks][t][pw[p]], x[0][s0 + ks][px[p]], acc[0][t]);
                    acc[1][t] = bf_mfma(f[ks][t][pw[p]], x[1][s0 + ks][px[p]], acc[1][t]);
                }
    }
}

// input rows of one sub-tile (in_mode 0: cat(obs[b], weights[w]), row -> (b, w) by row_order) -> x[0 .. K0S)
template <int K0S>
__device__ __forceinline__ void bf2_input(const BfChain& p, int row, bool row_ok, int q, bf_u32x4 (&x)[8][3]) {
    int b = row, w = row;
    if (p.row_order == 0) { b = row / p.W; w = row - b * p.W; }
    else if (p.row_order == 1) { w = row / p.B; b = row - w * p.B; }
    const int K0 = p.D + p.R;
    const float* src_a = p.obs + (size_t)b * p.D;
    const float* src_w = p.weights + (size_t)w * p.R;
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * s + 8 * q + e;
            float a = 0.f;
            if (row_ok && k < K0) a = (k < p.D) ? src_a[k] : src_w[k - p.D];
            v[e] = a;
        }
        if (p.x0_out != nullptr && row_ok) {
            const int k = 32 * s + 8 * q;
            float* o = p.x0_out + (size_t)row * p.ldx0 + k;
            if (k < p.ldx0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);            // (ldx0 is a multiple of 4)
            if (k + 4 < p.ldx0) *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bf_split2(v[2 * u], v[2 * u + 1], hi[u], mid[u], lo[u]);
        x[s][0] = bf_u32x4{hi[0], hi[1], hi[2], hi[3]};
        x[s][1] = bf_u32x4{mid[0], mid[1], mid[2], mid[3]};
        x[s][2] = bf_u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
}

// ---- one 64-row next-state tile + one 64-row training tile through the chain -------------------------------------------------------------
// pn: the no-grad next-state chain (head -> the online slab; amax: arg-max of the tile's transitions), pc: the training chain (saves,
// sign bits, x0_out, head -> Q).  Same stream, same steps' shapes and biases (the online network).
template <int K0S>
__device__ __forceinline__ void bf2_chain_body(const BfChain& pn, const BfChain& pc, int row0, unsigned char* ring_lds, float* bias_lds,
                                               const float* am_w, int32_t* am_best, int32_t* am_pairs, int32_t* am_slot, int32_t* am_count,
                                               int am_epoch, int am_B, int am_W, int am_A, int am_R, int am_flags, long long* prof) {
#ifdef BF_PROF
    long long pt[6] = {0, 0, 0, 0, 0, 0}, tp_ = clock64();
    const long long tstart_ = tp_;
#endif
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int row = row0 + 16 * wave + m;
    const bool ok_n = row < pn.rows, ok_t = row < pc.rows;
    const size_t bits_idx = (size_t)((row0 >> 4) + wave) * 64 + lane;

    BfRing ring;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pc.stream, 0, pc.n_stages * BF_STAGE_BYTES, 0x00020000);
    ring.voff = wave * ((BF_STAGE_BLOCKS / 4) * BF_BLOCK) + lane * 16;
    ring.lds = ring_lds;
    ring.t = 0; ring.buf = 0; ring.n_stages = pc.n_stages; ring.wave = wave;
#ifdef BF_PROF
    ring.t_entry = 0; ring.t_dma = 0;
#endif
    bf_ring_issue<4>(ring, 0, 0);
    bf_ring_issue<4>(ring, 1, 1);
    float bv[BF_MAX_STEPS];
#pragma unroll
    for (int s = 0; s < BF_MAX_STEPS; ++s) {
        const float* bp = pc.step[s].bias;
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, (s < pc.n_steps && bp != nullptr) ? pc.step[s].N * 4 : 0, 0x00020000);
        bv[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, tid * 4, 0, 0));
    }
    bf_u32x4 x[2][8][3];
    bf2_input<K0S>(pn, row, ok_n, q, x[0]);
    bf2_input<K0S>(pc, row, ok_t, q, x[1]);
#pragma unroll
    for (int s = 0; s < BF_MAX_STEPS; ++s)
        if (s < pc.n_steps) bias_lds[s * BF_WIDE + tid] = bv[s];
    BF_VMCNT(0);
    __syncthreads();

    f32x4 acc[2][16];
    const int n_wide = pc.n_steps - 1;
    // ---- first step ---------------------------------------------------------------------------------------------------------------
    BF_T(0)
    bf_acc_init<16>(acc[0], bias_lds, q);
    bf_acc_init<16>(acc[1], bias_lds, q);
    bf2_wide_step<K0S, 0>(acc, x, ring, lane);
    BF_T(1)
    bf_wide_epilogue<0>(acc[0], x[0], pn.step[0], pn.rows, row, ok_n, bits_idx, q, ~0ull);
    bf_wide_epilogue<1>(acc[1], x[1], pc.step[0], pc.rows, row, ok_t, bits_idx, q, ~0ull);
    BF_T(2)
    // ---- the 256 x 256 steps --------------------------------------------------------------------------------------------------------
    for (int s = 1; s < n_wide; ++s) {
        bf_acc_init<16>(acc[0], bias_lds + s * BF_WIDE, q);
        bf_acc_init<16>(acc[1], bias_lds + s * BF_WIDE, q);
        bf2_wide_step<8, BF_SAVE_VMEM>(acc, x, ring, lane);
        BF_T(3)
        bf_wide_epilogue<0>(acc[0], x[0], pn.step[s], pn.rows, row, ok_n, bits_idx, q, ~0ull);
        bf_wide_epilogue<1>(acc[1], x[1], pc.step[s], pc.rows, row, ok_t, bits_idx, q, ~0ull);
        BF_T(4)
    }
    // ---- the head -----------------------------------------------------------------------------------------------------------------
    f32x4 hacc[2][2];
    {
        const BfStep& sn = pn.step[pn.n_steps - 1];
        const BfStep& st = pc.step[pc.n_steps - 1];
        bf_acc_init<2>(hacc[0], bias_lds + (pc.n_steps - 1) * BF_WIDE, q);
        bf_acc_init<2>(hacc[1], bias_lds + (pc.n_steps - 1) * BF_WIDE, q);
        if (st.N > 16) bf2_head_step<2, BF_SAVE_VMEM>(hacc, x, ring, lane);
        else bf2_head_step<1, BF_SAVE_VMEM>(hacc, x, ring, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const BfStep& so = t == 0 ? sn : st;
            const bool ok = t == 0 ? ok_n : ok_t;
            if (so.out != nullptr && ok) {
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * T + 4 * q + r;
                        if (n < so.N) so.out[(size_t)row * so.ldout + n] = hacc[t][T][r];
                        else if (n < so.ldout) so.out[(size_t)row * so.ldout + n] = 0.f;
                    }
            }
        }
    }
    BF_VMCNT(0);
    if (pn.amax) {
        // the next-state sub-tile is 1, 2 or 4 whole transitions (W = 64, 32, 16): their arg-max here, as in bf_chain_body
        __syncthreads();
        const int AR = pn.step[pn.n_steps - 1].N;
        const int wpt = am_W >> 4;
        const int sub = wave / wpt;
        float* region = reinterpret_cast<float*>(ring_lds) + sub * (am_W * 32 + am_W * MORL_MAX_OBJ + 2 * 4 * 64 + 3 * 64);
        EnvArgmaxLds L;
        L.qo = region;
        L.w = L.qo + am_W * 32;
        L.pv = L.w + am_W * MORL_MAX_OBJ;
        L.pc = reinterpret_cast<int*>(L.pv + 4 * 64);
        L.mark = L.pc + 4 * 64;
        L.slot = L.mark + 64;
        L.best = L.slot + 64;
        const int j = 16 * wave + m - sub * am_W;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * T + 4 * q + r;
                if (n < AR) L.qo[j * AR + n] = hacc[0][T][r];
            }
        const int b = row0 / am_W + sub;
#define BF2_AMAX(NWS) envelope_argmax_tile<NWS>(am_w, am_best, am_pairs, am_slot, am_count, am_epoch, am_B, am_W, am_A, am_R, am_flags & 1, 0, \
                                                (am_flags >> 1) & 1, (am_flags >> 2) & 1, b, L, sub)
        if (wpt == 4) BF2_AMAX(4);
        else if (wpt == 2) BF2_AMAX(2);
        else BF2_AMAX(1);
#undef BF2_AMAX
    }
    BF_T(5)
#ifdef BF_PROF
    if (prof != nullptr && lane == 0) {
        long long* o = prof + ((size_t)blockIdx.x * 4 + wave) * BF_PROF_SLOTS;
        for (int i = 0; i < 6; ++i) o[i] = pt[i];
        o[6] = ring.t_entry; o[7] = ring.t_dma; o[8] = tstart_; o[9] = tp_;
    }
#endif
}

// grid: one workgroup per pair of 64-row tiles (tile t of the next-state chain m.c[0] and tile t of the training chain m.c[1])
__global__ __launch_bounds__(256, 1) void mlp_chain_bf2_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    const int row0 = (int)blockIdx.x * BF_TM;
    float* bias_lds = reinterpret_cast<float*>(lds + BF_RING * BF_STAGE_BYTES);
#define BF2_AM m.td.weights, m.td.best_io, m.td.pairs_out, m.td.row_slot, m.td.count, m.td.epoch, m.td.B, m.td.W, m.td.A, m.td.R, \
               (m.td.diag_only | (m.td.fma_scal << 1) | (m.td.bmajor << 2)), m.prof
    if (m.c[1].k0_steps == 1) bf2_chain_body<1>(m.c[0], m.c[1], row0, lds, bias_lds, BF2_AM);
    else bf2_chain_body<2>(m.c[0], m.c[1], row0, lds, bias_lds, BF2_AM);
#undef BF2_AM
}

}  // namespace morl
