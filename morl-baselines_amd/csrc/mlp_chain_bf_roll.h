// Rolling epilogues for the 256 x 256 steps of the split-bf16 chain (round 6; included from the middle of mlp_chain_bf.h, whose
// fragments, ring and product steps it uses).
//
// mlp_chain_bf.h walks a 256 x 256 step k-step after k-step: all sixteen feature tiles stay open until the last k-step, then the wave
// runs the step's whole epilogue (64 values per lane: mask or ReLU, fp32 copy, three-way split) with the matrix pipe idle -- measured
// (BF_PROF, profiles/r06_chain_phase_cycles.txt) 17 400 of the backward chain's 87 400 cycles per wave, and with one wave per SIMD (the
// backward launch: 16 384 rows / 1 024 SIMDs) nothing else runs under them.  Here a step is walked PAIR of feature tiles after pair: the
// weight stream holds, per step, stage (j, h) = k-steps 4h .. 4h + 3 of tiles 2j, 2j + 1 (BfSplitJob::pair_major), a pair's two
// accumulators are finished after its two stages, and its epilogue -- 8 values per lane, exactly operand k-step j of the next step --
// is issued in slices BETWEEN the MFMAs of the pair that follows (six vector instructions per product step, under the two MFMAs' 32
// cycles).  Every accumulator still sees its products in the order (k-step, product) of mlp_chain_bf.h: the same bits.
// Registers: the step's operand (96) AND the next one (96) are live, two accumulator pairs instead of sixteen tiles -- about 300 per
// lane, i.e. one wave per SIMD: the form is the backward launch's (and of any launch with one 64-row tile per CU).
#pragma once

namespace morl {

// where a step's fp32 copies go + its mask word: what an epilogue slice needs of the step the finished pair belongs to
struct BfRollOut {
    float* out;                    // the step's output rows and their extent in bytes (0: no copy): plain uniform scalars -- the buffer
    int range;                     //   descriptor is made at the store (as a struct member it was carried in vector registers, spilled,
                                   //   and every store ran a readfirstlane loop behind s_waitcnt vmcnt(0))
    int voff;                      // this lane's row: (row * ldout + 4 q) * 4, or CH_OOB
    int keep_lo, keep_hi;          // MODE 2: the lane's mask word
};
struct BfRollPend {
    float v[8];
    unsigned hi[4], mid[4], lo[4];
};

__device__ __forceinline__ BfRollOut bf_roll_out(const BfStep& st, int rows, int row, bool row_ok, int q, unsigned long long keep) {
    BfRollOut o;
    o.out = st.out;
    o.range = st.out != nullptr ? rows * st.ldout * 4 : 0;
    o.voff = row_ok ? (row * st.ldout + 4 * q) * 4 : CH_OOB;
    o.keep_lo = (int)(unsigned)keep;
    o.keep_hi = (int)(unsigned)(keep >> 32);
    return o;
}

// slice `slot` (0 .. 10) of the epilogue of finished pair P (tiles 2P, 2P + 1: accumulators d0, d1) -> operand k-step `out` of the next
// step.  MODE 2 (backward): mask, fp32 copy, split -- bf_wide_epilogue<2>'s arithmetic for s = P, value for value.
// slots 0 - 3: two values each; 4, 5: the two row stores; 6 - 9: one split pair each; 10: the operand registers
template <int MODE>
__device__ __forceinline__ void bf_roll_slot(int slot, int P, BfRollPend& e, const BfRollOut& o, const f32x4& d0, const f32x4& d1,
                                             bf_u32x4 (&out)[3]) {
    static_assert(MODE == 2, "rolling epilogues: the backward chain");
    if (slot >= 0 && slot < 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ee = 2 * slot + i, r = ee & 3, k = 8 * P + ee;
            float a = (ee >> 2) ? d1[r] : d0[r];
            const int m = ((k < 32 ? o.keep_lo : o.keep_hi) << (31 - (k & 31))) >> 31;
            a = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & m);
            e.v[ee] = a;
        }
    } else if (slot == 4) {
        __builtin_amdgcn_raw_buffer_store_b128(bf_u32x4{__builtin_bit_cast(unsigned, e.v[0]), __builtin_bit_cast(unsigned, e.v[1]),
                                                        __builtin_bit_cast(unsigned, e.v[2]), __builtin_bit_cast(unsigned, e.v[3])},
                                               __builtin_amdgcn_make_buffer_rsrc((void*)o.out, 0, o.range, 0x00020000), o.voff, 128 * P, 0);
    } else if (slot == 5) {
        __builtin_amdgcn_raw_buffer_store_b128(bf_u32x4{__builtin_bit_cast(unsigned, e.v[4]), __builtin_bit_cast(unsigned, e.v[5]),
                                                        __builtin_bit_cast(unsigned, e.v[6]), __builtin_bit_cast(unsigned, e.v[7])},
                                               __builtin_amdgcn_make_buffer_rsrc((void*)o.out, 0, o.range, 0x00020000), o.voff + 64, 128 * P, 0);
    } else if (slot >= 6 && slot < 10) {
        const int u = slot - 6;
        bf_split2(e.v[2 * u], e.v[2 * u + 1], e.hi[u], e.mid[u], e.lo[u]);
    } else if (slot == 10) {
        out[0] = bf_u32x4{e.hi[0], e.hi[1], e.hi[2], e.hi[3]};
        out[1] = bf_u32x4{e.mid[0], e.mid[1], e.mid[2], e.mid[3]};
        out[2] = bf_u32x4{e.lo[0], e.lo[1], e.lo[2], e.lo[3]};
    }
}

// the whole epilogue of a pair in one go (the chain's last pair: nothing follows to hide it under)
template <int MODE>
__device__ __forceinline__ void bf_roll_pair_epilogue(int P, BfRollPend& e, const BfRollOut& o, const f32x4& d0, const f32x4& d1,
                                                      bf_u32x4 (&out)[3]) {
#pragma unroll
    for (int slot = 0; slot <= 10; ++slot) bf_roll_slot<MODE>(slot, P, e, o, d0, d1, out);
}

// bf_six_part with an epilogue slice behind every product step's MFMAs (slot0 + (p - P0); hook_on false: bf_six_part itself)
template <int MODE, int P0, int P1, bool READ, int R0, int R1, int R2, int R3, int R4, int R5, int DMA0>
__device__ __forceinline__ void bf_roll_six_part(f32x4& c0, f32x4& c1, const bf_u32x4 (&w0)[3], const bf_u32x4 (&w1)[3],
                                                 const bf_u32x4 (&x)[3], bf_u32x4 (&n0)[3], bf_u32x4 (&n1)[3],
                                                 const unsigned char* next_base, const BfRing* ring, bool hook_on, int slot0, int P,
                                                 BfRollPend& e, const BfRollOut& o, const f32x4& d0, const f32x4& d1, bf_u32x4 (&out)[3]) {
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
        c0 = bf_mfma(w0[pw[p]], x[px[p]], c0);
        c1 = bf_mfma(w1[pw[p]], x[px[p]], c1);
        if (hook_on) bf_roll_slot<MODE>(slot0 + (p - P0), P, e, o, d0, d1, out);
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

// One 256 x 256 step, pair after pair (four waves).  x: the step's operand; y[j]: operand k-step j of the NEXT step, written as pair j's
// epilogue completes (j = 0 .. 6: during pair j + 1); the LAST pair's accumulators stay in cc[1] for the caller -- the next step's
// first pair runs its epilogue (PENDING_IN: pair 7 of the step before, described by `prev`, lands in x[7], which this step first reads
// in its second stage), or the caller does after the chain's last step.
// Vector-memory bookkeeping of the counted waits (bf_stage_enter's EXTRA: instructions this lane issued behind the weight group waited
// for, other than the group that follows it): a pair's two stores sit in the FIRST stage of the pair after it, so from the step's
// fourth entry on every entry sees two; the third is waited for without them (it sees two only when PENDING_IN); the first two see what
// the caller says (E01: the step-0 epilogue's sixteen stores + this step's mask word, or a rolling step's 2 + 1).
template <int MODE, bool PENDING_IN, int E01>
__device__ __forceinline__ void bf_roll_wide_step(f32x4 (&cc)[2][2], bf_u32x4 (&x)[8][3], bf_u32x4 (&y)[8][3], BfRing& ring, int lane,
                                                  const float* bias, int q, BfRollPend& e, const BfRollOut& prev, const BfRollOut& cur) {
    constexpr int EX = MODE != 0 ? 2 : 0;
    bf_u32x4 fa[2][3], fb[2][3];
    BF_PIN();
    const unsigned char* base = bf_stage_enter<4, E01>(ring) + lane * 16;
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {
        fa[0][pl] = *reinterpret_cast<const bf_u32x4*>(base + pl * BF_BLOCK);
        fa[1][pl] = *reinterpret_cast<const bf_u32x4*>(base + (3 + pl) * BF_BLOCK);
        BF_PIN();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f32x4& c0 = cc[j & 1][0];
        f32x4& c1 = cc[j & 1][1];
        const f32x4& d0 = cc[(j & 1) ^ 1][0];        // the pair before (j = 0: pair 7 of the step before)
        const f32x4& d1 = cc[(j & 1) ^ 1][1];
        c0 = *reinterpret_cast<const f32x4*>(bias + 32 * j + 4 * q);
        c1 = *reinterpret_cast<const f32x4*>(bias + 32 * j + 16 + 4 * q);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool first = (j == 0 && h == 0), last = (j == 7 && h == 1);
            // the epilogue of the pair before rides in the first stage of this pair: slots 0 - 5 in its second six products, 6 - 10 in
            // its third (the first carries the weight group's last pieces, the fourth the next stage's entry)
            const bool hook = (h == 0) && (j > 0 || PENDING_IN);
            const int P = (j + 7) & 7;
            const BfRollOut& o = (j == 0) ? prev : cur;
            bf_u32x4 (&out)[3] = (j == 0) ? x[7] : y[(j + 7) & 7];
#define BF_ROLL_HOOK(ON, S0) &ring, (ON), (S0), P, e, o, d0, d1, out
            if (first) {
                bf_roll_six_part<MODE, 0, 6, true, 0, 1, 2, 3, 4, 5, 0>(c0, c1, fa[0], fa[1], x[4 * h + 0], fb[0], fb[1], base + 6 * BF_BLOCK, BF_ROLL_HOOK(false, 0));
                bf_roll_six_part<MODE, 0, 6, true, 0, 1, 2, 3, 4, 5, 6>(c0, c1, fb[0], fb[1], x[4 * h + 1], fa[0], fa[1], base + 12 * BF_BLOCK, BF_ROLL_HOOK(hook, 0));
            } else {
                bf_roll_six_part<MODE, 0, 6, true, 0, 1, 2, 3, 4, 5, 4>(c0, c1, fa[0], fa[1], x[4 * h + 0], fb[0], fb[1], base + 6 * BF_BLOCK, BF_ROLL_HOOK(false, 0));
                bf_roll_six_part<MODE, 0, 6, true, 0, 1, 2, 3, 4, 5, 10>(c0, c1, fb[0], fb[1], x[4 * h + 1], fa[0], fa[1], base + 12 * BF_BLOCK, BF_ROLL_HOOK(hook, 0));
            }
            bf_roll_six_part<MODE, 0, 6, true, 0, 1, 2, 3, 4, 5, -1>(c0, c1, fa[0], fa[1], x[4 * h + 2], fb[0], fb[1], base + 18 * BF_BLOCK, BF_ROLL_HOOK(hook, 6));
            if (!last) {
                bf_roll_six_part<MODE, 0, 2, false, 0, 1, 2, 3, 4, 5, -1>(c0, c1, fb[0], fb[1], x[4 * h + 3], fa[0], fa[1], base, BF_ROLL_HOOK(false, 0));
                // entries of this step: 1 (inside stage 0) sees E01, 2 (inside stage 1) none, the others two stores
                base = (first ? bf_stage_enter<4, E01>(ring) : (j == 0 && h == 1) ? bf_stage_enter<4, 0>(ring) : bf_stage_enter<4, EX>(ring)) + lane * 16;
                BF_PIN();
                bf_roll_six_part<MODE, 2, 6, true, 2, 2, 3, 3, 4, 5, 0>(c0, c1, fb[0], fb[1], x[4 * h + 3], fa[0], fa[1], base, BF_ROLL_HOOK(false, 0));
            } else {
                bf_roll_six_part<MODE, 0, 6, false, 0, 1, 2, 3, 4, 5, -1>(c0, c1, fb[0], fb[1], x[4 * h + 3], fa[0], fa[1], base, BF_ROLL_HOOK(false, 0));
            }
#undef BF_ROLL_HOOK
        }
    }
}

}  // namespace morl
