// Layer-fused MLP chain on the bf16 matrix cores with fp32-class accuracy (gfx950, wave64): every fp32 GEMM of a chain is
// evaluated as SIX products of three-way bf16 splits,
//
//     a = a_hi + a_mid + a_lo  (8 + 8 + 8 significand bits, each part the round-to-nearest bf16 of what the parts before it left),
//     a * b ~ a_lo*b_hi + a_mid*b_mid + a_hi*b_lo + a_mid*b_hi + a_hi*b_mid + a_hi*b_hi        (all terms >= 2^-16 of |a*b|),
//
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  bf16 keeps fp32's exponent, so the split is exact for every finite fp32 value
// (no range assumption: gradients of 1e-9 split as well as activations of 1e3); the dropped terms are <= 3 * 2^-24 of |a*b|, the
// class of an fp32 multiply's own rounding.  Measured on MI355X (tools/probes): max |Q - Q_float64| 2.4e-7 against 2.3e-7 for the
// k-ordered fp32 chain of mlp_chain2.h, at 6/16 of its matrix-core time (the f32-input MFMA runs at the VECTOR rate, 1/16 of bf16).
//
// Replaces, for the Envelope step's big launches (morl_hip.hip): QNet.forward of the online next-state pass and of the training
// pass (envelope.py:300, :420; common/networks.py:10-48) and the dX half of loss.backward() (envelope.py:323).
//
// Layout, chosen so that ACTIVATIONS NEVER LEAVE THE REGISTERS between layers:
//   * a wave owns 16 rows through the whole chain and computes the TRANSPOSED product D^T[feature][row] = W[feature][k] x^T[k][row]
//     per 16-feature tile: A operand = weights (lane (n, q): feature 16t + n, eight contraction slots of group q), B operand =
//     activations (lane (m, q): row m, the same eight slots), D: lane (m, q) holds features 16t + 4q + 0..3 of row m.
//   * which contraction index a slot stands for is free as long as A and B agree, so the slots of a hidden layer's k-step s are
//     DEFINED as what the lane already holds: slot (q, e) <-> feature 32s + 16(e >> 2) + 4q + (e & 3) = register e & 3 of tile
//     2s + (e >> 2).  The epilogue (bias is the accumulator's initial value; ReLU; split; pack) is lane-local: no LDS, no barrier,
//     no shuffle between layers.  The weights are laid out to match, once per step, by bf_split_kernel.
//   * weights are the shared operand: the four waves of a workgroup (64 rows) read them from LDS, staged ONCE per workgroup by
//     LDS-DMA (global_load_lds, 16 bytes per lane) through a ring of three 24 KB stages that runs ahead across layer boundaries
//     (the whole chain's weights are one linear stream of 1 KB fragment blocks in consumption order: block = one (k-step, tile,
//     split part), lane l's 16 bytes at offset 16 l -- the ds_read_b128 of a fragment is 1 KB contiguous, conflict-free).
//     One s_barrier per stage (48 MFMAs per wave); counted vmcnt, never 0 in the loop.
//   * two workgroups per CU (77 KB of LDS, <= 256 VGPRs): one's barriers and epilogues fall into the other's MFMA phases.
// Per row and layer: 2*K*N flop algorithmic, 6x that on the bf16 pipe; HBM bytes = the chain's inputs and outputs only (saved
// activations / gradients of the training passes are written once, fp32, from the accumulators).
#pragma once
#include "mlp_chain.h"
#include "envelope_kernels.h"

namespace morl {

constexpr int BF_ROWS_WAVE = 16;                 // rows a wave carries
constexpr int BF_TM = 64;                        // rows per workgroup of the 4-wave form (the 2-wave form: 32)
constexpr int BF_BLOCK = 1024;                   // bytes of one fragment block: 64 lanes x 8 bf16
constexpr int BF_STAGE_BLOCKS = 24;              // blocks per ring stage (24 KB): half a k-step of a 256-wide layer
constexpr int BF_STAGE_BYTES = BF_STAGE_BLOCKS * BF_BLOCK;
constexpr int BF_RING = 3;                       // stages resident: one being multiplied, two in flight
constexpr int BF_MAX_STEPS = MORL_MAX_LAYERS;
constexpr int BF_WIDE = 256;                     // columns of every wide step

typedef unsigned int bf_u32x4 __attribute__((ext_vector_type(4)));      // eight bf16
typedef __bf16 bf_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf_bf16x2 __attribute__((ext_vector_type(2)));

// One step (dense layer) of a chain.  Wide steps have N = 256 columns (16 tiles); the narrow last step (the Q head) N <= 32.
struct BfStep {
    const float* bias;                     // [N] fp32, or NULL (backward chain: no bias)
    float* out;                            // or NULL: fp32 copy of this step's output, [rows][ldout] (saved h_l / g_l; the head: Q)
    unsigned long long* bits_out;          // or NULL: (output > 0) of a wide step, one 64-bit word per lane: word[(row / 16) * 64 + lane],
    const unsigned long long* bits_in;     //          bit 4 * tile + r <-> feature 16 * tile + 4 * (lane >> 4) + r of row (lane & 15);
                                           // bits_in: the output is kept where the bit is set (ReLU backward)
    int N, K;                              // real columns / contraction length
    int ldout;
    int relu;
};

struct BfChain {
    const unsigned char* stream;           // split weights of the whole chain in consumption order (bf_split_kernel)
    BfStep step[BF_MAX_STEPS];
    int n_steps;                           // first step wide with k0_steps k-steps, then wide steps of 8 k-steps, then (head) one narrow step
    int k0_steps;                          // 1 or 2: k-steps (32 contraction indices each) of the first step, natural order
    int head;                              // 1: the last step is narrow
    int n_stages;                          // ring stages of the stream
    int rows;
    // input rows (as ChainArgs): 0 cat(obs[b], weights[i]) with row -> (b, i) by row_order; 1 dense src[rows][ldsrc], K0 columns;
    // 2 (backward chain): dLoss/dQ of TD row `row`, computed here (BfMulti::tdb), K0 = A * R columns
    int in_mode;
    const float* obs;
    const float* weights;
    int B, W, D, R, row_order;
    const float* src;
    int ldsrc, K0;
    float* x0_out;                         // in_mode 0, or NULL: the assembled rows also go to HBM as [rows][ldx0] (zero padded)
    int ldx0;
    // mlp_chain_bfn.h only, in_mode 3 (the lazily evaluated target rows): row r is the pair pairs[r] = b * W + j, i.e.
    // cat(obs[b], weights[j]); rows = the bound the grid was sized for, *rows_dev the count; count_mirror / count_tag as ChainArgs'
    const int* rows_dev;
    const int32_t* pairs;
    unsigned long long* count_mirror;
    unsigned int count_tag;
    int amax;                              // 1: the head's output is the online next-state slab of an Envelope step whose row tiles are
                                           //    whole transitions (tile rows = W, 2 W or 4 W): the workgroup also takes the arg-max of its
                                           //    transitions (BfMulti::td, envelope_argmax_tile) -- no separate arg-max launch
};

// The TD stage of a lazily evaluated Envelope step (envelope_td_kernel<2>: target from the compact target rows, TD error, loss
// gradient wrt Q, loss partials, priorities) taken over by the BACKWARD chain's workgroups: BfChain::in_mode = 2, the chain's input row
// (dLoss/dQ of its TD row) is computed where it is needed instead of by a launch in front (6 us of latencies).  Row tiles are whole
// transitions; same arithmetic, same summation order of the loss partials as the kernel (tests/test_chain_tilings.py holds the bits).
struct BfTdArgs {
    const int32_t* best_io;     // [rows] flattened (j*, a*) per TD row (row = b * W + i)
    const int32_t* row_slot;    // [rows] compact target row of the TD row's (b, j*)
    const float* qt;            // compact target rows [slots][A][R]
    const float* q_main;        // [rows][ldq] Q_online(s_b, w_i)
    const int32_t* actions;     // [B]
    const float* rewards;       // [B][R]
    const float* dones;         // [B]
    const float* weights;       // [W][R]
    float* dq;                  // [rows][ldq] out: dLoss/dQ (the head's weight gradients read it)
    double* loss_part;          // [B][2] out
    float* priority;            // [B] out, or NULL
    int B, W, A, R, ldq;
    float gamma, c_mse, c_aux;
};

constexpr int BF_MAX_MULTI = 2;
struct BfMulti {
    BfChain c[BF_MAX_MULTI];
    int tile_start[BF_MAX_MULTI + 1];      // 64-row tiles of chain q: [tile_start[q], tile_start[q + 1])
    int n;
    EnvArgmaxArgs td;                      // arguments of the arg-max stage for the chain with amax = 1
    BfTdArgs tdb;                          // arguments of the TD stage for the chain with in_mode = 2
    long long* prof;                       // development builds only (-DBF_PROF, MORL_BF_PROF=1 python build.py): [tile][wave][BF_PROF_SLOTS]
};
// Phase stamps of a development build: s_memtime sums per wave -- 0 prologue (input rows, biases, first weight stages), 1 / 2 first
// step's MFMA loop / epilogue, 3 / 4 the 256 x 256 steps' MFMA loops / epilogues, 5 head + drain, 6 of which waiting at stage
// entries (counted wait + barrier), 7 of which issuing the weight DMA, 8 first stamp, 9 last stamp
constexpr int BF_PROF_SLOTS = 10;
#ifdef BF_PROF
#define BF_T(q) { const long long t_ = clock64(); pt[q] += t_ - tp_; tp_ = t_; }
#else
#define BF_T(q)
#endif

// ---- three-way split ---------------------------------------------------------------------------------------------------------
// two fp32 -> packed pair of round-to-nearest-even bf16 (v_cvt_pk_bf16_f32): a in the low half, b in the high half
__device__ __forceinline__ unsigned bf_pack2(float a, float b) {
    const bf_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf_bf16x2));
}
__device__ __forceinline__ float bf_low(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_high(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// x - y / x + y as ONE scalar instruction each, out of the SLP vectoriser's reach: it pairs neighbouring fp32 operations into
// v_pk_add_f32, which issues at 24.5 cycles beside a wave that keeps the SIMD's matrix pipe busy -- every plain vector instruction
// (v_sub_f32, v_cvt_pk_bf16_f32, v_and_b32 ...) at 9.7 -- and wants its operands in register PAIRS (a v_mov per operand on top).
// Measured by tools/probes/valu_mfma_probe.hip; the split code below runs beside MFMAs in every kernel that uses it.
__device__ __forceinline__ float bf_sub(float x, float y) {
#ifdef HIPSIM_EMULATED
    return x - y;
#else
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#endif
}
__device__ __forceinline__ float bf_add(float x, float y) {
#ifdef HIPSIM_EMULATED
    return x + y;
#else
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#endif
}
// (a, b) -> hi / mid / lo pairs; the two subtractions are exact (the difference of an fp32 and its bf16 rounding fits fp32)
__device__ __forceinline__ void bf_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = bf_pack2(a, b);
    const float ra = bf_sub(a, bf_low(hi)), rb = bf_sub(b, bf_high(hi));
    mid = bf_pack2(ra, rb);
    lo = bf_pack2(bf_sub(ra, bf_low(mid)), bf_sub(rb, bf_high(mid)));
}

__device__ __forceinline__ f32x4 bf_mfma(const bf_u32x4& w, const bf_u32x4& x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf_bf16x8, w), __builtin_bit_cast(bf_bf16x8, x), c, 0, 0, 0);
}

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima)
#define BF_VMCNT(N) __builtin_amdgcn_s_waitcnt((((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (15 << 8)))
// s_waitcnt vmcnt(N) lgkmcnt(0): the stage boundary -- besides the DMA group, every LDS read this wave has issued has RETURNED (the
// scheduler moves the last fragment reads of a stage up to the barrier; the buffer they read is the one the DMA issued right after
// the barrier overwrites)
#define BF_VMCNT_LDS0(N) __builtin_amdgcn_s_waitcnt((((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (0 << 8)))

// ---- the weight ring ------------------------------------------------------------------------------------------------------------
struct BfRing {
    __amdgpu_buffer_rsrc_t rsrc;   // the whole stream
    int voff;                      // this lane's byte offset inside a stage: wave * (24 / NW) KB + lane * 16
    unsigned char* lds;            // ring base (workgroup)
    int t;                         // stage the workgroup multiplies next
    int buf;                       // t % 3
    int n_stages;
    int wave;
    int dma_soff;                  // the DMA group a wide stage's entry left to its MFMA steps (bf_ring_piece): byte offset of the
    unsigned char* dma_lds;        //   stage in the stream, this wave's share of the LDS buffer it goes to
#ifdef BF_PROF
    long long t_entry, t_dma;
#endif
};

// LDS-DMA of this wave's share (24 / NW blocks) of stage `st` (clamped to the stream's last stage: the issue count per stage is
// static) into `buf`: buffer_load_dwordx4 ... lds, 1 KB per instruction (lane l's 16 bytes land at M0 + offset + 16 l; the
// instruction offset counts on both sides).  The MUBUF form, not global_load_lds: a pending FLAT-encoded LDS access makes hipcc's
// wait insertion treat every LDS counter as unordered -- each fragment read was then waited for with lgkmcnt(0) instead of a
// counted wait.
template <int NW>
__device__ __forceinline__ void bf_ring_issue(const BfRing& r, int st, int buf) {
    constexpr int DPW = BF_STAGE_BLOCKS / NW;
    static_assert(DPW == 6 || DPW == 12, "four or two waves per workgroup");
    const int s = st < r.n_stages ? st : r.n_stages - 1;
    const int soff = s * BF_STAGE_BYTES;
    unsigned char* l = r.lds + buf * BF_STAGE_BYTES + r.wave * (DPW * BF_BLOCK);
    // (instruction offsets are 12-bit immediates: every group of four pieces goes through the scalar offset / M0)
#define BF_DMA4(G, N)                                                                                                              \
    {                                                                                                                              \
        __attribute__((address_space(3))) void* lg = (__attribute__((address_space(3))) void*)(l + (G) * 4 * BF_BLOCK);            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rsrc, lg, 16, r.voff, soff + (G) * 4 * BF_BLOCK, 0 * BF_BLOCK, 0);              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rsrc, lg, 16, r.voff, soff + (G) * 4 * BF_BLOCK, 1 * BF_BLOCK, 0);              \
        if ((N) > 2) {                                                                                                             \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rsrc, lg, 16, r.voff, soff + (G) * 4 * BF_BLOCK, 2 * BF_BLOCK, 0);          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rsrc, lg, 16, r.voff, soff + (G) * 4 * BF_BLOCK, 3 * BF_BLOCK, 0);          \
        }                                                                                                                          \
    }
    BF_DMA4(0, 4)
    if (DPW == 6) BF_DMA4(1, 2)
    else { BF_DMA4(1, 4) BF_DMA4(2, 4) }
#undef BF_DMA4
}

// One 1 KB piece (0 .. 24 / NW - 1) of the group set up by bf_stage_enter: issued BETWEEN the MFMAs of the stage (one piece behind
// a product step's two MFMAs), where its ~25 - 45 issue cycles run under the matrix pipe's 32 -- all of a group in one burst
// right behind the barrier was a seventh of a wave's time in the backward chain (one wave per SIMD: nobody else to feed the pipe).
template <int NW, int PIECE>
__device__ __forceinline__ void bf_ring_piece(const BfRing& r) {
    constexpr int DPW = BF_STAGE_BLOCKS / NW;
    if (PIECE >= 0 && PIECE < DPW) {
        constexpr int G = (PIECE < 0 ? 0 : PIECE) / 4, J = (PIECE < 0 ? 0 : PIECE) % 4;
        __attribute__((address_space(3))) void* lg = (__attribute__((address_space(3))) void*)(r.dma_lds + G * 4 * BF_BLOCK);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rsrc, lg, 16, r.voff, r.dma_soff + G * 4 * BF_BLOCK, J * BF_BLOCK, 0);
    }
}

// Entry of a WIDE stage: as bf_stage_begin, but the DMA group of the stage two ahead is only set up here; the stage's product steps
// issue its pieces (every one of them before the next entry, so the counted waits see the same issue order).
template <int NW, int EXTRA>
__device__ __forceinline__ const unsigned char* bf_stage_enter(BfRing& r) {
#ifdef BF_PROF
    const long long t0_ = clock64();
#endif
    BF_VMCNT_LDS0(BF_STAGE_BLOCKS / NW + EXTRA);
    __builtin_amdgcn_s_barrier();
#ifdef BF_PROF
    r.t_entry += clock64() - t0_;
#endif
    const int free_buf = r.buf == 0 ? 2 : r.buf - 1;
    const int st = r.t + 2 < r.n_stages ? r.t + 2 : r.n_stages - 1;       // (clamped: the issue count per stage is static)
    r.dma_soff = st * BF_STAGE_BYTES;
    r.dma_lds = r.lds + free_buf * BF_STAGE_BYTES + r.wave * ((BF_STAGE_BLOCKS / NW) * BF_BLOCK);
    const unsigned char* base = r.lds + r.buf * BF_STAGE_BYTES;
    r.buf = r.buf == 2 ? 0 : r.buf + 1;
    ++r.t;
    return base;
}

// PRODUCER-WAVE form (round 6, the backward launch: mlp_chain_bf_pw_kernel).  A fifth wave of the workgroup issues every weight group --
// all 24 pieces of a stage -- and waits for them; the four MFMA waves issue none.  Why: a 1 KB LDS-DMA instruction costs the wave that
// issues it ~40 issue cycles, six per stage and wave were ~5 of the backward chain's 23 cycles per MFMA, and a wave's own instructions
// do not run under its MFMAs (profiles/r06_mfma_same_wave_probe.txt) -- another wave's do.
// entry of a stage for an MFMA wave: its own fragment reads have returned, then the workgroup's barrier (the producer arrives there
// once the stage has landed)
__device__ __forceinline__ const unsigned char* bf_stage_enter_pw(BfRing& r) {
#ifdef BF_PROF
    const long long t0_ = clock64();
#endif
    __builtin_amdgcn_s_waitcnt((0 << 8) | (7 << 4) | 15 | (3 << 14));     // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
#ifdef BF_PROF
    r.t_entry += clock64() - t0_;
#endif
    const unsigned char* base = r.lds + r.buf * BF_STAGE_BYTES;
    r.buf = r.buf == 2 ? 0 : r.buf + 1;
    ++r.t;
    return base;
}
// the producer wave: stages 0 and 1 at once, then per stage entry -- stage t has landed (everything but the 24 pieces of stage t + 1),
// barrier (the MFMA waves are done with stage t - 1), stage t + 2 into the buffer that became free.  `pre_barriers`: workgroup barriers
// the MFMA waves pass before their first stage entry.
__device__ __forceinline__ void bf_ring_producer(const unsigned char* stream, int n_stages, unsigned char* ring_lds, int lane, int pre_barriers) {
    // (the producer at raised priority -- s_setprio 3 -- changes nothing in the backward launch and costs the one-round forward launch
    // 0.9 us: profiles/r06_producer_wave_ab.json)
    BfRing r;
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, n_stages * BF_STAGE_BYTES, 0x00020000);
    r.lds = ring_lds;
    r.n_stages = n_stages;
    // (as one "wave of a one-wave workgroup": share = the whole stage, 24 pieces in two calls of the two-wave issue)
    auto issue = [&](int st, int buf) {
        r.wave = 0; r.voff = lane * 16;
        bf_ring_issue<2>(r, st, buf);
        r.wave = 1; r.voff = 12 * BF_BLOCK + lane * 16;
        bf_ring_issue<2>(r, st, buf);
    };
    issue(0, 0);
    issue(1, 1);
    for (int b = 0; b < pre_barriers; ++b) __builtin_amdgcn_s_barrier();
    int buf = 2;
    for (int t = 0; t < n_stages; ++t) {
        BF_VMCNT(BF_STAGE_BLOCKS);
        __builtin_amdgcn_s_barrier();
        issue(t + 2, buf);
        buf = buf == 2 ? 0 : buf + 1;
    }
    BF_VMCNT(0);
}

// Entry of stage r.t: my share of it has landed (EXTRA = vector-memory instructions this lane issued AFTER the DMA group of the
// stage following it -- epilogue stores -- which retire in issue order behind the group waited for) and every LDS read this wave
// has issued has returned; after the barrier everybody's share has landed and everybody is done reading the stage before it, whose
// buffer takes the stage two ahead.  Returns the stage's LDS base.
template <int NW, int EXTRA>
__device__ __forceinline__ const unsigned char* bf_stage_begin(BfRing& r) {
#ifdef BF_PROF
    const long long t0_ = clock64();
#endif
    BF_VMCNT_LDS0(BF_STAGE_BLOCKS / NW + EXTRA);
    __builtin_amdgcn_s_barrier();
#ifdef BF_PROF
    const long long t1_ = clock64();
#endif
    const int free_buf = r.buf == 0 ? 2 : r.buf - 1;
    bf_ring_issue<NW>(r, r.t + 2, free_buf);
#ifdef BF_PROF
    r.t_entry += t1_ - t0_; r.t_dma += clock64() - t1_;
#endif
    const unsigned char* base = r.lds + r.buf * BF_STAGE_BYTES;
    r.buf = r.buf == 2 ? 0 : r.buf + 1;
    ++r.t;
    return base;
}

// the three split parts of the fragment of local tile `t` of a stage (blocks 3t .. 3t + 2), lane's 16 bytes each
__device__ __forceinline__ void bf_frag_load(bf_u32x4 (&f)[3], const unsigned char* lane_base, int t) {
#pragma unroll
    for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const bf_u32x4*>(lane_base + (3 * t + p) * BF_BLOCK);
}

// The six products of a tile pair, smallest first (lo*hi, mid*mid, hi*lo, mid*hi, hi*mid, hi*hi: w part, x part), the two tiles
// alternating so that no two consecutive MFMAs share an accumulator (a dependent v_mfma_f32_16x16x32_bf16 waits for its predecessor's
// last pass).  READ: the fragment reads of the NEXT pair ride along, one per product step, twelve MFMAs (>= 192 cycles) ahead of
// their first use.  The order is PINNED with sched_barrier: left alone, hipcc clusters the MFMAs of one accumulator and sinks
// every ds_read next to its first use (waiting for it on the spot).
#define BF_PIN() __builtin_amdgcn_sched_barrier(0)
// product steps [P0, P1) of a pair; READ: fragment k of the next pair (k = 0..5 in consumption order: lo, lo, mid, mid, hi, hi) is
// read in front of product step rd[k] (steps outside [P0, P1) read nothing)
// NW / DMA0: product step p also issues piece DMA0 + (p - P0) of the weight group its stage's entry set up (DMA0 < 0: none).
template <int P0, int P1, bool READ, int R0 = 0, int R1 = 1, int R2 = 2, int R3 = 3, int R4 = 4, int R5 = 5, int NW = 4, int DMA0 = -1>
__device__ __forceinline__ void bf_six_part(f32x4& c0, f32x4& c1, const bf_u32x4 (&w0)[3], const bf_u32x4 (&w1)[3], const bf_u32x4 (&x)[3],
                                            bf_u32x4 (&n0)[3], bf_u32x4 (&n1)[3], const unsigned char* next_base,
                                            const BfRing* ring = nullptr) {
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
        if (DMA0 >= 0) {
            // (compile-time piece index: the instruction's offset field is an immediate)
            if (p - P0 == 0) bf_ring_piece<NW, DMA0>(*ring);
            if (p - P0 == 1) bf_ring_piece<NW, DMA0 < 0 ? -1 : DMA0 + 1>(*ring);
            if (p - P0 == 2) bf_ring_piece<NW, DMA0 < 0 ? -1 : DMA0 + 2>(*ring);
            if (p - P0 == 3) bf_ring_piece<NW, DMA0 < 0 ? -1 : DMA0 + 3>(*ring);
            if (p - P0 == 4) bf_ring_piece<NW, DMA0 < 0 ? -1 : DMA0 + 4>(*ring);
            if (p - P0 == 5) bf_ring_piece<NW, DMA0 < 0 ? -1 : DMA0 + 5>(*ring);
        }
        BF_PIN();
    }
}
template <bool READ, int NW = 4, int DMA0 = -1>
__device__ __forceinline__ void bf_six_pair(f32x4& c0, f32x4& c1, const bf_u32x4 (&w0)[3], const bf_u32x4 (&w1)[3], const bf_u32x4 (&x)[3],
                                            bf_u32x4 (&n0)[3], bf_u32x4 (&n1)[3], const unsigned char* next_base,
                                            const BfRing* ring = nullptr) {
    bf_six_part<0, 6, READ, 0, 1, 2, 3, 4, 5, NW, DMA0>(c0, c1, w0, w1, x, n0, n1, next_base, ring);
}

// One wide step (256 columns = 16 tiles) over KSTEPS k-steps: two stages per k-step (tiles 0-7, 8-15), per stage four tile pairs.
// acc[T][r] += sum_k W[16T + 4q + r][k] * x[k]  for this lane's row.  The pipeline runs ACROSS the stage boundaries: the entry of
// stage i + 1 (wait, barrier, DMA issue) sits INSIDE the last pair of stage i, whose remaining MFMAs then cover the LDS latency of
// stage i + 1's first fragments -- its own fragments are all in registers by then, so the buffer of stage i is free for the DMA the
// entry issues.  EXTRA0: see bf_stage_begin, applies to the step's first two stages.
template <int NW, int KSTEPS, int EXTRA0, bool PW = false>
__device__ __forceinline__ void bf_wide_step(f32x4 (&acc)[16], const bf_u32x4 (&x)[8][3], BfRing& ring, int lane) {
    bf_u32x4 fa[2][3], fb[2][3];
    BF_PIN();
    const unsigned char* base = (PW ? bf_stage_enter_pw(ring) : bf_stage_enter<NW, EXTRA0>(ring)) + lane * 16;
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {       // (consumption order, pinned: the first MFMAs wait for the first two reads only)
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
            // the weight group this stage's entry set up goes out one piece per product step: from piece 0 when the entry was the
            // step's first (in front of the stage), from piece 4 when it sat inside the previous stage's last pair (pieces 0 - 3 there)
            if (PW) {
                bf_six_pair<true>(acc[T + 0], acc[T + 1], fa[0], fa[1], x[s], fb[0], fb[1], base + 6 * BF_BLOCK);
                bf_six_pair<true>(acc[T + 2], acc[T + 3], fb[0], fb[1], x[s], fa[0], fa[1], base + 12 * BF_BLOCK);
            } else if (s == 0 && hf == 0) {
                bf_six_pair<true, NW, 0>(acc[T + 0], acc[T + 1], fa[0], fa[1], x[s], fb[0], fb[1], base + 6 * BF_BLOCK, &ring);
                bf_six_pair<true, NW, 6>(acc[T + 2], acc[T + 3], fb[0], fb[1], x[s], fa[0], fa[1], base + 12 * BF_BLOCK, &ring);
            } else {
                bf_six_pair<true, NW, 4>(acc[T + 0], acc[T + 1], fa[0], fa[1], x[s], fb[0], fb[1], base + 6 * BF_BLOCK, &ring);
                bf_six_pair<true, NW, 10>(acc[T + 2], acc[T + 3], fb[0], fb[1], x[s], fa[0], fa[1], base + 12 * BF_BLOCK, &ring);
            }
            bf_six_pair<true>(acc[T + 4], acc[T + 5], fa[0], fa[1], x[s], fb[0], fb[1], base + 18 * BF_BLOCK);
            if (!last) {
                // two product steps first: by the time the wave reaches the stage entry's wait, this pair's own fragments (read
                // during the pair before) have long returned -- then the entry, then the other four steps with the next stage's first
                // fragments riding along (2, 2, 1, 1) and the first four pieces of the group the entry set up
                bf_six_part<0, 2, false>(acc[T + 6], acc[T + 7], fb[0], fb[1], x[s], fa[0], fa[1], base);
                base = (PW ? bf_stage_enter_pw(ring) : (s == 0 && hf == 0) ? bf_stage_enter<NW, EXTRA0>(ring) : bf_stage_enter<NW, 0>(ring)) + lane * 16;
                BF_PIN();
                if (PW) bf_six_part<2, 6, true, 2, 2, 3, 3, 4, 5>(acc[T + 6], acc[T + 7], fb[0], fb[1], x[s], fa[0], fa[1], base);
                else bf_six_part<2, 6, true, 2, 2, 3, 3, 4, 5, NW, 0>(acc[T + 6], acc[T + 7], fb[0], fb[1], x[s], fa[0], fa[1], base, &ring);
            } else {
                bf_six_pair<false>(acc[T + 6], acc[T + 7], fb[0], fb[1], x[s], fa[0], fa[1], base);
            }
        }
    }
}

// The narrow last step (N <= 32: NT tiles) over 8 k-steps: 24 / (3 NT) k-steps per stage.
template <int NW, int NT, int EXTRA0, bool PW = false>
__device__ __forceinline__ void bf_head_step(f32x4 (&acc)[2], const bf_u32x4 (&x)[8][3], BfRing& ring, int lane) {
    constexpr int KS_PER_STAGE = BF_STAGE_BLOCKS / (3 * NT);
    constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int s0 = 0; s0 < 8; s0 += KS_PER_STAGE) {
        BF_PIN();
        const unsigned char* base = (PW ? bf_stage_enter_pw(ring) : bf_stage_begin<NW, EXTRA0>(ring)) + lane * 16;       // (one or two stages: both behind the epilogue's stores)
        bf_u32x4 f[KS_PER_STAGE][NT][3];
#pragma unroll
        for (int ks = 0; ks < KS_PER_STAGE; ++ks)
#pragma unroll
            for (int t = 0; t < NT; ++t) bf_frag_load(f[ks][t], base, ks * NT + t);
        BF_PIN();
        // (one tile: its products are a dependent chain per k-step; the k-steps of the stage alternate instead)
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int ks = 0; ks < KS_PER_STAGE; ++ks)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = bf_mfma(f[ks][t][pw[p]], x[s0 + ks][px[p]], acc[t]);
    }
}

// accumulators of a step start from its bias (LDS copy, [256] floats per step; zeros for a step without bias)
template <int NTILES>
__device__ __forceinline__ void bf_acc_init(f32x4 (&acc)[NTILES], const float* bias_lds, int q) {
#pragma unroll
    for (int T = 0; T < NTILES; ++T) acc[T] = *reinterpret_cast<const f32x4*>(bias_lds + 16 * T + 4 * q);
}

// Epilogue of a wide step, lane-local: (ReLU | mask bits) -> fp32 copy to HBM -> sign bits -> split into the next step's B operand.
// Vector-memory instructions a wave issues per epilogue when MODE != 0: 16 row stores + 1 (the sign-bit word's store of the training
// forward, or the NEXT step's mask-word load of the backward chain) -- BF_SAVE_VMEM, counted by the next layer's first two waits.
// `keep`: the step's mask word (bits_in), loaded by the caller at the START of the step -- a load inside the epilogue would be waited
// for with vmcnt(0), i.e. behind both weight stages in flight (the first version did: one drained ring per layer).
constexpr int BF_SAVE_VMEM = 17;
// -(bit `pos` of `word`): v_bfe_i32's sign-extended one-bit field
__device__ __forceinline__ int bf_bit_mask(int word, int pos) {
#if defined(HIPSIM_EMULATED) || defined(BF_SHIFT_MASK)        // (BF_SHIFT_MASK: the A/B build of profiles/r06_mask_bfe_ab.txt)
    return (word << (31 - pos)) >> 31;
#else
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "s"(pos));
    return m;
#endif
}
// ReLU as ONE instruction: max on the bit patterns as signed integers (positive floats are positive integers, everything with the
// sign bit set -- negative values, -0 -- is a negative integer); fmaxf(a, 0) costs a second v_max (IEEE NaN quieting)
__device__ __forceinline__ float bf_relu(float a) {
    const int u = __builtin_bit_cast(int, a);
    return __builtin_bit_cast(float, u > 0 ? u : 0);
}

// MODE: 0 no-grad forward (nothing leaves but the next operand), 1 training forward (fp32 copy + sign bits out), 2 backward (mask
// word in, fp32 copy out) -- compile-time, so that neither chain carries the other's bit arithmetic (two to three vector
// instructions per value: a sixth of the epilogue).
template <int MODE>
__device__ __forceinline__ void bf_wide_epilogue(const f32x4 (&acc)[16], bf_u32x4 (&x)[8][3], const BfStep& st, int rows, int row, bool row_ok,
                                                 size_t bits_idx, int q, unsigned long long keep) {
    unsigned pos_lo = 0u, pos_hi = 0u;
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)st.out, 0, (MODE != 0 && st.out != nullptr) ? rows * st.ldout * 4 : 0, 0x00020000);
    const int keep_lo = (int)(unsigned)keep, keep_hi = (int)(unsigned)(keep >> 32);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int T = 2 * s + (e >> 2), r = e & 3, k = 4 * T + r;
            float a = acc[T][r];
            // (a wide step of a forward chain is a hidden layer: ReLU, always -- bf_launch refuses anything else; under the step's run-time
            // flag every value cost a v_max AND a v_cndmask)
            if (MODE != 2) a = bf_relu(a);
            if (MODE == 2) {
                // a &= -(bit k of keep): the sign-extended one-bit field is the AND mask (v_bfe_i32 + v_and)
                // (as the instruction: written with shifts -- or as __builtin_amdgcn_sbfe -- hipcc turns the pair into v_and / v_cmp /
                // s_nop / v_cndmask: four issue slots a value where these are two)
                a = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & bf_bit_mask(k < 32 ? keep_lo : keep_hi, k & 31));
            }
            if (MODE == 1) {
                // after the ReLU the bit pattern is 0 or a positive integer: min(u, 1) is the sign bit (v_min_u32 + v_lshl_or)
                const unsigned u = __builtin_bit_cast(unsigned, a);
                const unsigned bit = u < 1u ? u : 1u;
                if (k < 32) pos_lo |= bit << k; else pos_hi |= bit << (k - 32);
            }
            v[e] = a;
        }
        if (MODE != 0) {
            // BUFFER stores with an out-of-range offset for rows beyond the chain (and for a step without a copy: an empty range):
            // issued by every wave whatever its rows.  A wave that skipped them would see fewer vector-memory instructions in flight
            // than BF_SAVE_VMEM says, wait for too little at the next stage entry -- and the other waves read ITS share of the stage
            // after the barrier.
            // (the row offset per pair of stores, as written: hoisted out of the loop -- one multiply instead of a branch, a scalar load
            // and a multiply per pair -- the forward launch took 71.5 us instead of 70.0 on one box, profiles/r06_epilogue_variants.txt:
            // what the co-resident tiles do to each other moves with every such change)
            const int voff = row_ok ? (row * st.ldout + 32 * s + 4 * q) * 4 : CH_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(bf_u32x4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                                                            __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])},
                                                   orsrc, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(bf_u32x4{__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]),
                                                            __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])},
                                                   orsrc, voff + 64, 0, 0);
        }
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bf_split2(v[2 * u], v[2 * u + 1], hi[u], mid[u], lo[u]);
        x[s][0] = bf_u32x4{hi[0], hi[1], hi[2], hi[3]};
        x[s][1] = bf_u32x4{mid[0], mid[1], mid[2], mid[3]};
        x[s][2] = bf_u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
    if (MODE == 1 && st.bits_out != nullptr) st.bits_out[bits_idx] = ((unsigned long long)pos_hi << 32) | pos_lo;
}

}  // namespace morl
#include "mlp_chain_bf_roll.h"
namespace morl {

// ---- one 64-row tile through a whole chain ----------------------------------------------------------------------------------------
// K0S: k-steps of the first step (1 or 2); MODE: see bf_wide_epilogue -- the chain writes per-step outputs (1, 2) and sign bits (1) or
// reads mask words (2); compile-time also because the counted waits depend on the stores issued.
// ROLL: the 256 x 256 steps behind the first one walk pair after pair with rolling epilogues (mlp_chain_bf_roll.h; the stream is then
// pair-major: BfSplitJob::pair_major) -- the backward chain on 64-row tiles
// PW: a fifth wave issues the weight ring (bf_ring_producer) -- the backward chain on 64-row tiles, 320 work-items
template <int NW, int K0S, int MODE, bool ROLL = false, bool PW = false>
__device__ __forceinline__ void bf_chain_body(const BfChain& p, int row0, unsigned char* ring_lds, float* bias_lds, long long* prof,
                                              const float* am_w, int32_t* am_best, int32_t* am_pairs, int32_t* am_slot,
                                              int32_t* am_count, int am_epoch, int am_B, int am_W, int am_A, int am_R, int am_flags,
                                              const BfTdArgs& tdb) {
#ifdef BF_PROF
    long long pt[6] = {0, 0, 0, 0, 0, 0}, tp_ = clock64();
    const long long tstart_ = tp_;
#endif
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (PW && wave == NW) {
        static_assert(!PW || !ROLL, "producer wave: not with rolling epilogues");
        bf_ring_producer(p.stream, p.n_stages, ring_lds, lane, 1);       // (one barrier in front of the stage entries: the bias copy's)
        return;
    }
    const int m = lane & 15, q = lane >> 4;
    const int row = row0 + 16 * wave + m;
    const bool row_ok = row < p.rows;
    const size_t bits_idx = (size_t)((row0 >> 4) + wave) * 64 + lane;        // (the same word whichever tile size carried the row)
#if defined(BF_PRIO_T)
    if (MODE == 1) __builtin_amdgcn_s_setprio(3);       // (development A/B: the training tiles of a forward launch before the no-grad tiles)
#elif defined(BF_PRIO_N)
    if (MODE == 0) __builtin_amdgcn_s_setprio(3);
#endif

    BfRing ring;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.stream, 0, p.n_stages * BF_STAGE_BYTES, 0x00020000);
    ring.voff = wave * ((BF_STAGE_BLOCKS / NW) * BF_BLOCK) + lane * 16;
    ring.lds = ring_lds;
    ring.t = 0; ring.buf = 0; ring.n_stages = p.n_stages; ring.wave = wave;
#ifdef BF_PROF
    ring.t_entry = 0; ring.t_dma = 0;
#endif
    // the stream starts before the input rows are assembled
    if (!PW) {
        bf_ring_issue<NW>(ring, 0, 0);
        bf_ring_issue<NW>(ring, 1, 1);
    }
    // biases -> registers, [step][256] (zeros beyond a step's columns and for steps without bias): every load issued here, in one go
    // -- compile-time step indices: as a loop over (step, column) every iteration read the step's pointer and width from the
    // argument block per LANE and waited for each of its three dependent loads with vmcnt(0), i.e. also for the weight stages just
    // put in flight: 30 serialised round trips, a tenth of the launch -- and written to LDS behind the input rows' loads
    constexpr int BPT = BF_WIDE / (64 * NW);        // columns per work-item per step
    float bv[BF_MAX_STEPS][BPT];
    // (round 5: as range-checked buffer loads -- a descriptor of N floats, or of none -- instead of loads under
    // `if (step exists && has a bias && column < N)`: the tests read the argument block step by step, a scalar-cache round trip
    // and a branch in front of every load)
#pragma unroll
    for (int s = 0; s < BF_MAX_STEPS; ++s) {
        const float* bp = p.step[s].bias;
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, (s < p.n_steps && bp != nullptr) ? p.step[s].N * 4 : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < BPT; ++j)
            bv[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (tid + j * 64 * NW) * 4, 0, 0));
    }

    // ---- input rows: natural contraction order, slot (q, e) of k-step s <-> column 32 s + 8 q + e -------------------------------
    bf_u32x4 x[8][3];
    {
        int b = row, w = row;
        if (p.in_mode == 0) {
            if (p.row_order == 0) { b = row / p.W; w = row - b * p.W; }
            else if (p.row_order == 1) { w = row / p.B; b = row - w * p.B; }
        }
        const int K0 = (p.in_mode == 0) ? p.D + p.R : p.K0;
        const float* src_a = (p.in_mode == 0) ? p.obs + (size_t)b * p.D : p.src + (size_t)row * p.ldsrc;
        const float* src_w = (p.in_mode == 0) ? p.weights + (size_t)w * p.R : nullptr;
        // ---- in_mode 2: the TD stage of this lane's TD row (envelope_td_kernel<2>'s arithmetic, operation for operation) ---------------
        float tdg[MORL_MAX_OBJ];
        int td_act = -1;
        if (MODE == 2 && K0S == 1 && p.in_mode == 2) {
            const BfTdArgs& t = tdb;
            const int W = t.W, A = t.A, R = t.R;
            const int tb = row / W, ti = row - tb * W;
            double m2 = 0.0, a2 = 0.0;
            float prv = 0.f;
#pragma unroll
            for (int r = 0; r < MORL_MAX_OBJ; ++r) tdg[r] = 0.f;
            if (row_ok) {
                const int bc = t.best_io[row];
                const float* qt = t.qt + ((size_t)t.row_slot[row] * A + (bc % A)) * R;
                td_act = t.actions[tb];
                const float* qm = t.q_main + (size_t)row * t.ldq + td_act * R;
                const float ndg = __fmul_rn(__fsub_rn(1.0f, t.dones[tb]), t.gamma);      // (1 - d) * gamma
                float td[MORL_MAX_OBJ], wi[MORL_MAX_OBJ];
                float wq = 0.f, wtq = 0.f;
#pragma unroll
                for (int r = 0; r < MORL_MAX_OBJ; ++r) {
                    td[r] = 0.f; wi[r] = 0.f;
                    if (r < R) {
                        wi[r] = t.weights[ti * R + r];
                        const float tq = __fadd_rn(t.rewards[(size_t)tb * R + r], __fmul_rn(ndg, qt[r]));
                        const float qv = qm[r];
                        td[r] = __fsub_rn(qv, tq);
                        wq = (r == 0) ? __fmul_rn(qv, wi[0]) : __fadd_rn(wq, __fmul_rn(qv, wi[r]));
                        wtq = (r == 0) ? __fmul_rn(tq, wi[0]) : __fadd_rn(wtq, __fmul_rn(tq, wi[r]));
                    }
                }
                const float daux = __fsub_rn(wq, wtq);
#pragma unroll
                for (int r = 0; r < MORL_MAX_OBJ; ++r)
                    if (r < R) {
                        tdg[r] = t.c_mse * td[r] + t.c_aux * daux * wi[r];
                        m2 += (double)td[r] * (double)td[r];
                    }
                a2 = (double)daux * (double)daux;
                prv = __fmul_rn(td[0], wi[0]);
#pragma unroll
                for (int r = 1; r < MORL_MAX_OBJ; ++r)
                    if (r < R) prv = __fadd_rn(prv, __fmul_rn(td[r], wi[r]));
                if (ti == 0 && q == 0 && t.priority != nullptr) t.priority[tb] = fabsf(prv);
            }
            // loss partials: one (sum td^2, (wQ - wTQ)^2) pair per row into LDS (ring buffer 2 is not in flight yet), summed per
            // transition behind the barrier below in the kernel's order (lane <-> TD row, wave_sum)
            if (q == 0) {
                double* sl = reinterpret_cast<double*>(ring_lds + 2 * BF_STAGE_BYTES);
                sl[(16 * wave + m) * 2 + 0] = m2;
                sl[(16 * wave + m) * 2 + 1] = a2;
            }
        }
#pragma unroll
        for (int s = 0; s < K0S; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * s + 8 * q + e;
                float a = 0.f;
                if (MODE == 2 && K0S == 1 && p.in_mode == 2) {
                    // column k = action * R + objective of dLoss/dQ: the row's gradient at the taken action, zero elsewhere
                    const int R = tdb.R, ka = k / R, kr = k - ka * R;
#pragma unroll
                    for (int r = 0; r < MORL_MAX_OBJ; ++r)
                        if (r == kr && ka == td_act && k < tdb.A * R) a = tdg[r];
                } else if (row_ok && k < K0) {
                    if (p.in_mode == 0) a = (k < p.D) ? src_a[k] : src_w[k - p.D];
                    else a = src_a[k];
                }
                v[e] = a;
            }
            if (MODE == 2 && K0S == 1 && p.in_mode == 2 && row_ok) {
                // the row of dLoss/dQ also goes to HBM, pad columns as zeros: the head's weight gradients read it (ldq is a multiple of 4)
                const int k = 8 * q;
                float* o = tdb.dq + (size_t)row * tdb.ldq + k;
                if (k < tdb.ldq) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                if (k + 4 < tdb.ldq) *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
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
    unsigned long long keep = ~0ull;     // the first step's mask word (backward chain)
    if (MODE == 2 && p.step[0].bits_in != nullptr) keep = p.step[0].bits_in[bits_idx];
#pragma unroll
    for (int s = 0; s < BF_MAX_STEPS; ++s)
#pragma unroll
        for (int j = 0; j < BPT; ++j)
            if (s < p.n_steps) bias_lds[s * BF_WIDE + tid + j * 64 * NW] = bv[s][j];
    // everything this lane loaded or stored so far has to be out of the way of the counted waits: drain once, before the loop
    // (the two DMA groups in flight are waited for here too -- the only vmcnt(0) of the kernel, at its very start)
    BF_VMCNT(0);
    __syncthreads();                     // the bias copy is complete before the first accumulator takes its bias
    if (MODE == 2 && K0S == 1 && p.in_mode == 2) {
        // loss partials of this tile's transitions: the first wave of each transition's wave group, lane <-> TD row
        const int wpt = tdb.W >> 4;                             // waves per transition (W = 16, 32 or 64)
        if (wave % wpt == 0) {
            const double* sl = reinterpret_cast<const double*>(ring_lds + 2 * BF_STAGE_BYTES);
            const int tb = row0 / tdb.W + wave / wpt;
            const bool live = lane < tdb.W && tb < tdb.B;
            double s0 = live ? sl[(16 * wave + lane) * 2 + 0] : 0.0, s1 = live ? sl[(16 * wave + lane) * 2 + 1] : 0.0;
            s0 = wave_sum(s0);
            s1 = wave_sum(s1);
            if (lane == 0 && tb < tdb.B) { tdb.loss_part[(size_t)tb * 2 + 0] = s0; tdb.loss_part[(size_t)tb * 2 + 1] = s1; }
        }
    }

    f32x4 acc[16];
    const int n_wide = p.n_steps - (p.head ? 1 : 0);
    // ---- first step ---------------------------------------------------------------------------------------------------------------
    BF_T(0)
    bf_acc_init<16>(acc, bias_lds, q);
    bf_wide_step<NW, K0S, 0, PW>(acc, x, ring, lane);
    BF_T(1)
    bf_wide_epilogue<MODE>(acc, x, p.step[0], p.rows, row, row_ok, bits_idx, q, keep);
    BF_T(2)
    // ---- the 256 x 256 steps --------------------------------------------------------------------------------------------------------
    if constexpr (ROLL) {
        static_assert(!ROLL || (NW == 4 && MODE == 2), "rolling epilogues: the backward chain on 64-row tiles");
        f32x4 cc[2][2];
        bf_u32x4 y[8][3];
        BfRollPend pend;
        // the operand arrays swap roles from step to step (x -> y, y -> x, ...): no copies, and whichever array the register
        // allocator keeps in accumulation registers is read from there by the MFMAs directly
#define BF_ROLL_STEP(S, PEND, E01, A, B)                                                                                             \
        {                                                                                                                            \
            keep_prev = keep;                                                                                                        \
            keep = ~0ull;                                                                                                            \
            if (p.step[S].bits_in != nullptr) keep = p.step[S].bits_in[bits_idx];     /* (one vector-memory instruction: in E01) */  \
            /* (both descriptors from the argument block, here: carried from step to step their pointers ended up in vector   */     \
            /*  registers and every store of the pending pair ran a readfirstlane loop)                                        */     \
            const BfRollOut prev_ = bf_roll_out(p.step[(S) > 1 ? (S) - 1 : 1], p.rows, row, row_ok, q, keep_prev);                   \
            const BfRollOut cur_ = bf_roll_out(p.step[S], p.rows, row, row_ok, q, keep);                                             \
            bf_roll_wide_step<MODE, PEND, E01>(cc, A, B, ring, lane, bias_lds + (S) * BF_WIDE, q, pend, prev_, cur_);                \
            BF_T(3)                                                                                                                  \
        }
        unsigned long long keep_prev = ~0ull;
        if (n_wide > 1) BF_ROLL_STEP(1, false, BF_SAVE_VMEM, x, y)
        int s = 2;
        for (; s + 1 < n_wide; s += 2) {
            BF_ROLL_STEP(s, true, 3, y, x)
            BF_ROLL_STEP(s + 1, true, 3, x, y)
        }
        if (s < n_wide) {
            BF_ROLL_STEP(s, true, 3, y, x)
            bf_roll_pair_epilogue<MODE>(7, pend, bf_roll_out(p.step[n_wide - 1], p.rows, row, row_ok, q, keep), cc[1][0], cc[1][1], x[7]);
        } else if (n_wide > 1) {
            bf_roll_pair_epilogue<MODE>(7, pend, bf_roll_out(p.step[n_wide - 1], p.rows, row, row_ok, q, keep), cc[1][0], cc[1][1], y[7]);
        }
        BF_T(4)
#undef BF_ROLL_STEP
    } else
    for (int s = 1; s < n_wide; ++s) {
        // this step's mask word: the seventeenth vector-memory instruction behind the previous epilogue's sixteen stores, in front of
        // this step's DMA groups -- retired (in issue order) long before the epilogue reads it
        keep = ~0ull;
        if (MODE == 2 && p.step[s].bits_in != nullptr) keep = p.step[s].bits_in[bits_idx];
        bf_acc_init<16>(acc, bias_lds + s * BF_WIDE, q);
        bf_wide_step<NW, 8, MODE != 0 ? BF_SAVE_VMEM : 0, PW>(acc, x, ring, lane);
        BF_T(3)
        bf_wide_epilogue<MODE>(acc, x, p.step[s], p.rows, row, row_ok, bits_idx, q, keep);
        BF_T(4)
    }
    // ---- the head -----------------------------------------------------------------------------------------------------------------
    f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (p.head) {
        const BfStep& st = p.step[p.n_steps - 1];
        bf_acc_init<2>(hacc, bias_lds + (p.n_steps - 1) * BF_WIDE, q);
        if (st.N > 16) bf_head_step<NW, 2, MODE != 0 ? BF_SAVE_VMEM : 0, PW>(hacc, x, ring, lane);
        else bf_head_step<NW, 1, MODE != 0 ? BF_SAVE_VMEM : 0, PW>(hacc, x, ring, lane);
        if (st.out != nullptr && row_ok) {
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * T + 4 * q + r;
                    if (n < st.N) st.out[(size_t)row * st.ldout + n] = hacc[T][r];
                    else if (n < st.ldout) st.out[(size_t)row * st.ldout + n] = 0.f;          // (pad columns of a padded Q buffer)
                }
        }
    }
    // nothing of this workgroup may still be in flight into LDS when the next workgroup of the CU takes the allocation
    BF_VMCNT(0);
    if (MODE == 0 && p.amax) {
        // The tile is 1, 2 or 4 whole transitions (rows b * W + j, j = 0 .. W - 1; W = 16 NW, 8 NW or 4 NW) and the head's
        // accumulators are their online slabs Q(s'_b, a, w_j): arg-max of each transition's W TD rows here, from LDS (the ring is
        // drained: its first stages are free), the waves of a transition working as one group
        __syncthreads();
        const int AR = p.step[p.n_steps - 1].N;
        const int wpt = am_W >> 4;                              // waves per transition
        const int sub = wave / wpt;
        float* region = reinterpret_cast<float*>(ring_lds) + sub * (am_W * 32 + am_W * MORL_MAX_OBJ + 2 * 4 * 64 + 3 * 64);
        EnvArgmaxLds L;
        L.qo = region;
        L.w = L.qo + am_W * 32;                                 // (AR <= 32)
        L.pv = L.w + am_W * MORL_MAX_OBJ;
        L.pc = reinterpret_cast<int*>(L.pv + 4 * 64);
        L.mark = L.pc + 4 * 64;
        L.slot = L.mark + 64;
        L.best = L.slot + 64;
        const int j = 16 * wave + m - sub * am_W;               // this lane's row inside its transition
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * T + 4 * q + r;
                if (n < AR) L.qo[j * AR + n] = hacc[T][r];
            }
        const int b = row0 / am_W + sub;
#define BF_AMAX(NWS) envelope_argmax_tile<NWS>(am_w, am_best, am_pairs, am_slot, am_count, am_epoch, am_B, am_W, am_A, am_R, am_flags & 1, 0, \
                                               (am_flags >> 1) & 1, (am_flags >> 2) & 1, b, L, sub)
        if (wpt == NW) BF_AMAX(NW);
        else if (NW >= 2 && wpt * 2 == NW) BF_AMAX((NW >= 2 ? NW / 2 : 1));
        else BF_AMAX((NW >= 4 ? NW / 4 : 1));
#undef BF_AMAX
    }
    BF_T(5)
#ifdef BF_PROF
    if (prof != nullptr && lane == 0) {
        long long* o = prof + ((size_t)blockIdx.x * NW + wave) * BF_PROF_SLOTS;
        for (int i = 0; i < 6; ++i) o[i] = pt[i];
        o[6] = ring.t_entry; o[7] = ring.t_dma; o[8] = tstart_; o[9] = tp_;
    }
#endif
}

constexpr int BF_LDS_BYTES = BF_RING * BF_STAGE_BYTES + BF_MAX_STEPS * BF_WIDE * 4;

// grid: one workgroup per (16 NW)-row tile over the launch's chains; two workgroups per CU.  NW = 4 (64-row tiles) when that fills
// the chip twice over (the two forward passes of a step: 512 tiles); NW = 2 (32-row tiles, 128 work-items) for launches of fewer
// tiles (the backward chain: 256 64-row tiles would leave ONE workgroup per CU, nothing to cover its barriers and epilogues)
template <int NW, bool PW = false>
__device__ __forceinline__ void mlp_chain_bf_entry(const BfMulti& m, unsigned char* lds) {
    const int tile = (int)blockIdx.x;
    int qn = 0;
    while (qn + 1 < m.n && tile >= m.tile_start[qn + 1]) ++qn;
    const BfChain& p = m.c[qn];
    const int row0 = (tile - m.tile_start[qn]) * (16 * NW);
    float* bias_lds = reinterpret_cast<float*>(lds + BF_RING * BF_STAGE_BYTES);
    // (the arg-max arguments as scalars read HERE: a reference to the argument block handed into the chain body made hipcc copy
    // the whole block to scratch memory, 1.1 KB per work-item)
#define BF_AM m.td.weights, m.td.best_io, m.td.pairs_out, m.td.row_slot, m.td.count, m.td.epoch, m.td.B, m.td.W, m.td.A, m.td.R, \
              (m.td.diag_only | (m.td.fma_scal << 1) | (m.td.bmajor << 2)), m.tdb
    const int mode = p.step[0].bits_in != nullptr ? 2 : (p.step[0].out != nullptr || p.step[0].bits_out != nullptr) ? 1 : 0;
    if (p.k0_steps == 1) {
        if (mode == 2) bf_chain_body<NW, 1, 2, false, PW>(p, row0, lds, bias_lds, m.prof, BF_AM);
        else if (mode == 1) bf_chain_body<NW, 1, 1, false, PW>(p, row0, lds, bias_lds, m.prof, BF_AM);
        else bf_chain_body<NW, 1, 0, false, PW>(p, row0, lds, bias_lds, m.prof, BF_AM);
    } else {
        if (mode == 2) bf_chain_body<NW, 2, 2, false, PW>(p, row0, lds, bias_lds, m.prof, BF_AM);
        else if (mode == 1) bf_chain_body<NW, 2, 1, false, PW>(p, row0, lds, bias_lds, m.prof, BF_AM);
        else bf_chain_body<NW, 2, 0, false, PW>(p, row0, lds, bias_lds, m.prof, BF_AM);
    }
}

__global__ __launch_bounds__(256, 2) void mlp_chain_bf_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    mlp_chain_bf_entry<4>(m, lds);
}

// the backward chain on 64-row tiles with rolling epilogues (one workgroup per CU: ~300 registers per lane); its stream is pair-major
__global__ __launch_bounds__(256, 1) void mlp_chain_bf_roll_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    // (one chain, the backward one, whose first step is the 32-wide head transposed: the host launches nothing else here)
    float* bias_lds = reinterpret_cast<float*>(lds + BF_RING * BF_STAGE_BYTES);
    bf_chain_body<4, 1, 2, true>(m.c[0], (int)blockIdx.x * BF_TM, lds, bias_lds, m.prof, m.td.weights, m.td.best_io, m.td.pairs_out,
                                 m.td.row_slot, m.td.count, m.td.epoch, m.td.B, m.td.W, m.td.A, m.td.R,
                                 (m.td.diag_only | (m.td.fma_scal << 1) | (m.td.bmajor << 2)), m.tdb);
}

// the backward chain on 64-row tiles with a producer wave (320 work-items, one workgroup per CU)
__global__ __launch_bounds__(320, 1) void mlp_chain_bf_pw_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    float* bias_lds = reinterpret_cast<float*>(lds + BF_RING * BF_STAGE_BYTES);
    bf_chain_body<4, 1, 2, false, true>(m.c[0], (int)blockIdx.x * BF_TM, lds, bias_lds, m.prof, m.td.weights, m.td.best_io, m.td.pairs_out,
                                        m.td.row_slot, m.td.count, m.td.epoch, m.td.B, m.td.W, m.td.A, m.td.R,
                                        (m.td.diag_only | (m.td.fma_scal << 1) | (m.td.bmajor << 2)), m.tdb);
}

// the forward launch in its one-round form (at most a 64-row tile per CU: one workgroup per CU, one MFMA wave per SIMD) with the producer
// wave: whichever chains the launch carries (no-grad with the arg-max at the end -- the producer has ended by then: a barrier waits for
// the surviving waves --, training)
__global__ __launch_bounds__(320, 1) void mlp_chain_bf_fwd_pw_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    mlp_chain_bf_entry<4, true>(m, lds);
}

// ... the same on 32-row tiles (forward launches of fewer 64-row tiles than CUs: two MFMA waves + the producer)
__global__ __launch_bounds__(192, 1) void mlp_chain_bf32_fwd_pw_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    mlp_chain_bf_entry<2, true>(m, lds);
}

// ... and on 32-row tiles (launches of fewer 64-row tiles than CUs): two MFMA waves + the producer, 192 work-items -- here every MFMA
// wave issued TWELVE pieces per stage
__global__ __launch_bounds__(192, 1) void mlp_chain_bf32_pw_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    float* bias_lds = reinterpret_cast<float*>(lds + BF_RING * BF_STAGE_BYTES);
    bf_chain_body<2, 1, 2, false, true>(m.c[0], (int)blockIdx.x * 32, lds, bias_lds, m.prof, m.td.weights, m.td.best_io, m.td.pairs_out,
                                        m.td.row_slot, m.td.count, m.td.epoch, m.td.B, m.td.W, m.td.A, m.td.R,
                                        (m.td.diag_only | (m.td.fma_scal << 1) | (m.td.bmajor << 2)), m.tdb);
}

__global__ __launch_bounds__(128, 1) void mlp_chain_bf32_kernel(BfMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS_BYTES];
    kernarg_warm<sizeof(BfMulti)>();
    mlp_chain_bf_entry<2>(m, lds);
}

// ---- the weight stream: split + fragment order, once per optimiser step ---------------------------------------------------------
// One job = one chain step's matrix M[n][k] (n: output feature, k: contraction index) given as base + n * sn + k * sk (forward:
// nn.Linear.weight [out][in], sn = in, sk = 1; backward: the transposed view, sn = 1, sk = in) of N x K real elements, written
// as ksteps * ntiles * 3 blocks from block `block0` on; natural = 1: slot (q, e) of k-step s <-> k = 32 s + 8 q + e (the first step,
// fed from memory), 0: <-> k = 32 s + 16 (e >> 2) + 4 q + (e & 3) (fed from the registers of the step before).
struct BfSplitJob {
    const float* base;
    long long sn, sk;
    int N, K, ksteps, ntiles, natural, block0;
    int pair_major;     // 8 k-steps x 16 tiles only: stage i of the job = k-steps 4 (i & 1) .. + 3 of tiles 2 (i >> 1), + 1 (mlp_chain_bf_roll.h)
};
constexpr int BF_MAX_JOBS = 3 * BF_MAX_STEPS;      // forward + backward stream of the online network, forward stream of the target network
struct BfSplitArgs {
    BfSplitJob job[BF_MAX_JOBS];
    int unit_start[BF_MAX_JOBS + 1];       // (k-step, tile) units of job j: [unit_start[j], unit_start[j + 1])
    int n;
};

// one thread per (k-step, tile, lane): eight weights -> three 16-byte fragments; `unit0` = first unit of this call's grid
__device__ __forceinline__ void bf_split_body(const BfSplitArgs& a, unsigned char* __restrict__ stream, long long gtid) {
    const int unit = (int)(gtid >> 6), lane = (int)(gtid & 63);
    if (unit >= a.unit_start[a.n]) return;
    int j = 0;
    while (j + 1 < a.n && unit >= a.unit_start[j + 1]) ++j;
    const BfSplitJob& jb = a.job[j];
    const int u = unit - a.unit_start[j];
    int s = u / jb.ntiles, t = u - s * jb.ntiles;
    if (jb.pair_major) {
        // unit u of the stream: stage u / 8 = (pair j = u / 16, half h = (u / 8) & 1), slot u % 8 = (k-step 4 h + slot / 2, tile 2 j + (slot & 1))
        s = 4 * ((u >> 3) & 1) + ((u & 7) >> 1);
        t = 2 * (u >> 4) + (u & 1);
    }
    const int n = 16 * t + (lane & 15), q = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = jb.natural ? 32 * s + 8 * q + e : 32 * s + 16 * (e >> 2) + 4 * q + (e & 3);
        v[e] = (n < jb.N && k < jb.K) ? jb.base[(long long)n * jb.sn + (long long)k * jb.sk] : 0.f;
    }
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) bf_split2(v[2 * w], v[2 * w + 1], hi[w], mid[w], lo[w]);
    unsigned char* dst = stream + ((size_t)jb.block0 + (size_t)u * 3) * BF_BLOCK + lane * 16;
    *reinterpret_cast<bf_u32x4*>(dst) = bf_u32x4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<bf_u32x4*>(dst + BF_BLOCK) = bf_u32x4{mid[0], mid[1], mid[2], mid[3]};
    *reinterpret_cast<bf_u32x4*>(dst + 2 * BF_BLOCK) = bf_u32x4{lo[0], lo[1], lo[2], lo[3]};
}

__global__ __launch_bounds__(256) void bf_split_kernel(BfSplitArgs a, unsigned char* __restrict__ stream) {
    bf_split_body(a, stream, (long long)blockIdx.x * 256 + threadIdx.x);
}

}  // namespace morl
