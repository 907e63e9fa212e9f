// Layer-fused MLP engine -- argument structures (the kernels are in mlp_chain2.h).  A workgroup carries a row tile through a
// whole chain of dense layers with the activations RESIDENT IN LDS -- they never round-trip through HBM between layers:
//
//   step s:   out_s[TM][N_s] = epilogue_s( act[TM][K_s] @ Bmat_s[K_s][N_s] ),   act <- out_s
//
// Used three ways (same kernel, different step tables):
//   * no-grad forward      Q(s', w) slabs:  Bmat = W_l^T (K-major shadow copy), epilogue bias+ReLU, only Q leaves
//   * training forward     same, but every hidden activation is also saved to HBM for the backward pass (+ its sign bits)
//   * backward (dX chain)  Bmat = W_l as stored ([out][in] is already K-major for g_l @ W_l), epilogue = ReLU mask from the
//                          sign bits; every g_l is written out for the weight-gradient GEMM
// Steps with N <= 32 (the Q head, N = A*R) are "narrow": the four waves split the contraction instead of the columns and
// read the operand N-major (Bt).  Widths up to 256; Bmat row strides must be multiples of 4 floats and Bmat must be zero in
// columns [N, ldb) (host guarantees both, else the per-layer GEMM path is used).
// Roofline: fp32 MFMA; per row sum_s 2*K_s*N_s flop; weights re-read from L2; algorithmic HBM bytes = inputs + outputs only.
#pragma once
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

constexpr int CH_MAXW = 256;        // widest layer
constexpr int CH_BK = 32;           // K chunk (one register set of B)
constexpr int CH_THREADS = 256;

struct ChainStep {
    const float* Bmat;   // [K][ldb] K-major operand (row k contiguous over n), zero in columns [N, ldb)
    const float* Bt;     // [N][ldbt] the same matrix N-major (row n contiguous over k); needed when N <= 32
    const float* bias;   // [N] or NULL
    const float* mask;   // [rows][ldmask] or NULL: result kept where mask > 0 (ReLU backward)
    const unsigned long long* bits_in;   // or NULL: the same mask as one 64-bit word per work-item (see bits_out)
    unsigned long long* bits_out;        // or NULL: (output > 0) of this wide step, packed per work-item in the
                                         // 64-row tiling: word[(row0/64)*256 + tid], bit (tm*2 + tn)*16 + r
    float* out;          // [rows][ldout] or NULL: global copy of this step's output
    int K, N, ldb, ldbt, ldmask, ldout;
    int relu;
    int kpad;            // mlp_chain2: rows of Bmat physically present (>= K, zero beyond K); 0 = exactly K
    // mlp_chain2, batched chains (ChainArgs::nb > 1: nb networks of one shape in one ChainArgs): floats between consecutive
    // networks' weights (Bmat, Bt, bias), outputs (out) and sign-bit words (bits_in / bits_out)
    long long sW, sOut, sBits;
};

struct ChainArgs {
    ChainStep step[MORL_MAX_LAYERS];
    int n_steps;
    int rows;
    // input assembly
    int in_mode;            // 0: cat(obs[b], weights[k]) with row -> (b, k) by row_order ; 1: dense matrix src[rows][ldsrc] ;
                            // 3 (mlp_chain2 / mlp_chain16 / mlp_chain4): cat(obs[b], weights[j]) of the listed pairs, see rows_dev / pairs
    const float* obs;       // [B][D]
    const float* weights;   // [W][R]  (row_order 2: [rows][R], paired with obs rows)
    int B, W, D, R, row_order;
    const float* src;       // in_mode 1
    int ldsrc, K0;
    float* x0_out;          // in_mode 0, or NULL: the assembled rows also go to HBM as [rows][ldx0] (zero padded) -- the
    int ldx0;               // weight-gradient GEMM of layer 0 reads them; saves the separate input-assembly launch
    long long* prof;        // development probe only (tools/probes): [blocks][8] phase cycle counters
    int nb;                 // mlp_chain2: networks batched in this chain (0 / 1: one); unit u -> network u / units_per_net
    long long sSrc;         //   floats between the input matrices (in_mode 1) of consecutive input groups
    int src_div;            //   networks per input group (twin critics share their input rows)
    // in_mode 3 (the lazily evaluated target rows of an Envelope step): row r is the pair pairs[r] =
    // b * W + j, i.e. cat(obs[b], weights[j]); the row count is what the arg-max launch left in *rows_dev (rows = the upper bound the
    // grid was sized for: tiles beyond the count exit at once)
    const int* rows_dev;
    const int32_t* pairs;
    // in_mode 3: workgroup 0 also reports the count to the host -- *count_mirror = (count_tag << 32) | count, one 8-byte store into
    // host-mapped memory (morl_ctx::lz_mirror: the library sizes the NEXT steps' target launch by what earlier steps selected)
    unsigned long long* count_mirror;
    unsigned int count_tag;
    int fast;               // mlp_chain2: every wide step has ldb == 256 and kpad a multiple of 64 (constant-stride weight stream)
};

// Post-op of a hidden layer of a LayerNorm / Dropout network (common/networks.py:10-48: Linear -> [Dropout] -> [LayerNorm] -> ReLU)
// carried by the 16-row chain (mlp_chain16.h, ChainPostSet): what ac_post_fwd_kernel / ac_post_bwd_kernel of ac_kernels.h do in
// launches of their own, on the tile's rows while they are in LDS.  A 16-row tile holds WHOLE rows (<= 256 columns), so the row
// statistics are one wave's work: wave w takes rows 4w .. 4w+3, lane <-> columns lane + 64 j -- the very arithmetic, in the very
// order, of the per-layer kernels (bit-identical given the same Linear output).
//   forward  (mode 1): z = Linear output (in LDS) -> keep mask, dropped z, mean / rstd, xhat -> HBM (what the backward needs),
//                      h = relu(xhat * gamma + beta) -> LDS (the next step's input; the chain's deferred copy takes it to st.out)
//   backward (mode 2): dh = dLoss/dh (in LDS, the dX step's output) -> HBM copy for the LayerNorm affine gradients
//                      (ac_ln_grad_kernel reads dh, h, xhat), dz = LN' / Dropout' of relu'(dh) -> LDS (next step's input, st.out)
struct ChainPost {
    float* xhat;              // [G][cap][ld]  forward: out; backward: in
    float* rstd;              // [G][cap]
    unsigned long long* mask; // [G][cap][ceil(N / 64)] keep bits (PostArgs::mask of ac_kernels.h), or NULL (no dropout in this pass)
    const uint8_t* ext_mask;  // forward: explicit keep flags [G][rows][N] (parity tests) or NULL -> counter-based RNG
    const float* gamma;       // params + offset of the layer's LayerNorm weight (beta follows at +N), or NULL (no LayerNorm)
    const float* h;           // backward: [G][cap][ld] the forward's post-ReLU activation
    float* dh_out;            // backward: [G][cap][ld] copy of dLoss/dh, or NULL
    unsigned long long seed;  // forward, counter-based RNG
    long long gstride;        // floats between the nets' rows in xhat / h / dh_out (cap * ld)
    int ld;                   // row stride of xhat / h / dh_out
    int active;               // 0: this step has no post-op (the output layer; a backward chain's last step)
    int drop;                 // dropout active in this pass
};
struct ChainPostSet {
    ChainPost st[MORL_MAX_LAYERS];
    long long pstride;        // floats between the nets' parameters
    long long ext_gstride;    // bytes between the nets' explicit masks
    int cap;                  // row capacity of the tape (stride of rstd / mask)
    float drop_p, inv_keep;
};

constexpr int CH_OOB = 0x40000000;   // byte offset beyond any matrix (forces the out-of-range zero of a buffer access)
constexpr int CH_MAX_MULTI = 3;      // independent chains per launch (the three forward passes of an Envelope step)

}  // namespace morl
