// libmorl_hip.so -- C ABI over the gfx950 kernels (see include/morl_hip.h for the contract).
// Host side is plain C++: it validates arguments, owns the scratch workspace and enqueues kernels on the
// caller's stream.  Nothing here synchronises the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <chrono>
#include <thread>
#include <vector>

#include "morl_hip.h"
#include "morl_host.h"
#include "morl_device.h"
#include "gemm_f32.h"
#include "envelope_kernels.h"
#include "mlp_chain.h"
#include "mlp_chain2.h"
#include "mlp_chain16.h"
#include "mlp_chain4.h"
#include "mlp_chain_bf.h"
#include "dw_tiles.h"
#include "dw_bf.h"
#include "mlp_chain_bfn.h"
#include "mlp_chain_bf2.h"
#include "optim_kernels.h"
#include "replay_kernels.h"
#include "pareto_kernels.h"
#include "metrics_kernels.h"

using namespace morl;

// ------------------------------------------------------------------------------------------------
// error plumbing (morl_host.h; the per-thread message buffer lives here)
// ------------------------------------------------------------------------------------------------
thread_local char morl_host::g_err[512] = "";
using morl_host::fail;
using morl_host::g_err;
using morl_host::round_up;
using morl_host::vec_ok;
using morl_host::stream_grid;
using morl_host::dmalloc;

extern "C" const char* morl_last_error(void) { return g_err; }
extern "C" int morl_abi_version(void) { return MORL_ABI_VERSION; }
extern "C" int morl_is_device_build(void) {
#ifdef HIPSIM_EMULATED  // defined only by tests/hipsim/hip/hip_runtime.h (host-emulated test build)
    return 0;
#else
    return 1;
#endif
}

// ------------------------------------------------------------------------------------------------
// kernels that need the tile engine
// ------------------------------------------------------------------------------------------------
namespace morl {

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(GemmProblem g) {
    const int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    gemm_tile<A_KC, B_KC, EPI>(g, id / g.tiles_n, id % g.tiles_n, (int)blockIdx.y);
}

struct GemmGroup {
    GemmProblem p[MORL_MAX_LAYERS];
    int tile_start[MORL_MAX_LAYERS + 1];
    int n;
};

// All weight-gradient GEMMs of one backward pass in a single launch (split-K over the batch rows):
// dW_l[o][i] = sum_m g_l[m][o] * h_l[m][i], db_l[o] = sum_m g_l[m][o].
// Grid: 1-D over (split, tile) pairs, XCD-remapped so that ALL tiles of a K-split -- which share the split's slices of
// dZ_l and h_l -- run on one XCD and hit its L2 (the dispatcher round-robins consecutive workgroups over the 8 XCDs).
__global__ __launch_bounds__(GEMM_THREADS) void gemm_grouped_tn_kernel(GemmGroup grp) {
    const int n_tiles = grp.tile_start[grp.n];
    const int logical = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int split = logical / n_tiles, id = logical % n_tiles;
    int q = 0;
    while (q + 1 < grp.n && id >= grp.tile_start[q + 1]) ++q;
    const int local = id - grp.tile_start[q];
    const GemmProblem& g = grp.p[q];
    gemm_tile<false, false, EPI_STORE>(g, local / g.tiles_n, local % g.tiles_n, split);
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst,
                                                        int ldd, long long rows, int cols) {
    const long long total = rows * cols;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / cols;
        const int c = (int)(e % cols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}

// dst[i * B + b][0 .. cols) = src[b * W + i][0 .. cols): rows kept transition-major by the layer-fused engines, handed out in the
// reference's TD-row order (envelope.py:284-291) -- parity / debug outputs only
__global__ __launch_bounds__(256) void copy_rows_bmajor_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst,
                                                               int ldd, int B, int W, int cols) {
    const long long total = (long long)B * W * cols;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / cols;                 // destination row i * B + b
        const int c = (int)(e % cols);
        const int i = (int)(r / B), b = (int)(r % B);
        dst[r * ldd + c] = src[((long long)b * W + i) * lds + c];
    }
}

// The step's prologue in ONE launch: batch selection + gather + weight upload (workgroups [0, sg_blocks)) beside the K-major
// shadow weights of both networks (the remaining workgroups) -- two independent pieces of work that round 1 and the first
// half of round 2 ran as two launches (and, before that, as five).
__global__ __launch_bounds__(256) void step_prologue_kernel(SampleGatherArgs sg, int sg_blocks, const float* __restrict__ params,
                                                            float* __restrict__ wt, const float* __restrict__ params2,
                                                            float* __restrict__ wt2, ShadowArgs sh, int sh_nets, BfSplitArgs bf,
                                                            unsigned char* __restrict__ bf_stream) {
    __shared__ float tile[SH_T * (SH_T + 1)];
    const int b = (int)blockIdx.x;
    if (b < sg_blocks) { sample_gather_body(sg, b, sg_blocks); return; }
    const int t = b - sg_blocks;
    if (t < sh.tiles) shadow_tiles_body(params, wt, sh, t, sh.tiles, tile);
    else if (t < sh_nets * sh.tiles) shadow_tiles_body(params2, wt2, sh, t - sh.tiles, sh.tiles, tile);
    else bf_split_body(bf, bf_stream, (long long)(t - sh_nets * sh.tiles) * 256 + threadIdx.x);     // (split-bf16 weight stream: mlp_chain_bf.h)
}

}  // namespace morl
static_assert(ST_MAX_B == morl_host::TREE_UPDATE_MAX, "morl_host.h's TREE_UPDATE_MAX mirrors ST_MAX_B");

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
constexpr int LZ_SLOTS = 32;        // mirror slots (a power of two > LZ_LAG)
constexpr int LZ_LAG = 8;           // the target launch of lazily evaluated step e is sized by the count of step e - LZ_LAG

struct morl_ctx {
    morl_net_desc net;
    int L;
    int64_t P;
    int64_t offW[MORL_MAX_LAYERS], offB[MORL_MAX_LAYERS];
    int max_batch, max_weights;
    int64_t max_rows;
    int ld0, ldq, max_h;
    int dw_tiles;
    // workspace (device)
    float* x0n = nullptr;   // [rows][ld0]  cat(next_obs_b, w_j), row b*W+j
    float* x0m = nullptr;   // [rows][ld0]  cat(obs_b, w_i),      row i*B+b
    float* ping = nullptr;  // [rows][max_h] no-grad activations
    float* pong = nullptr;
    float* h[MORL_MAX_LAYERS] = {};   // h[l], l = 1..L-1: saved post-ReLU activations of the training pass
    float* g[MORL_MAX_LAYERS] = {};   // g[l], l = 0..L-2: dLoss/dz_l ; g[L-1] aliases dq
    float* qo = nullptr;    // [rows][A*R]
    float* qt = nullptr;
    float* qm = nullptr;    // [rows][ldq]
    float* dq = nullptr;    // [rows][ldq]
    float* slabs = nullptr; // [max_splits][P]
    int max_splits;
    double* sumsq_part = nullptr;  // [OPT_MAX_BLOCKS]
    double* loss_part = nullptr;   // [max_batch][2]
    // layer-fused path (mlp_chain.h): K-major transposed weight copies, refreshed at the start of every call
    float* wt_online = nullptr;
    float* wt_target = nullptr;
    int64_t wt_count = 0;
    int64_t offWt[MORL_MAX_LAYERS];          // Wt_l [round_up(in, 64)][ldn] (zero rows / columns beyond [in][out])
    int64_t offWb[MORL_MAX_LAYERS];          // row-padded copy of W_l for the backward chain (out % 64 != 0; every layer in the
                                             // K4 layout when k4), or -1
    bool k4 = false;                         // every wide chain step has 256 columns: constant-stride weight stream over the K4
                                             // layout (mlp_chain2.h: c2_load_fast); ChainArgs::fast of this context's chains
    int ldn[MORL_MAX_LAYERS];
    int dw_mode = 3;         // weight-gradient engine: 3 balanced wave-layout tiles (dw_tiles.h: 16-byte aligned operand rows);
                             // 2 the per-layer engine's LDS tiles as one grouped split-K launch (any shape: the fall-back)
    bool fused_ok = false;   // architecture fits the fused engine
    bool use_fused = false;  // fused_ok and not disabled by morl_ctx_set_fused
    unsigned long long* relu_bits[MORL_MAX_LAYERS] = {};  // [l]: (h[l] > 0) packed by the 64-row forward tiling
    bool bits_valid = false;                               // written by the last training forward
    float* zeros = nullptr;  // 16 zero floats: target of the invalid elements of the chain's operand gathers
    unsigned char* per_scratch = nullptr;   // ST_SCRATCH_BYTES: part 1 -> part 2 of a PER tree update split over two launches
    // optional per-launch timing of the dominant kernel (mlp_chain): HIP event pairs on the caller's stream
    bool timing = false;                 // chain launches of the current step are bracketed by events
    int timing_every = 0;                // 0 = off, n = every n-th Envelope step is timed (event records cost ~4 us of stream
    long long timing_step = 0;           //     time each, so timing every step would perturb what it measures)
    int timing_idx = 0, timing_prev_launches = 0, timing_rotate = -1;   // every == -1: one launch per step, taking turns
    std::vector<hipEvent_t> ev_start, ev_stop;
    std::vector<int> ev_kind;            // MORL_TIMED_* of each recorded launch
    int last_B = 0, last_WI = 0;         // shape of the last training forward (morl_ctx_debug_hidden)
    int last_step_W = 0;                 // weights of the last morl_envelope_update (what morl_envelope_prepare expects of the next)
    long long prepare_rows = -1;         // one-shot, set by morl_envelope_rank_step: the TD rows THIS rank's step will run (prepare is
    bool prepare_weight_shard = false;   //   handed the job's whole batch) and whether it is the weight-sharded step (its own threshold)
    // lazy target evaluation (envelope_kernels.h, EnvelopeTdArgs::phase): scratch + the one-shot hand-over from
    // morl_envelope_update to update_core
    int lazy_targets = 1;                // MORL_LAZY_TARGETS=0 / morl_ctx_set_lazy_targets: 0 evaluate the whole target slab instead,
                                         // 1 lazily from MORL_LAZY_MIN_ROWS (8 192) TD rows on, 2 lazily at every size
    int32_t* lz_best = nullptr;          // [max_rows] flattened (j*, a*) per TD row
    int lz_epoch = 0;                    // lazily evaluated steps so far (its parity picks the counter)
    bool lz_last = false;                // the last morl_envelope_update ran lazily
    int32_t* lz_slot = nullptr;          // [max_rows] TD row -> compact row
    int32_t* lz_pairs = nullptr;         // [max_rows] compact row -> pair
    int32_t* lz_count = nullptr;         // [2] distinct pairs of the steps of even / odd epoch
    // Adaptive sizing of the target launch.  The 8-row tiles are the right tool for the FEW rows a step normally selects (1 161 - 1 828
    // of 16 384 at the flagship shape) and the wrong one for a batch whose TD rows select (nearly) all B * W pairs -- there the
    // 64-row tiles of the f32 chain are, on the same compact rows.  Which one a step launches is decided on the host from the count
    // an EARLIER lazily evaluated step reported (LZ_LAG steps back: the host may run that far ahead of the device without waiting),
    // read from a host-mapped mirror the target launch's workgroup 0 writes: slot epoch % LZ_SLOTS = (epoch << 32) | count.  The
    // decision is a function of that count alone, so a run is reproducible whatever the host / device timing.
    unsigned long long* lz_mirror = nullptr;       // host-mapped, LZ_SLOTS entries
    unsigned long long* lz_mirror_dev = nullptr;   // its device address
    long long lz_big_rows = 6144;                  // counts above this take the large tiles (MORL_LAZY_BIG_ROWS; the measured cross-over
                                                   // at the flagship shape: profiles/r05_lazy_target_rows_sweep.json)
    double host_wait_s = 0.0;                      // host time spent waiting for a count (the device more than LZ_LAG steps behind):
                                                   // back-pressure, not host work (morl_ctx_backpressure_seconds)
    int lz_last_big = 0;                           // what the last lazily evaluated step launched
    bool last_dual = false;                        // the step's two online forward passes ran as tile pairs (mlp_chain_bf2.h)
    bool lz_last_bfn = false;                      // ... its target rows ran on the few-row split-bf16 chain (mlp_chain_bfn.h)
    bool lz_count_missed = false;                  // the last lazily evaluated step could not read the count it is sized by (bounded
                                                   // wait ran out, or inside the re-arm window after one): it took the small tiles
    long long lz_count_misses = 0;                 // bounded waits that ran out, ever
    int lz_skip_until = 0;                         // epoch from which the count is asked for again
    int timing_kind_override = -1;       // MORL_TIMED_* of the next bracketed chain launch (-1: by its arguments)
    bool lz_argmax_done = false;         // one-shot: this step's forward launch took the arg-max (mlp_chain_bf.h, BfChain::amax)
    bool td_in_chain_done = false;       // one-shot inside update_core: the backward chain's launch took the TD stage
    bool lz_now = false;                 // this step runs lazily: the three below are what the target launch needs
    const float* lz_params_target = nullptr;
    const float* lz_next_obs = nullptr;
    const float* lz_weights_all = nullptr;   // a weight-sharded step: the job's W_total weight vectors (the selected pairs' j* range over all of them)
    float* td_zero_ptr = nullptr;        // one-shot request of the batch-sharded step to the next TD launch: zero this range ...
    int td_zero_n = 0, td_keep_lo = 0, td_keep_hi = 0;   // ... except [keep_lo, keep_hi) (the rank's own priorities)
    // split-bf16 chain (mlp_chain_bf.h): the online network's weights as fragment-ordered bf16 triples, forward stream then backward
    // stream, re-made once per optimiser step (by morl_envelope_prepare's launch, or by the step itself)
    bool bf_ok = false;                  // the architecture fits: hidden layers of 256, head <= 32 columns, input <= 64
    int bf_mode = 1;                     // 0: MORL_EXACT_F32=1 / morl_ctx_set_exact_f32 -- every GEMM on the f32-input MFMA (rounds 1-3)
    long long bf_min_rows = 4096;        // steps of fewer TD rows stay on the fp32 chains (latency-bound; MORL_BF_MIN_ROWS).  The
                                         // unsharded step (and a batch-sharded rank's) from 4 096 rows on -- 64 transitions x 64 weights:
                                         // 0.1766 -> 0.1694 ms --, the weight-sharded rank step from 8 192 (its slab, training pass
                                         // and target rows are separate launches: 0.2094 -> 0.2206 ms at 4 096;
                                         // profiles/r05_rank_step_thresholds_ab.json)
    bool bf_min_rows_env = false;        // MORL_BF_MIN_ROWS given: one threshold for every step
    unsigned char* bf_stream = nullptr;
    int bf_fwd_blocks = 0, bf_bwd_blocks = 0, bf_k0_steps = 0, bf_head_tiles = 0;
    const float* fresh_bf = nullptr;     // parameters the streams were split from by this step's morl_envelope_prepare (one-shot)
    const float* fresh_bft = nullptr;    // TARGET parameters whose forward stream (third region of bf_stream) that launch also made:
                                         // the lazily evaluated target rows then run on the few-row split-bf16 chain (mlp_chain_bfn.h)
    int bf_roll = 0;                     // MORL_BF_ROLL=1: the backward chain's 64-row launch walks its 256 x 256 steps pair after pair with rolling
                                         //   epilogues (mlp_chain_bf_roll.h); its stream is then split pair-major
    bool split_pair_major = false;       // ... what the NEXT split of the backward stream writes (bf_split_args)
    bool bwd_pair_major = false;         // ... what the backward stream holds now
    const float* bf_split_src = nullptr; // ... the parameters it was split from
    bool last_roll = false;              // the last step's backward chain ran with rolling epilogues
    int bf_pw = 15;                      // MORL_BF_PW: which chain launches with one MFMA wave per SIMD get a PRODUCER wave issuing their weight ring (bit mask, default
                                         //   all): 1 backward on 64-row tiles, one round (mlp_chain_bf_pw_kernel); 2 backward on 32-row tiles (mlp_chain_bf32_pw_kernel);
                                         //   4 forward on 64-row tiles, one round (mlp_chain_bf_fwd_pw_kernel); 8 forward on 32-row tiles (mlp_chain_bf32_fwd_pw_kernel)
    bool last_pw = false, last_fwd_pw = false;
    int bf_t_first = 1;                  // MORL_BF_T_FIRST: the training pass's tiles in front of the no-grad pass's in the forward launch's grid -- 1: in one-round
                                         //   launches (default), 0: never, 2: always
    int bf_dual = 0;                     // MORL_BF_DUAL=1: the two online forward passes as tile pairs sharing the weight fragments (mlp_chain_bf2.h)
    int bfn_eager3 = 1;                  // MORL_BFN_EAGER3=0: the target pass of an eagerly evaluated few-row step as a launch of its own on the
                                         // f32 tiles instead of a third chain of the few-row forward launch (A/B)
    long long bfn_max_rows = 4096;       // chain launches of at most this many rows (over their chains) take the few-row split-bf16 chain
                                         // (mlp_chain_bfn.h: 16-row tiles, the waves split the output features); MORL_BFN_MAX_ROWS, 0 = never
    bool bft_ready = false;              // this step: the target network's forward stream is current (set at the step's entry, dropped by its end)
    int bfn_targets = 0;                 // MORL_BFN_TARGETS=1: the lazily evaluated target rows on the few-row split-bf16 chain instead of
                                         // the 8-row f32 tiles (mlp_chain4.h).  Measured and NOT the default: a tile streams the whole network
                                         // through ONE CU's 64 B/clk vector-memory path, and the split stream is 1.5 x the fp32 bytes --
                                         // 20.6 us against 18.5 at the flagship shape (profiles/r06_target_rows_ab.json)
    bool dw_bf_last = false;             // the last step's weight gradients ran on dw_bf.h
    bool bits_bf = false;                // the last training forward left its sign bits in mlp_chain_bf.h's lane layout
    const unsigned int* skip_flag = nullptr;   // one-shot: the next clip + Adam launch leaves the optimiser state alone if this
                                         // device word is non-zero (a timed-out collective of the single-hop transport)
    int main_rows = -1;                  // rows of a hoisted training forward (morl_envelope_main_forward), -1 = none
    // shadow copies made by morl_envelope_prepare for exactly these parameter buffers; consumed (one-shot) by the step's first
    // library entry, dropped by every optimiser step of the library
    const float* fresh_online = nullptr;
    const float* fresh_target = nullptr;
    const float* bf_stream_src = nullptr;   // parameters the split-bf16 streams were made from by this step's morl_envelope_slab_online
                                         // (one-shot for morl_envelope_main_forward; cleared by every optimiser step of the library)
    const float* wt_online_src = nullptr;   // parameters wt_online was transposed from by this step's morl_envelope_slabs
                                         // (cleared by every optimiser step of the library)
    size_t ev_used = 0;
    int chain_stagger = 3;   // mlp_chain2: job-order staggering of co-resident workgroups (Chain2Multi::stagger)
    unsigned int* cu_tickets = nullptr;
    int fused_tm = 0;        // 0: pick the row tile per launch (>= 2 workgroups per CU when possible), else 64 / 32
    int num_cus = 256;
};


extern "C" int64_t morl_param_count(const morl_net_desc* net) {
    if (!net || net->n_layers < 1 || net->n_layers > MORL_MAX_LAYERS) return -1;
    int64_t p = 0;
    for (int l = 0; l < net->n_layers; ++l) p += (int64_t)net->dims[l + 1] * net->dims[l] + net->dims[l + 1];
    return p;
}

static int validate_net(const morl_net_desc* net) {
    if (!net) return fail(MORL_ERR_ARG, "net is NULL");
    if (net->n_layers < 1 || net->n_layers > MORL_MAX_LAYERS)
        return fail(MORL_ERR_ARG, "n_layers=%d outside [1,%d]", net->n_layers, MORL_MAX_LAYERS);
    if (net->reward_dim < 1 || net->reward_dim > MORL_MAX_OBJ)
        return fail(MORL_ERR_ARG, "reward_dim=%d outside [1,%d]", net->reward_dim, MORL_MAX_OBJ);
    if (net->obs_dim < 1 || net->n_actions < 1) return fail(MORL_ERR_ARG, "obs_dim / n_actions must be >= 1");
    if (net->dims[0] != net->obs_dim + net->reward_dim)
        return fail(MORL_ERR_ARG, "dims[0]=%d != obs_dim+reward_dim=%d", net->dims[0], net->obs_dim + net->reward_dim);
    if (net->dims[net->n_layers] != net->n_actions * net->reward_dim)
        return fail(MORL_ERR_ARG, "dims[last]=%d != n_actions*reward_dim=%d", net->dims[net->n_layers],
                    net->n_actions * net->reward_dim);
    for (int l = 0; l <= net->n_layers; ++l)
        if (net->dims[l] < 1) return fail(MORL_ERR_ARG, "dims[%d]=%d", l, net->dims[l]);
    return MORL_OK;
}

extern "C" int morl_ctx_destroy(morl_ctx* c) {
    if (!c) return MORL_OK;
    float* fl[] = {c->x0n, c->x0m, c->ping, c->pong, c->qo, c->qt, c->qm, c->dq, c->slabs};
    for (float* p : fl)
        if (p) (void)hipFree(p);
    for (int l = 0; l < MORL_MAX_LAYERS; ++l) {
        if (c->h[l]) (void)hipFree(c->h[l]);
        if (c->g[l] && c->g[l] != c->dq) (void)hipFree(c->g[l]);
    }
    if (c->sumsq_part) (void)hipFree(c->sumsq_part);
    if (c->loss_part) (void)hipFree(c->loss_part);
    if (c->lz_best) (void)hipFree(c->lz_best);
    if (c->lz_slot) (void)hipFree(c->lz_slot);
    if (c->lz_pairs) (void)hipFree(c->lz_pairs);
    if (c->lz_count) (void)hipFree(c->lz_count);
    for (hipEvent_t e : c->ev_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_stop) (void)hipEventDestroy(e);
    for (int l = 0; l < MORL_MAX_LAYERS; ++l)
        if (c->relu_bits[l]) (void)hipFree(c->relu_bits[l]);
    if (c->zeros) (void)hipFree(c->zeros);
    if (c->per_scratch) (void)hipFree(c->per_scratch);
    if (c->cu_tickets) (void)hipFree(c->cu_tickets);
    if (c->wt_online) (void)hipFree(c->wt_online);
    if (c->wt_target) (void)hipFree(c->wt_target);
    if (c->bf_stream) (void)hipFree(c->bf_stream);
    if (c->lz_mirror) (void)hipHostFree(c->lz_mirror);
    delete c;
    return MORL_OK;
}


extern "C" int morl_ctx_create(morl_ctx** out, const morl_net_desc* net, int max_batch, int max_weights) {
    if (!out) return fail(MORL_ERR_ARG, "out is NULL");
    *out = nullptr;
    int rc = validate_net(net);
    if (rc) return rc;
    if (max_batch < 1 || max_weights < 1) return fail(MORL_ERR_ARG, "max_batch / max_weights must be >= 1");
    if ((int64_t)max_weights * net->n_actions * net->reward_dim > ENV_MAX_SLAB)
        return fail(MORL_ERR_ARG, "W*A*R=%lld exceeds the LDS slab of the envelope kernel (%d floats)",
                    (long long)max_weights * net->n_actions * net->reward_dim, ENV_MAX_SLAB);
    if (max_weights * net->reward_dim > ENV_MAX_WR) return fail(MORL_ERR_ARG, "W*R exceeds %d", ENV_MAX_WR);
    morl_ctx* c = new (std::nothrow) morl_ctx();
    if (!c) return fail(MORL_ERR_ALLOC, "out of host memory");
    c->net = *net;
    c->L = net->n_layers;
    c->P = morl_param_count(net);
    int64_t off = 0;
    c->max_h = 1;
    c->dw_tiles = 0;
    for (int l = 0; l < c->L; ++l) {
        c->offW[l] = off;
        off += (int64_t)net->dims[l + 1] * net->dims[l];
        c->offB[l] = off;
        off += net->dims[l + 1];
        c->max_h = std::max(c->max_h, net->dims[l + 1]);
        c->dw_tiles += ((net->dims[l + 1] + GEMM_BM - 1) / GEMM_BM) * ((net->dims[l] + GEMM_BN - 1) / GEMM_BN);
    }
    c->max_batch = max_batch;
    c->max_weights = max_weights;
    c->max_rows = (int64_t)max_batch * max_weights;
    c->ld0 = round_up(net->dims[0], 4);
    c->ldq = round_up(net->dims[c->L], 4);
    c->max_splits = 64;
    const size_t rows = (size_t)c->max_rows;
#define ALLOC(field, count)                                                        \
    do {                                                                           \
        rc = dmalloc((void**)&c->field, (size_t)(count) * sizeof(*c->field));      \
        if (rc) { morl_ctx_destroy(c); return rc; }                                \
    } while (0)
    ALLOC(x0n, rows * c->ld0);
    ALLOC(x0m, rows * c->ld0);
    ALLOC(ping, rows * c->max_h);
    ALLOC(pong, rows * c->max_h);
    ALLOC(qo, rows * net->dims[c->L]);
    ALLOC(qt, rows * net->dims[c->L]);
    ALLOC(qm, rows * c->ldq);
    ALLOC(dq, rows * c->ldq);
    for (int l = 1; l < c->L; ++l) ALLOC(h[l], rows * net->dims[l]);
    for (int l = 0; l + 1 < c->L; ++l) ALLOC(g[l], rows * net->dims[l + 1]);
    c->g[c->L - 1] = c->dq;
    ALLOC(slabs, (size_t)c->max_splits * c->P);
    ALLOC(sumsq_part, OPT_MAX_BLOCKS);
    ALLOC(loss_part, std::max((size_t)max_batch * 2 * 4, ((size_t)rows / 16 + 8) * 2));   // per (transition, group) or per 16 rows
    c->fused_ok = true;
    c->wt_count = 0;
    for (int l = 0; l < c->L; ++l) {
        c->ldn[l] = round_up(net->dims[l + 1], 4);
        c->offWt[l] = c->wt_count;
        c->wt_count += (int64_t)round_up(net->dims[l], 64) * c->ldn[l];
        if (net->dims[l] > CH_MAXW || net->dims[l + 1] > CH_MAXW) c->fused_ok = false;
        if (l >= 1 && (net->dims[l] & 3)) c->fused_ok = false;   // 8-byte operand / output pairs need even strides
    }
    // K4 weight stream: every wide step of the forward chain (N = dims[l+1] > 32) and of the backward chain (N = dims[l] > 32,
    // l >= 1) has exactly 256 columns, and the backward chain has no narrow step (it would read the forward copy N-major)
    c->k4 = true;
    for (int l = 0; l < c->L; ++l) {
        if (net->dims[l + 1] > 32 && c->ldn[l] != 256) c->k4 = false;
        if (l >= 1 && net->dims[l] != 256) c->k4 = false;
    }
    for (int l = 0; l < c->L; ++l) {
        c->offWb[l] = -1;
        if (l >= 1 && ((net->dims[l + 1] & 63) || c->k4)) {
            c->offWb[l] = c->wt_count;
            c->wt_count += (int64_t)round_up(net->dims[l + 1], 64) * net->dims[l];
        }
    }
    if (net->dims[c->L] > 32 && (net->dims[c->L] & 3)) c->fused_ok = false;
    for (int l = 0; l < c->L; ++l) {
        if (net->dims[l + 1] <= 32 && (net->dims[l] & 1)) c->fused_ok = false;           // forward narrow step: K = dims[l]
        if (l >= 1 && net->dims[l] <= 32 && (net->dims[l + 1] & 1)) c->fused_ok = false;  // backward narrow step: K = dims[l+1]
    }
    for (int l = 0; l < c->L; ++l) {     // narrow steps contract over a multiple of 4
        if (net->dims[l + 1] <= 32 && (net->dims[l] & 3)) c->fused_ok = false;            // forward narrow step: K = dims[l]
        if (l >= 1 && net->dims[l] <= 32 && (net->dims[l + 1] & 3)) c->fused_ok = false;   // backward narrow step: K = dims[l+1]
    }
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            c->num_cus = prop.multiProcessorCount;
    }
    for (int l = 1; l < c->L; ++l) ALLOC(relu_bits[l], ((size_t)c->max_rows + 63) / 64 * CH_THREADS);
    ALLOC(zeros, 16);
    ALLOC(per_scratch, ST_SCRATCH_BYTES);
    ALLOC(lz_best, rows);
    ALLOC(lz_slot, rows);
    ALLOC(lz_pairs, rows);
    ALLOC(lz_count, 4);
    if (hipMemset(c->lz_count, 0, 16) != hipSuccess) {
        morl_ctx_destroy(c);
        return fail(MORL_ERR_HIP, "hipMemset failed");
    }
    if (const char* e = getenv("MORL_LAZY_TARGETS")) c->lazy_targets = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("MORL_LAZY_BIG_ROWS")) c->lz_big_rows = atoll(e);
    {
        void* hm = nullptr;
        void* dm = nullptr;
        if (hipHostMalloc(&hm, LZ_SLOTS * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dm, hm, 0) != hipSuccess) {
            if (hm) (void)hipHostFree(hm);
            morl_ctx_destroy(c);
            return fail(MORL_ERR_ALLOC, "host-mapped count mirror: hipHostMalloc failed");
        }
        c->lz_mirror = (unsigned long long*)hm;
        c->lz_mirror_dev = (unsigned long long*)dm;
        for (int i = 0; i < LZ_SLOTS; ++i) c->lz_mirror[i] = 0ull;
    }
    ALLOC(cu_tickets, C2_CU_SLOTS);
    if (hipMemsetAsync(c->cu_tickets, 0, C2_CU_SLOTS * sizeof(unsigned int), nullptr) != hipSuccess) { morl_ctx_destroy(c); return fail(MORL_ERR_HIP, "zero-fill failed"); }
    {
        hipError_t ez = hipMemsetAsync(c->zeros, 0, 16 * sizeof(float), nullptr);
        if (ez == hipSuccess) ez = hipStreamSynchronize(nullptr);
        if (ez != hipSuccess) { morl_ctx_destroy(c); return fail(MORL_ERR_HIP, "zero-fill failed: %s", hipGetErrorString(ez)); }
    }
    ALLOC(wt_online, c->wt_count + 4);   // (+4: an 8-byte operand load may touch one float past the last row)
    ALLOC(wt_target, c->wt_count + 4);
    c->use_fused = c->fused_ok;
    // split-bf16 chain: first step <= 64 inputs (1 or 2 k-steps of 32), every hidden layer 256 wide, head <= 32 columns
    c->bf_ok = c->fused_ok && c->L >= 2 && c->L <= BF_MAX_STEPS && net->dims[0] <= 64 && net->dims[c->L] <= 32 && (c->ld0 & 3) == 0;
    for (int l = 1; l < c->L; ++l) c->bf_ok = c->bf_ok && net->dims[l] == BF_WIDE;
    if (c->bf_ok) {
        c->bf_k0_steps = (net->dims[0] + 31) / 32;
        c->bf_head_tiles = (net->dims[c->L] + 15) / 16;
        c->bf_fwd_blocks = c->bf_k0_steps * 48 + (c->L - 2) * 384 + 8 * c->bf_head_tiles * 3;
        c->bf_bwd_blocks = 48 + (c->L - 2) * 384;
        ALLOC(bf_stream, (size_t)(2 * c->bf_fwd_blocks + c->bf_bwd_blocks) * BF_BLOCK);      // online forward | online backward | target forward
    }
    if (const char* e = getenv("MORL_EXACT_F32")) c->bf_mode = atoi(e) != 0 ? 0 : 1;
    if (const char* e = getenv("MORL_BFN_TARGETS")) c->bfn_targets = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("MORL_BFN_MAX_ROWS")) c->bfn_max_rows = atoll(e);
    if (const char* e = getenv("MORL_BF_PW")) c->bf_pw = atoi(e);
    if (const char* e = getenv("MORL_BF_ROLL")) c->bf_roll = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("MORL_BF_T_FIRST")) c->bf_t_first = atoi(e);
    if (const char* e = getenv("MORL_BF_DUAL")) c->bf_dual = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("MORL_BFN_EAGER3")) c->bfn_eager3 = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("MORL_BF_MIN_ROWS")) { c->bf_min_rows = atoll(e); c->bf_min_rows_env = true; }
#undef ALLOC
    *out = c;
    return MORL_OK;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------

template <bool A_KC, bool B_KC, int EPI>
static int launch_gemm(GemmProblem g, hipStream_t s, const char* name) {
    g.tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
    g.tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
    g.a_vec = vec_ok(g.A, g.lda);
    g.b_vec = vec_ok(g.B, g.ldb);
    if (g.k_per_split <= 0) g.k_per_split = round_up(g.K, GEMM_BK);
    const int splits = (g.K + g.k_per_split - 1) / g.k_per_split;
    hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, EPI>), dim3(g.tiles_m * g.tiles_n, splits), dim3(GEMM_THREADS), 0, s, g);
    LAUNCH_CHECK(name);
    return MORL_OK;
}

// Q-network forward over `rows` assembled input rows x0 [rows][ld0].
// save = true keeps the post-ReLU activations in ctx->h[1..L-1] (training pass).
static int net_forward(morl_ctx* c, const float* params, const float* x0, int rows, bool save, float* q_out,
                       int ldq_out, hipStream_t s) {
    const float* in = x0;
    int ld_in = c->ld0;
    for (int l = 0; l < c->L; ++l) {
        const bool last = (l == c->L - 1);
        GemmProblem g{};
        g.A = in;
        g.lda = ld_in;
        g.B = params + c->offW[l];
        g.ldb = c->net.dims[l];
        g.bias = params + c->offB[l];
        g.M = rows;
        g.N = c->net.dims[l + 1];
        g.K = c->net.dims[l];
        float* outp;
        if (last) { outp = q_out; g.ldc = ldq_out; }
        else {
            outp = save ? c->h[l + 1] : ((l & 1) ? c->pong : c->ping);
            g.ldc = c->net.dims[l + 1];
        }
        g.C = outp;
        int rc = last ? launch_gemm<true, true, EPI_BIAS>(g, s, "gemm_fwd_out")
                      : launch_gemm<true, true, EPI_BIAS_RELU>(g, s, "gemm_fwd_hidden");
        if (rc) return rc;
        in = outp;
        ld_in = g.ldc;
    }
    return MORL_OK;
}

static int build_input(const float* obs, const float* weights, float* x0, int B, int W, int D, int R, int ldx,
                       int row_order, hipStream_t s) {
    const long long total = (long long)B * W * ldx;
    hipLaunchKernelGGL(build_input_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, s, obs, weights, x0, B, W, D, R,
                       ldx, row_order);
    LAUNCH_CHECK("build_input");
    return MORL_OK;
}

// ---- layer-fused path --------------------------------------------------------------------------
static ShadowArgs shadow_args(morl_ctx* c) {
    ShadowArgs t{};
    int tiles = 0, n = 0;
    for (int l = 0; l < c->L; ++l) {
        ShadowJob& j = t.job[n++];
        j.src_off = c->offW[l];
        j.dst_off = c->offWt[l];
        j.rows_src = c->net.dims[l + 1];
        j.cols_src = c->net.dims[l];
        j.dst_rows = round_up(c->net.dims[l], 64);
        j.dst_ld = c->ldn[l];
        j.mode = 0;
        j.k4 = (c->k4 && c->net.dims[l + 1] > 32) ? 1 : 0;
        j.tiles_c = (j.dst_ld + SH_T - 1) / SH_T;
        j.tile_start = tiles;
        tiles += (j.dst_rows / SH_T) * j.tiles_c;
    }
    for (int l = 1; l < c->L; ++l) {
        if (c->offWb[l] < 0) continue;
        ShadowJob& j = t.job[n++];
        j.src_off = c->offW[l];
        j.dst_off = c->offWb[l];
        j.rows_src = c->net.dims[l + 1];
        j.cols_src = c->net.dims[l];
        j.dst_rows = round_up(c->net.dims[l + 1], 64);
        j.dst_ld = c->net.dims[l];
        j.mode = 1;
        j.k4 = c->k4 ? 1 : 0;
        j.tiles_c = (j.dst_ld + SH_T - 1) / SH_T;
        j.tile_start = tiles;
        tiles += (j.dst_rows / SH_T) * j.tiles_c;
    }
    t.n = n;
    t.tiles = tiles;
    return t;
}

// `step_entry`: the caller is the gradient step that directly follows morl_envelope_prepare in Envelope.update (the unsharded
// step, the slabs call of a sharded one) -- the ONLY consumers of the shadow copies that launch made.  The copies are keyed on
// the parameter pointer, which does not change when the parameters do (morl_polyak, load_state_dict, copy_ write in place), so
// every other entry point (morl_qnet_forward, greedy actions, a skipped step's successor) refreshes unconditionally and drops
// the flags; morl_ctx_invalidate_shadows drops them explicitly.
static int bf_split_launch(morl_ctx* c, const float* params, hipStream_t s);
static int refresh_transposed(morl_ctx* c, const float* params, float* wt, hipStream_t s, const float* params2 = nullptr,
                              float* wt2 = nullptr, bool step_entry = false) {
    // already made by this step's morl_envelope_prepare for exactly these buffers?  (one-shot)
    const bool have1 = step_entry && wt == c->wt_online && c->fresh_online == params;
    const bool have2 = params2 == nullptr || (step_entry && wt2 == c->wt_target && c->fresh_target == params2);
    c->fresh_online = c->fresh_target = c->fresh_bf = c->fresh_bft = nullptr; c->bft_ready = false;
    if (have1 && have2) return MORL_OK;
    const ShadowArgs t = shadow_args(c);
    if (have1)          // (the prologue made the split-bf16 streams instead of the online copy, or only the target copy is stale)
        hipLaunchKernelGGL(shadow_weights_kernel, dim3(t.tiles, 1), dim3(256), 0, s, params2, wt2, (const float*)nullptr, (float*)nullptr, t);
    else
        hipLaunchKernelGGL(shadow_weights_kernel, dim3(t.tiles, (params2 && !have2) ? 2 : 1), dim3(256), 0, s, params, wt, params2, wt2, t);
    LAUNCH_CHECK("shadow_weights");
    return MORL_OK;
}

// what a gradient step on the bf16 matrix cores streams: the online network's split weights + the K-major copy of the TARGET network
// (its rows run on the fp32 few-row tiles, or its whole slab on the fp32 chain when the caller asks for it)
static bool bf_roll_wanted(const morl_ctx* c, long long rows);
static int bf_split_launch(morl_ctx* c, const float* params, hipStream_t s, bool pair_major);
static int refresh_bf_step(morl_ctx* c, const float* params_online, const float* params_target, hipStream_t s, long long rows_bwd = 0) {
    const bool have_bf = c->fresh_bf == params_online, have_t = c->fresh_target == params_target;
    const bool have_bft = have_bf && c->fresh_bft == params_target;
    c->fresh_online = c->fresh_target = c->fresh_bf = c->fresh_bft = nullptr; c->bft_ready = false;
    c->bft_ready = have_bft;             // (this step's lazily evaluated target rows may take the target network's split stream)
    c->wt_online_src = nullptr;
    int rc;
    if (!have_bf && (rc = bf_split_launch(c, params_online, s, bf_roll_wanted(c, rows_bwd)))) return rc;
    if (!have_t) {
        const ShadowArgs t = shadow_args(c);
        hipLaunchKernelGGL(shadow_weights_kernel, dim3(t.tiles, 1), dim3(256), 0, s, params_target, c->wt_target, (const float*)nullptr,
                           (float*)nullptr, t);
        LAUNCH_CHECK("shadow_weights(target)");
    }
    return MORL_OK;
}

// ---- split-bf16 chain (mlp_chain_bf.h) ------------------------------------------------------------------------------------------
// does a gradient step over `rows` TD rows run its two online forward passes and its backward pass on the bf16 matrix cores?
// few rows: the split-bf16 chain in its few-row form (mlp_chain_bfn.h) -- a rank's share of a sharded step, small batches
static bool bfn_few(const morl_ctx* c, long long rows) {
    return c->bf_ok && c->bf_mode != 0 && c->use_fused && c->fused_tm == 0 && rows >= 1 && rows <= c->bfn_max_rows;
}
static bool bf_wanted(const morl_ctx* c, long long rows, bool weight_shard = false) {
    const long long min_rows = (weight_shard && !c->bf_min_rows_env) ? std::max(c->bf_min_rows, 8192ll) : c->bf_min_rows;
    // (the few-row form serves the unsharded / batch-sharded step; the weight-sharded rank step below its threshold keeps the f32
    // engines: its slab, training pass and TD rows are separate entries that were measured there)
    return c->bf_ok && c->bf_mode != 0 && c->use_fused && c->fused_tm == 0 && (rows >= min_rows || (!weight_shard && bfn_few(c, rows)));
}

// does the backward chain of a step over `rows` TD rows take the rolling-epilogue kernel (64-row tiles, a tile for every CU)?
static bool bf_roll_wanted(const morl_ctx* c, long long rows) {
    return c->bf_roll && rows > 0 && !bfn_few(c, rows) && (rows + BF_TM - 1) / BF_TM >= (long long)c->num_cus;
}

// the split jobs of the online network: forward stream (blocks [0, bf_fwd_blocks)) then backward stream
static BfSplitArgs bf_split_args(const morl_ctx* c, const float* params, const float* params_target = nullptr) {
    BfSplitArgs a{};
    const morl_net_desc& n = c->net;
    const int L = c->L;
    int j = 0, units = 0, block = 0;
    auto add = [&](const float* base, long long sn, long long sk, int N, int K, int ksteps, int ntiles, int natural) {
        BfSplitJob& jb = a.job[j];
        jb.base = base; jb.sn = sn; jb.sk = sk; jb.N = N; jb.K = K; jb.ksteps = ksteps; jb.ntiles = ntiles; jb.natural = natural;
        jb.block0 = block;
        a.unit_start[j++] = units;
        units += ksteps * ntiles;
        block += ksteps * ntiles * 3;
    };
    if (params != nullptr) {
        // forward: M[n][k] = W_l[n][k] (nn.Linear.weight [out][in])
        for (int l = 0; l < L; ++l) {
            const bool head = (l == L - 1);
            add(params + c->offW[l], n.dims[l], 1, n.dims[l + 1], n.dims[l], l == 0 ? c->bf_k0_steps : 8, head ? c->bf_head_tiles : 16, l == 0 ? 1 : 0);
        }
        // backward: step k <-> layer l = L - 1 - k: g_{l-1} = g_l W_l, i.e. M[n = i][k = o] = W_l[o][i]
        for (int l = L - 1; l >= 1; --l) {
            add(params + c->offW[l], 1, n.dims[l], n.dims[l], n.dims[l + 1], l == L - 1 ? 1 : 8, 16, l == L - 1 ? 1 : 0);
            if (l != L - 1 && c->split_pair_major) a.job[j - 1].pair_major = 1;
        }
    }
    block = c->bf_fwd_blocks + c->bf_bwd_blocks;
    // the target network's forward stream, behind both streams of the online one (mlp_chain_bfn.h: the lazily evaluated target rows)
    if (params_target != nullptr)
        for (int l = 0; l < L; ++l) {
            const bool head = (l == L - 1);
            add(params_target + c->offW[l], n.dims[l], 1, n.dims[l + 1], n.dims[l], l == 0 ? c->bf_k0_steps : 8, head ? c->bf_head_tiles : 16, l == 0 ? 1 : 0);
        }
    a.unit_start[j] = units;
    a.n = j;
    return a;
}

static int bf_split_launch(morl_ctx* c, const float* params, hipStream_t s, bool pair_major) {
    c->split_pair_major = pair_major;
    const BfSplitArgs a = bf_split_args(c, params);
    c->bwd_pair_major = pair_major;
    c->bf_split_src = params;
    const int blocks = (a.unit_start[a.n] * 64 + 255) / 256;
    hipLaunchKernelGGL(bf_split_kernel, dim3(blocks), dim3(256), 0, s, a, c->bf_stream);
    LAUNCH_CHECK("bf_split");
    return MORL_OK;
}

// forward chain over rows assembled from (obs, weights), row b * W + i; save => hidden activations to ctx->h[], sign bits, x0
static BfChain bf_forward_chain(morl_ctx* c, const float* params, const float* obs, const float* weights, int B, int W, int rows,
                                bool save, float* q_out, int ldq_out, bool target_stream = false) {
    BfChain a{};
    const int L = c->L;
    // (target_stream: `params` is the TARGET network, whose forward stream is the third region of bf_stream -- mlp_chain_bfn.h only)
    a.stream = c->bf_stream + (target_stream ? (size_t)(c->bf_fwd_blocks + c->bf_bwd_blocks) * BF_BLOCK : 0);
    a.n_steps = L; a.k0_steps = c->bf_k0_steps; a.head = 1;
    a.n_stages = c->bf_fwd_blocks / BF_STAGE_BLOCKS;
    a.rows = rows;
    a.in_mode = 0; a.obs = obs; a.weights = weights;
    a.B = B; a.W = W; a.D = c->net.obs_dim; a.R = c->net.reward_dim; a.row_order = 0;
    for (int l = 0; l < L; ++l) {
        BfStep& st = a.step[l];
        const bool last = (l == L - 1);
        st.bias = params + c->offB[l];
        st.N = c->net.dims[l + 1]; st.K = c->net.dims[l];
        st.relu = last ? 0 : 1;
        if (last) { st.out = q_out; st.ldout = ldq_out; }
        else if (save) { st.out = c->h[l + 1]; st.ldout = c->net.dims[l + 1]; st.bits_out = c->relu_bits[l + 1]; }
    }
    if (save) { a.x0_out = c->x0m; a.ldx0 = c->ld0; }
    return a;
}

// backward chain: g[L-1] = dq -> g[l-1] = (g[l] W_l) * (h[l] > 0), every g[l-1] written to ctx->g[] for the weight gradients
static BfChain bf_backward_chain(morl_ctx* c, int rows) {
    BfChain a{};
    const int L = c->L;
    a.stream = c->bf_stream + (size_t)c->bf_fwd_blocks * BF_BLOCK;
    a.n_steps = L - 1; a.k0_steps = 1; a.head = 0;
    a.n_stages = c->bf_bwd_blocks / BF_STAGE_BLOCKS;
    a.rows = rows;
    a.in_mode = 1; a.src = c->dq; a.ldsrc = c->ldq; a.K0 = c->net.dims[L];
    for (int l = L - 1, k = 0; l >= 1; --l, ++k) {
        BfStep& st = a.step[k];
        st.N = c->net.dims[l]; st.K = c->net.dims[l + 1];
        st.relu = 0;
        st.bits_in = c->relu_bits[l];
        st.out = c->g[l - 1]; st.ldout = c->net.dims[l];
    }
    return a;
}

static int timing_open(morl_ctx* c, int kind, hipStream_t s, int* slot);
static int timing_close(morl_ctx* c, int slot, hipStream_t s);
// row tile of a bf16 chain launch: 64-row tiles (4 waves) when there is at least one for every CU, else 32-row tiles (2 waves).
// (Round 4 first asked for TWO 64-row workgroups per CU; one 4-wave workgroup per CU stages the weight stream into LDS once
// where two 2-wave ones stage it twice: the flagship's backward launch 41.9 -> 39 us.)
constexpr int BF_TILE_FEW = 1;      // bf_tile_rows: the launch takes the few-row chain (mlp_chain_bfn.h), whose tiles are never whole transitions
static int bf_tile_rows(morl_ctx* c, const BfChain* chains, int n) {
    long long tiles64 = 0, rows_all = 0;
    for (int q = 0; q < n; ++q) rows_all += chains[q].rows;
    // the few-row form (32-row tiles whose waves split the output features, no in-chain arg-max / TD stage) while the LAUNCH has at most
    // 4 096 rows over its chains, i.e. 128 tiles: measured per launch (profiles/r06_rank_step_bfn_ab.json) -- the two forward passes of
    // 2 048 rows each 23.8 us against 36 on the 32-row tiles of mlp_chain_bf.h, the backward pass of 4 096 rows 22.2 against 34.3, but
    // the two forward passes of 4 096 rows each 44.0 against 41.1
    if (n <= 2 && bfn_few(c, rows_all)) return BF_TILE_FEW;
    // (three chains: the forward launch of an EAGERLY evaluated few-row step with the target network's pass riding along -- 3 x 2 048
    // rows = 192 tiles, still one per CU)
    if (n == 3 && c->bfn_eager3 && bfn_few(c, rows_all * 2 / 3)) return BF_TILE_FEW;
    for (int q = 0; q < n; ++q) tiles64 += (chains[q].rows + BF_TM - 1) / BF_TM;
    const bool small = tiles64 < (long long)c->num_cus;
    return small ? 32 : BF_TM;
}

static int bf_launch(morl_ctx* c, const BfChain* chains, int n, int kind, hipStream_t s, const EnvelopeTdArgs* td = nullptr,
                     const BfTdArgs* tdb = nullptr) {
    const int tm = bf_tile_rows(c, chains, n);
    // a backward chain whose stream was split pair-major (for the rolling-epilogue kernel) by a launch that guessed this step's rows
    // wrong: the stream again, k-step-major
    const bool backward = n == 1 && chains[0].step[0].bits_in != nullptr;
    const bool roll = backward && c->bwd_pair_major && tm == BF_TM;
    if (backward && c->bwd_pair_major && !roll) {
        if (c->bf_split_src == nullptr) return fail(MORL_ERR_STATE, "internal: pair-major stream without its source");
        int rc2;
        if ((rc2 = bf_split_launch(c, c->bf_split_src, s, false))) return rc2;
    }
    if (tm == BF_TILE_FEW) {
        // 16-row tiles while that is at most a tile per CU, 32-row tiles beyond (mlp_chain_bfn.h)
        long long tiles16 = 0;
        for (int q = 0; q < n; ++q) tiles16 += (chains[q].rows + 15) / 16;
        const int trows = tiles16 <= (long long)c->num_cus ? 16 : BFN_TM;
        if (td || tdb) return fail(MORL_ERR_STATE, "internal: the few-row chain takes no in-chain arg-max / TD stage");
        BfnMulti f{};
        f.n = n;
        int tiles = 0;
        for (int q = 0; q < n; ++q) {
            f.c[q] = chains[q];
            f.c[q].amax = 0;
            f.tile_start[q] = tiles;
            tiles += (chains[q].rows + trows - 1) / trows;
            // which of the three streams the chain reads (their block counts bound the stream's descriptor)
            const unsigned char* st = chains[q].stream;
            f.n_blocks[q] = (st == c->bf_stream + (size_t)c->bf_fwd_blocks * BF_BLOCK) ? c->bf_bwd_blocks : c->bf_fwd_blocks;
        }
        for (int q = n; q <= BFN_MAX_MULTI; ++q) f.tile_start[q] = tiles;
        int slot = -1, rc;
        if ((rc = timing_open(c, kind, s, &slot))) return rc;
        if (trows == 16) hipLaunchKernelGGL(mlp_chain_bfn16_kernel, dim3(tiles), dim3(BFN_THREADS), 0, s, f);
        else hipLaunchKernelGGL(mlp_chain_bfn_kernel, dim3(tiles), dim3(BFN_THREADS), 0, s, f);
        LAUNCH_CHECK("mlp_chain_bfn");
        return timing_close(c, slot, s);
    }
    BfMulti m{};
    m.n = n;
    const bool small = tm == 32;
    if (tdb) m.tdb = *tdb;
    if (td) {
        m.td.weights = td->weights; m.td.best_io = td->best_io; m.td.pairs_out = td->pairs_out; m.td.row_slot = td->row_slot;
        m.td.count = td->count; m.td.epoch = td->epoch; m.td.B = td->B; m.td.W = td->W; m.td.A = td->A; m.td.R = td->R;
        m.td.diag_only = td->diag_only; m.td.i_offset = td->i_offset; m.td.fma_scal = td->fma_scal; m.td.bmajor = td->bmajor;
    }
    int tiles = 0;
    for (int q = 0; q < n; ++q) {
        // (bf_wide_epilogue applies the ReLU of a forward chain's wide steps without asking: every step but the head is a hidden layer)
        if (chains[q].step[0].bits_in == nullptr)
            for (int l = 0; l + (chains[q].head ? 1 : 0) < chains[q].n_steps; ++l)
                if (!chains[q].step[l].relu) return fail(MORL_ERR_STATE, "internal: a wide step of a forward split-bf16 chain without ReLU");
        m.c[q] = chains[q];
        m.tile_start[q] = tiles;
        tiles += (chains[q].rows + tm - 1) / tm;
    }
    for (int q = n; q <= BF_MAX_MULTI; ++q) m.tile_start[q] = tiles;
#ifdef BF_PROF
    // development build: every 50th pair of launches prints the per-wave phase sums (see BF_PROF_SLOTS)
    static long long* prof_dev = nullptr;
    static int prof_calls = 0;
    const bool prof_now = (++prof_calls % 50) <= 1;
    if (!prof_dev) hipMalloc(&prof_dev, (size_t)4096 * 4 * BF_PROF_SLOTS * 8);
    m.prof = prof_dev;
    if (prof_now) hipStreamSynchronize(s);
#endif
    int slot = -1, rc;
    if ((rc = timing_open(c, kind, s, &slot))) return rc;
    // the two online passes of a step as pairs of tiles sharing every weight fragment (mlp_chain_bf2.h): a no-grad chain and a training
    // chain over the same stream, the same number of rows, observation-and-weight input rows both
    const bool dual = c->bf_dual && !small && n == 2 && !tdb && chains[0].stream == chains[1].stream && chains[0].rows == chains[1].rows &&
                      chains[0].n_steps == chains[1].n_steps && chains[0].head && chains[1].head && chains[0].in_mode == 0 &&
                      chains[1].in_mode == 0 && chains[0].step[0].out == nullptr && chains[0].step[0].bits_out == nullptr &&
                      chains[0].step[0].bits_in == nullptr && chains[1].step[0].bits_in == nullptr && chains[0].x0_out == nullptr &&
                      (chains[1].step[0].out != nullptr || chains[1].step[0].bits_out != nullptr);
    if (dual) c->last_dual = true;
    // (one round: the producer form is one 320-work-item workgroup per CU; a launch of more tiles than CUs keeps the two co-resident
    // 256-work-item workgroups of mlp_chain_bf_kernel)
    const bool pw = backward && !roll && (((c->bf_pw & 1) && tm == BF_TM && tiles <= c->num_cus) || ((c->bf_pw & 2) && small));
    const bool fwd_pw = !backward && !dual && !tdb && (((c->bf_pw & 4) && tm == BF_TM && tiles <= c->num_cus) || ((c->bf_pw & 8) && small));
    if (fwd_pw) c->last_fwd_pw = true;
    if (roll) c->last_roll = true;
    if (pw) c->last_pw = true;
    if (dual) hipLaunchKernelGGL(mlp_chain_bf2_kernel, dim3(tiles / 2), dim3(256), 0, s, m);
    else if (roll) hipLaunchKernelGGL(mlp_chain_bf_roll_kernel, dim3(tiles), dim3(256), 0, s, m);
    else if (fwd_pw && small) hipLaunchKernelGGL(mlp_chain_bf32_fwd_pw_kernel, dim3(tiles), dim3(192), 0, s, m);
    else if (fwd_pw) hipLaunchKernelGGL(mlp_chain_bf_fwd_pw_kernel, dim3(tiles), dim3(320), 0, s, m);
    else if (pw && small) hipLaunchKernelGGL(mlp_chain_bf32_pw_kernel, dim3(tiles), dim3(192), 0, s, m);
    else if (pw) hipLaunchKernelGGL(mlp_chain_bf_pw_kernel, dim3(tiles), dim3(320), 0, s, m);
    else if (small) hipLaunchKernelGGL(mlp_chain_bf32_kernel, dim3(tiles), dim3(128), 0, s, m);
    else hipLaunchKernelGGL(mlp_chain_bf_kernel, dim3(tiles), dim3(256), 0, s, m);
    LAUNCH_CHECK("mlp_chain_bf");
#ifdef BF_PROF
    if (prof_now && tiles <= 4096) {
        hipStreamSynchronize(s);
        const int nw = small ? 2 : 4;
        if (dual) tiles /= 2;
        std::vector<long long> h((size_t)tiles * nw * BF_PROF_SLOTS);
        hipMemcpy(h.data(), prof_dev, h.size() * 8, hipMemcpyDeviceToHost);
        double a[BF_PROF_SLOTS] = {0};
        for (int w = 0; w < tiles * nw; ++w)
            for (int i = 0; i < BF_PROF_SLOTS; ++i) a[i] += (double)h[(size_t)w * BF_PROF_SLOTS + i] / (tiles * nw);
        fprintf(stderr, "[bf_prof] %d tiles x %d waves, %s: prologue %.0f | step 0: mfma loop %.0f epilogue %.0f | wide steps: mfma loops %.0f epilogues %.0f | "
                "head + drain %.0f | of which stage-entry waits %.0f, DMA issue %.0f | total %.0f cycles per wave\n",
                tiles, nw, chains[0].step[0].bits_in ? "backward" : "forward", a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[9] - a[8]);
        if (n == 2 && !dual) {
            // the two chains of a forward launch apart (tile_start[1] = first tile of the second), start / end stamps relative to the launch's first
            long long t0 = h[8];
            for (int w = 0; w < tiles * nw; ++w) t0 = std::min(t0, h[(size_t)w * BF_PROF_SLOTS + 8]);
            for (int qn = 0; qn < 2; ++qn) {
                const int w0 = m.tile_start[qn] * nw, w1 = m.tile_start[qn + 1] * nw;
                double b[BF_PROF_SLOTS] = {0}, st = 0, en = 0, en_max = 0;
                for (int w = w0; w < w1; ++w) {
                    for (int i = 0; i < BF_PROF_SLOTS; ++i) b[i] += (double)h[(size_t)w * BF_PROF_SLOTS + i] / (w1 - w0);
                    st += (double)(h[(size_t)w * BF_PROF_SLOTS + 8] - t0) / (w1 - w0);
                    en += (double)(h[(size_t)w * BF_PROF_SLOTS + 9] - t0) / (w1 - w0);
                    en_max = std::max(en_max, (double)(h[(size_t)w * BF_PROF_SLOTS + 9] - t0));
                }
                fprintf(stderr, "[bf_prof]   chain %d: prologue %.0f | step 0 %.0f + %.0f | wide %.0f + %.0f | head + drain %.0f | entry waits %.0f | total %.0f | mean start %.0f, mean end %.0f, last end %.0f\n",
                        qn, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[9] - b[8], st, en, en_max);
            }
        }
    }
#endif
    return timing_close(c, slot, s);
}

// envelope_td_kernel<p.phase>
static int launch_envelope_td(const EnvelopeTdArgs& p, int blocks, int waves, hipStream_t s, const char* what) {
    if (p.phase == 1) hipLaunchKernelGGL(envelope_td_kernel<1>, dim3(blocks), dim3(64 * waves), 0, s, p);
    else if (p.phase == 2) hipLaunchKernelGGL(envelope_td_kernel<2>, dim3(blocks), dim3(64 * waves), 0, s, p);
    else hipLaunchKernelGGL(envelope_td_kernel<0>, dim3(blocks), dim3(64 * waves), 0, s, p);
    LAUNCH_CHECK(what);
    return MORL_OK;
}

// ---- optional event brackets around the GEMM launches of a step (bench.py's roofline figures) --------------------------------
// rotating mode (every == -1): ONE launch per step is bracketed, the k-th timed launch site of the step on step k (mod sites).
static int timing_open(morl_ctx* c, int kind, hipStream_t s, int* slot) {
    *slot = -1;
    const int idx_in_step = c->timing_idx++;
    const bool timed = c->timing && (c->timing_rotate < 0 || idx_in_step == c->timing_rotate);
    if (!timed) return MORL_OK;
    if (c->ev_used == c->ev_start.size()) {
        // no system-scope fence at the record: the fence is not part of the kernel, delays the launch behind it (a
        // bracketed chain launch measured 128.9 us with it, 126.4 us without; rocprofv3's kernel trace says 124.7 us)
        // and costs the step 2 us per record
        const unsigned ev_flags = (unsigned)hipEventDisableSystemFence;
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreateWithFlags(&e0, ev_flags));
        HIP_TRY(hipEventCreateWithFlags(&e1, ev_flags));
        c->ev_start.push_back(e0);
        c->ev_stop.push_back(e1);
        c->ev_kind.push_back(kind);
    }
    *slot = (int)c->ev_used++;
    c->ev_kind[*slot] = kind;
    HIP_TRY(hipEventRecord(c->ev_start[*slot], s));
    return MORL_OK;
}
static int timing_close(morl_ctx* c, int slot, hipStream_t s) {
    if (slot >= 0) HIP_TRY(hipEventRecord(c->ev_stop[slot], s));
    return MORL_OK;
}

// ---- second-generation chain (mlp_chain2.h): one persistent launch of 2 workgroups per CU over all units --------------
// the persistent schedule of `n` chains over S workgroups (0: two per CU); returns S
static int chain2_fill(morl_ctx* c, Chain2Multi& m, const ChainArgs* chains, int n, int S) {
    m.n = n;
    int units = 0;
    for (int q = 0; q < n; ++q) {
        m.p[q] = chains[q];
        m.unit_start[q] = units;
        units += std::max(1, chains[q].nb) * ((chains[q].rows + 63) / 64);
    }
    for (int q = n; q <= CH_MAX_MULTI; ++q) m.unit_start[q] = units;
    if (S <= 0) S = 2 * c->num_cus;      // (measured: 1 / 1.5 / 3 / 4 workgroups per CU are all slower, profiles/r03_knob_sweeps.json)
    // small jobs: no more slots than half units, so that every slot has work
    S = std::max(1, std::min(S, 2 * units));
    m.full_rounds = units / S;
    m.tail_base = m.full_rounds * S;
    m.tail_units = units - m.tail_base;
    m.tail_halves = (2 * m.tail_units <= S) ? 1 : 0;
    if (c->fused_tm == 64) m.tail_halves = 0;
    m.stagger = c->chain_stagger;
    m.cu_tickets = c->cu_tickets;
    return S;
}

static int chain2_launch(morl_ctx* c, const ChainArgs* chains, int n, hipStream_t s) {
    Chain2Multi m{};
    const int S = chain2_fill(c, m, chains, n, 0);
    // (backward chain: the one whose input is the TD kernel's dLoss/dQ)
    const int kind = c->timing_kind_override >= 0 ? c->timing_kind_override
                     : (chains[0].in_mode == 1 && chains[0].src == c->dq) ? MORL_TIMED_BACKWARD
                                                                                                       : MORL_TIMED_FORWARD;
    int slot = -1, rc_t;
    if ((rc_t = timing_open(c, kind, s, &slot))) return rc_t;
    static const bool small_rows = [] { const char* e = getenv("MORL_CHAIN16"); return e ? atoi(e) != 0 : true; }();
    static const bool few_rows = [] { const char* e = getenv("MORL_CHAIN4"); return e ? atoi(e) != 0 : true; }();
    if (small_rows && few_rows && n == 1 && chains[0].rows <= C4_MAX_ROWS && chain4_ok(chains[0])) {
        // one no-grad forward chain over at most a tile per CU (acting, evaluation, greedy actions): 8-row tiles (mlp_chain4.h),
        // bit-identical rows to the 16-row tiles at two thirds of their latency
#ifdef C4_PROF
        // development build (-DC4_PROF): wall_clock64 stamps (10 ns) of every workgroup, printed for a few launches
        static long long* prof_dev = nullptr;
        static int launch_no = 0;
        if (!prof_dev) hipMalloc(&prof_dev, (size_t)(48 * 4096 + 1 + 4096) * 8);
        ChainArgs c4a = chains[0];
        c4a.prof = prof_dev;
        hipLaunchKernelGGL(mlp_chain4_kernel, dim3((chains[0].rows + C4_TM - 1) / C4_TM), dim3(CH_THREADS), 0, s, c4a);
#else
        hipLaunchKernelGGL(mlp_chain4_kernel, dim3((chains[0].rows + C4_TM - 1) / C4_TM), dim3(CH_THREADS), 0, s, chains[0]);
#endif
#ifdef C4_PROF
        if (++launch_no % 100 == 50) {
            const int tiles = (chains[0].rows + C4_TM - 1) / C4_TM;
            hipStreamSynchronize(s);
            std::vector<long long> h((size_t)48 * 4096 + 1 + 4096);
            hipMemcpy(h.data(), prof_dev, h.size() * 8, hipMemcpyDeviceToHost);
            long long t_first = h[48 * 4096 + 1];
            for (int b = 0; b < tiles; ++b) t_first = std::min(t_first, h[48 * 4096 + 1 + b]);
            for (int b : {0, tiles / 2, tiles - 1}) {
                const long long* o = h.data() + (size_t)b * 48;
                fprintf(stderr, "C4_PROF launch %d tile %d/%d: entry +%.2f us;", launch_no, b, tiles, (h[48 * 4096 + 1 + b] - t_first) * 0.01);
                for (int i = 0; i < (int)o[0]; ++i) fprintf(stderr, " %.2f", (o[2 + i] - h[48 * 4096 + 1 + b]) * 0.01);
                fprintf(stderr, " (us after entry: body start, input staged, then per wide step: loop start, loop end, epilogue done, barrier passed; narrow step)\n");
            }
        }
#endif
    } else if (small_rows && chain16_wanted(chains, n)) {
        // few rows (small batches, shards of a strong-scaled job): 16-row tiles, one workgroup each (mlp_chain16.h)
        Chain16Multi m16{};
        const int tiles = chain16_fill(m16, chains, n);
        hipLaunchKernelGGL(mlp_chain16_kernel, dim3(tiles), dim3(CH_THREADS), 0, s, m16);
    } else hipLaunchKernelGGL(mlp_chain2_kernel<1>, dim3(S), dim3(CH_THREADS), 0, s, m);
    LAUNCH_CHECK("mlp_chain2");
    return timing_close(c, slot, s);
}

static int launch_chain(morl_ctx* c, const ChainArgs& a, hipStream_t s) { return chain2_launch(c, &a, 1, s); }

// forward chain over rows assembled on the fly from (obs, weights); save => hidden activations to ctx->h[]
static ChainArgs make_forward_chain(morl_ctx* c, const float* params, const float* wt, const float* obs,
                                    const float* weights, int B, int W, int row_order, int rows, bool save,
                                    float* q_out, int ldq_out, bool emit_bits = false) {
    ChainArgs a{};
    a.n_steps = c->L;
    a.rows = rows;
    a.in_mode = 0;
    a.fast = c->k4 ? 1 : 0;
    emit_bits = true;   // the backward chain takes its ReLU masks as bits only
    a.obs = obs; a.weights = weights;
    a.B = B; a.W = W; a.D = c->net.obs_dim; a.R = c->net.reward_dim; a.row_order = row_order;
    for (int l = 0; l < c->L; ++l) {
        ChainStep& st = a.step[l];
        const bool last = (l == c->L - 1);
        st.Bmat = wt + c->offWt[l];
        st.ldb = c->ldn[l];
        st.Bt = params + c->offW[l];       // nn.Linear layout [out][in] = N-major
        st.ldbt = c->net.dims[l];
        st.K = c->net.dims[l];
        st.kpad = round_up(st.K, 64);
        st.N = c->net.dims[l + 1];
        st.bias = params + c->offB[l];
        st.relu = last ? 0 : 1;
        if (last) { st.out = q_out; st.ldout = ldq_out; }
        else if (save) {
            st.out = c->h[l + 1];
            st.ldout = c->net.dims[l + 1];
            if (emit_bits && st.N > 32) st.bits_out = c->relu_bits[l + 1];
        }
    }
    return a;
}

static int chain_forward(morl_ctx* c, const float* params, const float* wt, const float* obs, const float* weights,
                         int B, int W, int row_order, int rows, bool save, float* q_out, int ldq_out, hipStream_t s) {
    return launch_chain(c, make_forward_chain(c, params, wt, obs, weights, B, W, row_order, rows, save, q_out, ldq_out), s);
}

// the three forward passes of one Envelope step in a single launch of 64-row tiles
static int chain_forward_multi(morl_ctx* c, const ChainArgs* chains, int n, hipStream_t s);
static int chain_forward_x3(morl_ctx* c, const ChainArgs& a0, const ChainArgs& a1, const ChainArgs& a2, hipStream_t s) {
    const ChainArgs a[3] = {a0, a1, a2};
    return chain_forward_multi(c, a, 3, s);
}
// up to CH_MAX_MULTI forward passes in one launch
static int chain_forward_multi(morl_ctx* c, const ChainArgs* chains, int n, hipStream_t s) { return chain2_launch(c, chains, n, s); }

// backward chain: g[L-1] = dq  ->  g[l-1] = (g[l] @ W_l) * (h[l] > 0), every g[l-1] written to ctx->g[]
static int chain_backward(morl_ctx* c, const float* params, int rows, hipStream_t s) {
    const int L = c->L;
    if (L < 2) return MORL_OK;
    int rc_chain;
    ChainArgs a{};
    a.n_steps = L - 1;
    a.rows = rows;
    a.in_mode = 1;
    a.fast = c->k4 ? 1 : 0;
    a.src = c->dq; a.ldsrc = c->ldq; a.K0 = c->net.dims[L];
    for (int l = L - 1, k = 0; l >= 1; --l, ++k) {
        ChainStep& st = a.step[k];
        st.Bmat = params + c->offW[l];
        st.ldb = c->net.dims[l];
        if (c->offWb[l] >= 0) { st.Bmat = c->wt_online + c->offWb[l]; st.kpad = round_up(c->net.dims[l + 1], 64); }
        st.Bt = c->wt_online + c->offWt[l];   // [in][ldn(out)] = N-major for the backward contraction (narrow steps; never K4)
        st.ldbt = c->ldn[l];
        st.K = c->net.dims[l + 1];
        st.N = c->net.dims[l];
        st.mask = c->h[l];
        st.ldmask = c->net.dims[l];
        if (c->bits_valid && st.N > 32) st.bits_in = c->relu_bits[l];
        st.out = c->g[l - 1];
        st.ldout = c->net.dims[l];
    }
    if ((rc_chain = launch_chain(c, a, s))) return rc_chain;
    return MORL_OK;
}

// does a chain launch over `rows` rows take the 16-row tiles (mlp_chain16.h)?  -- the decision chain2_launch makes
static bool chain_rows_take_16(long long rows) {
    static const bool small_rows = [] { const char* e = getenv("MORL_CHAIN16"); return e ? atoi(e) != 0 : true; }();
    ChainArgs probe{};
    probe.rows = (int)std::min<long long>(rows, 0x7fffffff);
    return small_rows && chain16_wanted(&probe, 1);
}

extern "C" int morl_ctx_set_fused(morl_ctx* c, int enable) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    c->use_fused = enable && c->fused_ok;
    c->fused_tm = (enable == 2) ? 64 : (enable == 3) ? 32 : 0;
    c->bits_bf = false;              // (what the last step of the OTHER engine left behind says nothing about the next one)
    c->dw_bf_last = false;
    return c->use_fused ? (enable >= 1 && enable <= 3 ? enable : 1) : 0;
}

extern "C" int morl_ctx_set_dw_mode(morl_ctx* c, int mode) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    if (mode != 2 && mode != 3) return fail(MORL_ERR_ARG, "weight-gradient engine %d: 3 (dw_tiles.h, default) or 2 (generic LDS tiles)", mode);
    c->dw_mode = mode;
    return MORL_OK;
}

extern "C" int morl_ctx_set_timing(morl_ctx* c, int every) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    if (every < -2) return fail(MORL_ERR_ARG, "every < -2");
    c->timing_every = every;
    c->timing_idx = c->timing_prev_launches = 0;
    c->timing_rotate = -1;
    c->timing_step = 0;
    c->timing = false;
    c->ev_used = 0;
    // the first event pairs are made HERE, not lazily inside the region the caller is about to time (a pair costs ~10 us of host
    // time to create; a 20-step run used to create its 20 pairs one per step)
    if (every != 0) {
        const unsigned ev_flags = (unsigned)hipEventDisableSystemFence;
        while (c->ev_start.size() < 64) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreateWithFlags(&e0, ev_flags));
            HIP_TRY(hipEventCreateWithFlags(&e1, ev_flags));
            c->ev_start.push_back(e0);
            c->ev_stop.push_back(e1);
            c->ev_kind.push_back(0);
        }
    }
    return MORL_OK;
}

// called at the first library entry of an Envelope step (morl_envelope_update, or morl_envelope_slabs of a sharded step)
static void timing_begin_step(morl_ctx* c) {
    if (c->timing_every == -1 || c->timing_every == -2) {
        // every step (-1) or every second one (-2), one launch: the launches of a step take turns (two event records instead of
        // two per launch)
        const long long period = c->timing_every == -2 ? 2 : 1;
        if (c->timing_idx > 0) c->timing_prev_launches = c->timing_idx;
        c->timing = (c->timing_step % period) == 0;
        c->timing_rotate = c->timing_prev_launches > 0 ? (int)((c->timing_step / period) % c->timing_prev_launches) : 0;
    } else {
        c->timing = c->timing_every > 0 && (c->timing_step % c->timing_every) == 0;
        c->timing_rotate = -1;
    }
    c->timing_idx = 0;
    ++c->timing_step;
}

// Blocks until the recorded launches finished; per kind (MORL_TIMED_*) their count and summed duration, then clears the record.
extern "C" int morl_ctx_read_timing_kinds(morl_ctx* c, int* n_launches, double* total_ms) {
    if (!c || !n_launches || !total_ms) return fail(MORL_ERR_ARG, "NULL argument");
    for (int k = 0; k < MORL_TIMED_KINDS; ++k) { n_launches[k] = 0; total_ms[k] = 0.0; }
    for (size_t k = 0; k < c->ev_used; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(c->ev_stop[k]));
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_start[k], c->ev_stop[k]));
        const int kind = c->ev_kind[k];
        ++n_launches[kind];
        total_ms[kind] += ms;
    }
    c->ev_used = 0;
    return MORL_OK;
}

// the chain launches only (forward + backward-dX), as before the weight-gradient launch was bracketed too
extern "C" int morl_ctx_read_timing(morl_ctx* c, int* n_launches, double* total_ms) {
    if (!c || !n_launches || !total_ms) return fail(MORL_ERR_ARG, "NULL argument");
    int n[MORL_TIMED_KINDS];
    double ms[MORL_TIMED_KINDS];
    const int rc = morl_ctx_read_timing_kinds(c, n, ms);
    if (rc) return rc;
    *n_launches = n[MORL_TIMED_FORWARD] + n[MORL_TIMED_FORWARD2] + n[MORL_TIMED_BACKWARD];
    *total_ms = ms[MORL_TIMED_FORWARD] + ms[MORL_TIMED_FORWARD2] + ms[MORL_TIMED_BACKWARD];
    return MORL_OK;
}

static int check_bw(const morl_ctx* c, int B, int W) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    if (B < 1 || W < 1) return fail(MORL_ERR_ARG, "B=%d W=%d must be >= 1", B, W);
    if (B > c->max_batch || W > c->max_weights)
        return fail(MORL_ERR_STATE, "B=%d W=%d exceed ctx capacity (%d, %d)", B, W, c->max_batch, c->max_weights);
    return MORL_OK;
}

// ------------------------------------------------------------------------------------------------
// public entry points
// ------------------------------------------------------------------------------------------------
extern "C" int morl_gather_batch(const float* records, int record_floats, int64_t capacity, const int64_t* idx, int B,
                                 int D, int R, int action_dim, float* obs, float* next_obs, float* rewards, float* dones,
                                 float* actions_f, int32_t* actions_i, void* stream) {
    if (!records || !idx || !obs || !next_obs || !rewards || !dones || (!actions_f && !actions_i))
        return fail(MORL_ERR_ARG, "NULL array");
    if (B < 1 || D < 1 || R < 1 || action_dim < 1 || capacity < 1)
        return fail(MORL_ERR_ARG, "bad sizes B=%d D=%d R=%d Ad=%d", B, D, R, action_dim);
    if (record_floats != 2 * D + R + 1 + action_dim)
        return fail(MORL_ERR_ARG, "record_floats=%d != 2*D+R+1+Ad=%d", record_floats, 2 * D + R + 1 + action_dim);
    const int blocks = std::min(1024, (B + 3) / 4);
    hipLaunchKernelGGL(gather_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, records, record_floats,
                       (long long)capacity, idx, B, D, R, action_dim, obs, next_obs, rewards, dones, actions_f, actions_i);
    LAUNCH_CHECK("gather_batch");
    return MORL_OK;
}

static int fill_sample_gather(SampleGatherArgs& a, const double* tree, int n_levels, const double* u01, const int64_t* idx_in,
                              const float* records, int record_floats, int64_t capacity, int B, int D, int R, int action_dim,
                              float* obs, float* next_obs, float* rewards, float* dones, float* actions_f, int32_t* actions_i,
                              int64_t* idx_out, const float* aux_src, float* aux_dst, int aux_floats) {
    if (!records || !obs || !next_obs || !rewards || !dones || (!actions_f && !actions_i)) return fail(MORL_ERR_ARG, "NULL array");
    if (tree ? !u01 : !idx_in) return fail(MORL_ERR_ARG, "sample_gather: uniforms (with a tree) or indices (without) are required");
    if (tree && (n_levels < 1 || n_levels > 40)) return fail(MORL_ERR_ARG, "bad n_levels");
    if (B < 1 || D < 1 || R < 1 || action_dim < 1 || capacity < 1)
        return fail(MORL_ERR_ARG, "bad sizes B=%d D=%d R=%d Ad=%d", B, D, R, action_dim);
    if (record_floats != 2 * D + R + 1 + action_dim)
        return fail(MORL_ERR_ARG, "record_floats=%d != 2*D+R+1+Ad=%d", record_floats, 2 * D + R + 1 + action_dim);
    if (aux_src && (!aux_dst || aux_floats < 1)) return fail(MORL_ERR_ARG, "sample_gather: aux copy without destination / size");
    a.tree = tree; a.u01 = u01; a.idx_in = idx_in; a.records = records;
    a.obs = obs; a.next_obs = next_obs; a.rewards = rewards; a.dones = dones; a.actions_f = actions_f; a.actions_i = actions_i;
    a.idx_out = idx_out; a.aux_src = aux_src; a.aux_dst = aux_dst;
    a.capacity = (long long)capacity; a.n_levels = n_levels; a.record_floats = record_floats;
    a.B = B; a.D = D; a.R = R; a.Ad = action_dim; a.aux_floats = aux_floats;
    return MORL_OK;
}

extern "C" int morl_sample_gather(const double* tree, int n_levels, const double* u01, const int64_t* idx_in,
                                  const float* records, int record_floats, int64_t capacity, int B, int D, int R,
                                  int action_dim, float* obs, float* next_obs, float* rewards, float* dones, float* actions_f,
                                  int32_t* actions_i, int64_t* idx_out, const float* aux_src, float* aux_dst, int aux_floats,
                                  void* stream) {
    SampleGatherArgs a{};
    int rc = fill_sample_gather(a, tree, n_levels, u01, idx_in, records, record_floats, capacity, B, D, R, action_dim, obs, next_obs,
                                rewards, dones, actions_f, actions_i, idx_out, aux_src, aux_dst, aux_floats);
    if (rc) return rc;
    const int blocks = std::min(1024, (B + 3) / 4);
    hipLaunchKernelGGL(sample_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK("sample_gather");
    return MORL_OK;
}

extern "C" int morl_envelope_prepare(morl_ctx* c, const float* params_online, const float* params_target, const double* tree,
                                     int n_levels, const double* u01, const int64_t* idx_in, const float* records,
                                     int record_floats, int64_t capacity, int B, int D, int R, int action_dim, float* obs,
                                     float* next_obs, float* rewards, float* dones, float* actions_f, int32_t* actions_i,
                                     int64_t* idx_out, const float* aux_src, float* aux_dst, int aux_floats, void* stream) {
    if (!c || !params_online || !params_target) return fail(MORL_ERR_ARG, "NULL argument");
    SampleGatherArgs a{};
    int rc = fill_sample_gather(a, tree, n_levels, u01, idx_in, records, record_floats, capacity, B, D, R, action_dim, obs, next_obs,
                                rewards, dones, actions_f, actions_i, idx_out, aux_src, aux_dst, aux_floats);
    if (rc) return rc;
    const int blocks = std::min(1024, (B + 3) / 4);
    if (!c->use_fused) {                                    // per-layer engine: no shadow copies to make
        hipLaunchKernelGGL(sample_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        LAUNCH_CHECK("sample_gather");
        return MORL_OK;
    }
    const ShadowArgs sh = shadow_args(c);
    // what the step that follows will stream: on the bf16 matrix cores (a step of B x max_weights rows qualifies) the online
    // network's split weights + the target network's K-major copy; on the fp32 chains the K-major copies of both.  A guess about
    // the step's weight count that turns out wrong costs that step a launch of its own, nothing else (refresh_*).
    // (the step's weight count is not an argument here: the last step's, or the context's capacity before the first one)
    const long long rows_next = c->prepare_rows >= 0 ? c->prepare_rows : (long long)B * (c->last_step_W > 0 ? c->last_step_W : c->max_weights);
    const bool shard_next = c->prepare_rows >= 0 && c->prepare_weight_shard;
    c->prepare_rows = -1;
    if (bf_wanted(c, rows_next, shard_next)) {
        // the target network's forward stream too: for the target rows of a lazily evaluated step (MORL_BFN_TARGETS=1) or for the target
        // pass of an eagerly evaluated few-row step, which rides in its forward launch (mlp_chain_bfn.h)
        const bool bft = (c->bfn_targets && c->lazy_targets != 0) || (c->bfn_eager3 && bfn_few(c, 2 * rows_next));
        c->split_pair_major = bf_roll_wanted(c, rows_next);
        const BfSplitArgs bf = bf_split_args(c, params_online, bft ? params_target : nullptr);
        c->bwd_pair_major = c->split_pair_major;
        c->bf_split_src = params_online;
        const int bf_blocks = (bf.unit_start[bf.n] * 64 + 255) / 256;
        hipLaunchKernelGGL(step_prologue_kernel, dim3(blocks + sh.tiles + bf_blocks), dim3(256), 0, (hipStream_t)stream, a, blocks,
                           params_target, c->wt_target, (const float*)nullptr, (float*)nullptr, sh, 1, bf, c->bf_stream);
        LAUNCH_CHECK("step_prologue");
        c->fresh_online = nullptr;
        c->fresh_target = params_target;
        c->fresh_bf = params_online;
        c->fresh_bft = bft ? params_target : nullptr;
        return MORL_OK;
    }
    hipLaunchKernelGGL(step_prologue_kernel, dim3(blocks + 2 * sh.tiles), dim3(256), 0, (hipStream_t)stream, a, blocks, params_online,
                       c->wt_online, params_target, c->wt_target, sh, 2, BfSplitArgs{}, (unsigned char*)nullptr);
    LAUNCH_CHECK("step_prologue");
    c->fresh_online = params_online;
    c->fresh_target = params_target;
    c->fresh_bf = c->fresh_bft = nullptr;
    return MORL_OK;
}

// bit 0: the last morl_envelope_update on this context ran its online forward passes and its dX backward pass as split-bf16
// products (mlp_chain_bf.h); bit 1: its weight gradients too (dw_bf.h); bit 2: its lazily evaluated target rows took the large
// f32 tiles (an earlier step had selected more than MORL_LAZY_BIG_ROWS pairs); bit 3: its target launch was sized WITHOUT the count it
// should have read (bounded wait ran out / re-arm window); bit 4: that has happened on this context; bit 5: its target network -- the lazily
// evaluated rows, or the whole pass of an eagerly evaluated few-row step -- ran on the few-row split-bf16 chain (mlp_chain_bfn.h) rather than the f32 tiles; 0: everything on the f32-input MFMA
extern "C" int morl_ctx_last_step_bf16(morl_ctx* c) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    return (c->bits_bf ? 1 : 0) | (c->dw_bf_last ? 2 : 0) | ((c->lz_last && c->lz_last_big) ? 4 : 0) |
           ((c->lz_last && c->lz_count_missed) ? 8 : 0) | (c->lz_count_misses > 0 ? 16 : 0) | (c->lz_last_bfn ? 32 : 0) | (c->last_dual ? 64 : 0) | (c->last_roll ? 128 : 0) | (c->last_pw ? 256 : 0) | (c->last_fwd_pw ? 512 : 0);
}

extern "C" int morl_ctx_set_exact_f32(morl_ctx* c, int enable) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    const int was = c->bf_mode == 0 ? 1 : 0;
    c->bf_mode = enable ? 0 : 1;
    c->fresh_online = c->fresh_target = c->fresh_bf = c->fresh_bft = nullptr; c->bft_ready = false;
    return was;
}

extern "C" int morl_ctx_set_lazy_targets(morl_ctx* c, int enable) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    const int was = c->lazy_targets;
    c->lazy_targets = std::max(0, std::min(2, enable));
    return was;
}

// target rows the last morl_envelope_update evaluated lazily (its distinct (transition, weight) pairs; 0: it ran eagerly); synchronises
extern "C" int morl_ctx_lazy_target_rows(morl_ctx* c, int* rows, void* stream) {
    if (!c || !rows) return fail(MORL_ERR_ARG, "NULL argument");
    int32_t n = 0;
    if (!c->lz_last) { *rows = 0; return MORL_OK; }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(&n, c->lz_count + (c->lz_epoch & 1), sizeof(n), hipMemcpyDeviceToHost));
    *rows = (int)n;
    return MORL_OK;
}

// host seconds this context has spent blocked on the device so far (lazy_count_was_big): a caller that measures its own enqueue
// cost subtracts it
extern "C" int morl_ctx_backpressure_seconds(morl_ctx* c, double* seconds) {
    if (!c || !seconds) return fail(MORL_ERR_ARG, "NULL argument");
    *seconds = c->host_wait_s;
    return MORL_OK;
}

extern "C" int morl_ctx_invalidate_shadows(morl_ctx* c) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    c->fresh_online = c->fresh_target = c->fresh_bf = c->fresh_bft = nullptr; c->bft_ready = false;
    c->wt_online_src = nullptr;
    c->bf_stream_src = nullptr;
    return MORL_OK;
}

// parity-test aid: the post-ReLU activations h_l [rows][dims[l]] the LAST training forward on this context saved (what the
// weight-gradient GEMM reads; h_l > 0 is the ReLU mask the backward pass applied)
extern "C" int morl_ctx_debug_hidden(morl_ctx* c, int layer, int rows, float* out, void* stream) {
    if (!c || !out) return fail(MORL_ERR_ARG, "NULL argument");
    if (layer < 1 || layer >= c->L) return fail(MORL_ERR_ARG, "layer %d outside [1, %d)", layer, c->L);
    if (rows < 1 || rows > c->max_rows) return fail(MORL_ERR_ARG, "rows %d outside [1, %d]", rows, c->max_rows);
    if (c->use_fused) {
        // the layer-fused engines keep the rows transition-major (row b * W + i): hand them out in the reference's order
        if (c->last_B < 1 || c->last_WI < 1 || (long long)c->last_B * c->last_WI != rows)
            return fail(MORL_ERR_STATE, "rows %d != the %d x %d rows of the last training forward", rows, c->last_B, c->last_WI);
        const int cols = c->net.dims[layer];
        hipLaunchKernelGGL(copy_rows_bmajor_kernel, dim3(stream_grid((long long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)c->h[layer], cols, out, cols, c->last_B, c->last_WI, cols);
        LAUNCH_CHECK("copy_rows_bmajor");
        return MORL_OK;
    }
    HIP_TRY(hipMemcpyAsync(out, c->h[layer], (size_t)rows * c->net.dims[layer] * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return MORL_OK;
}

extern "C" int morl_host_device_pointer(void* host_ptr, void** device_ptr) {
    if (!host_ptr || !device_ptr) return fail(MORL_ERR_ARG, "NULL argument");
    HIP_TRY(hipHostGetDevicePointer(device_ptr, host_ptr, 0));
    return MORL_OK;
}

extern "C" int morl_gather_fields(const float* records, int record_floats, int64_t capacity, const int64_t* idx, int B,
                                  int n_fields, const int32_t* offsets, const int32_t* widths, float* const* outs,
                                  void* stream) {
    if (!records || !idx || !offsets || !widths || !outs) return fail(MORL_ERR_ARG, "gather_fields: NULL argument");
    if (B < 1 || capacity < 1 || record_floats < 1) return fail(MORL_ERR_ARG, "gather_fields: bad sizes");
    if (n_fields < 1 || n_fields > GATHER_MAX_FIELDS) return fail(MORL_ERR_ARG, "gather_fields: 1..%d fields", GATHER_MAX_FIELDS);
    GatherFields f{};
    f.n = n_fields;
    for (int k = 0; k < n_fields; ++k) {
        if (offsets[k] < 0 || widths[k] < 1 || offsets[k] + widths[k] > record_floats || !outs[k])
            return fail(MORL_ERR_ARG, "gather_fields: field %d [%d, +%d) outside the record / NULL output", k, offsets[k], widths[k]);
        f.offset[k] = offsets[k]; f.width[k] = widths[k]; f.dst[k] = outs[k];
    }
    hipLaunchKernelGGL(gather_fields_kernel, dim3(std::min(1024, (B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, records,
                       record_floats, (long long)capacity, idx, B, f);
    LAUNCH_CHECK("gather_fields");
    return MORL_OK;
}

extern "C" int morl_qnet_forward(morl_ctx* c, const float* params, const float* obs, const float* weights, int B, int W,
                                 int row_order, float* q_out, void* stream) {
    int rc = check_bw(c, B, W);
    if (rc) return rc;
    if (!params || !obs || !weights || !q_out) return fail(MORL_ERR_ARG, "NULL array");
    if (row_order < 0 || row_order > 2) return fail(MORL_ERR_ARG, "row_order must be 0, 1 or 2");
    if (row_order == 2 && W != 1) return fail(MORL_ERR_ARG, "row_order 2 (paired rows) takes W = 1 and weights [B][R]");
    hipStream_t s = (hipStream_t)stream;
    const int rows = (row_order == 2) ? B : B * W;
    if (c->use_fused) {
        if ((rc = refresh_transposed(c, params, c->wt_online, s))) return rc;
        return chain_forward(c, params, c->wt_online, obs, weights, B, W, row_order, rows, false, q_out,
                             c->net.dims[c->L], s);
    }
    rc = build_input(obs, weights, c->x0n, B, W, c->net.obs_dim, c->net.reward_dim, c->ld0, row_order, s);
    if (rc) return rc;
    return net_forward(c, params, c->x0n, rows, false, q_out, c->net.dims[c->L], s);
}

extern "C" int morl_envelope_reduce(const float* qo, const float* qt, const float* weights, int B, int W, int A, int R,
                                    int diag_only, float* target, int32_t* pref, int32_t* ac, void* stream) {
    if (!qo || !qt || !weights || !target) return fail(MORL_ERR_ARG, "NULL array");
    if (B < 1 || W < 1 || A < 1 || R < 1 || R > MORL_MAX_OBJ) return fail(MORL_ERR_ARG, "bad sizes");
    if ((long long)W * A * R > ENV_MAX_SLAB || W * R > ENV_MAX_WR)
        return fail(MORL_ERR_ARG, "W*A*R=%lld exceeds the LDS slab (%d floats)", (long long)W * A * R, ENV_MAX_SLAB);
    EnvelopeTdArgs p{};
    p.qo = qo; p.qt = qt; p.weights = weights;
    p.target = target; p.pref = pref; p.ac = ac;
    p.B = B; p.W = W; p.A = A; p.R = R; p.diag_only = diag_only;
    return launch_envelope_td(p, B, 4, (hipStream_t)stream, "envelope_td(reduce)");
}

extern "C" int morl_envelope_reduce_rows(const float* qo, const float* qt, const float* row_weights, int n_rows, int W,
                                         int A, int R, float* target, int32_t* pref, int32_t* ac, void* stream) {
    if (!qo || !qt || !row_weights || !target) return fail(MORL_ERR_ARG, "NULL array");
    if (n_rows < 1 || W < 1 || A < 1 || R < 1 || R > MORL_MAX_OBJ) return fail(MORL_ERR_ARG, "bad sizes");
    if ((long long)W * A * R > ENV_MAX_SLAB || W * R > ENV_MAX_WR)
        return fail(MORL_ERR_ARG, "W*A*R=%lld exceeds the LDS slab (%d floats)", (long long)W * A * R, ENV_MAX_SLAB);
    EnvelopeTdArgs p{};
    p.qo = qo; p.qt = qt; p.row_weights = row_weights;
    p.target = target; p.pref = pref; p.ac = ac;
    p.B = n_rows; p.W = W; p.A = A; p.R = R;
    return launch_envelope_td(p, n_rows, 4, (hipStream_t)stream, "envelope_td(reduce_rows)");
}

extern "C" int morl_envelope_greedy_actions(morl_ctx* c, const float* params, const float* obs, const float* w, int n,
                                            int32_t* actions_out, void* stream) {
    if (!c || !params || !obs || !w || !actions_out) return fail(MORL_ERR_ARG, "NULL argument");
    if (n < 1 || (int64_t)n > c->max_rows) return fail(MORL_ERR_STATE, "n=%d outside [1, %lld]", n, (long long)c->max_rows);
    hipStream_t s = (hipStream_t)stream;
    const int A = c->net.n_actions, R = c->net.reward_dim;
    int rc;
    // rows through the network (paired rows: row r = (obs_r, w_r)) into the context's Q buffer
    if (c->use_fused) {
        if ((rc = refresh_transposed(c, params, c->wt_online, s))) return rc;
        c->wt_online_src = nullptr;
        if ((rc = chain_forward(c, params, c->wt_online, obs, w, n, 1, 2, n, false, c->qo, A * R, s))) return rc;
    } else {
        if ((rc = build_input(obs, w, c->x0n, n, 1, c->net.obs_dim, R, c->ld0, 2, s))) return rc;
        if ((rc = net_forward(c, params, c->x0n, n, false, c->qo, A * R, s))) return rc;
    }
    EnvelopeTdArgs p{};
    p.qo = c->qo; p.qt = c->qo; p.row_weights = w;
    p.target = nullptr; p.pref = nullptr; p.ac = actions_out;
    p.B = n; p.W = 1; p.A = A; p.R = R;
    p.fma_scal = 1;
    return launch_envelope_td(p, n, 4, s, "envelope_td(greedy_actions)");
}

// the argument block of envelope_td_kernel for the TD rows of `WI` scalarisation vectors against slabs over `W` candidates
static EnvelopeTdArgs td_args(morl_ctx* c, const morl_update_cfg* cfg, const morl_update_out* out, const int32_t* actions,
                              const float* rewards, const float* dones, const float* weights_i, int WI, const float* qo, const float* qt,
                              int W, int i_offset, long long rows_total, int B, int* td_waves_out) {
    const int R = c->net.reward_dim, A = c->net.n_actions;
    EnvelopeTdArgs p{};
    p.qo = qo; p.qt = qt; p.weights = weights_i; p.q_main = c->qm;
    p.actions = actions; p.rewards = rewards; p.dones = dones;
    p.target = out->target; p.pref = out->pref; p.ac = out->ac;
    p.dq = c->dq; p.loss_part = c->loss_part; p.priority = (i_offset == 0) ? out->priority : nullptr;
    p.priority_clear = (i_offset != 0) ? out->priority : nullptr;
    p.B = B; p.W = W; p.A = A; p.R = R; p.ldq = c->ldq;
    p.WI = WI; p.i_offset = i_offset;
    p.bmajor = c->use_fused ? 1 : 0;
    p.diag_only = cfg->envelope ? 0 : 1;
    p.gamma = cfg->gamma;
    const float lam = cfg->homotopy_lambda > 0.f ? cfg->homotopy_lambda : 0.f;
    p.c_mse = (float)((1.0 - (double)lam) * 2.0 / ((double)rows_total * R));
    p.c_aux = (float)((double)lam * 2.0 / (double)rows_total);
    p.i_groups = std::max(1, std::min(4, (WI + 63) / 64));
    if (cfg->slab_parts > 1) {      // all-gathered slabs read in place: [G][2][B][W/G][A][R] -- or, lazily evaluated, [G][B][W/G][A][R]
        p.part_floats = (W / cfg->slab_parts) * A * R;
        p.part_stride = (cfg->shard_params_target ? 1ll : 2ll) * B * p.part_floats;
    }
    // waves per workgroup ~ candidates per TD row: 4 at the single-GPU 64 x 6, up to 16 when a sharded job reduces over
    // all gathered weights
    const long long n_cand = cfg->envelope ? (long long)W * A : A;
    const int td_waves = n_cand >= 256 ? 16 : (n_cand >= 64 ? 8 : 4);
    *td_waves_out = td_waves;
    return p;
}

// Lazy target evaluation, the part that needs the ONLINE next-state slab only: (1) arg-max, every selected (b, j*) pair takes a
// compact target row; (2) the target network on those rows -- few-row tiles sized for the worst case, tiles beyond the count exit
// at once.  (Measured and dropped in round 3: the same tiles as extra workgroups of the training pass's launch.)
// the arg-max stage's arguments of a lazily evaluated step (starts the step's epoch: its parity picks the pair counter)
static EnvelopeTdArgs lazy_argmax_args(morl_ctx* c, const EnvelopeTdArgs& p) {
    c->lz_epoch = (c->lz_epoch & 0x3fffffff) + 1;
    EnvelopeTdArgs a1 = p;
    a1.phase = 1; a1.best_io = c->lz_best;
    a1.pairs_out = c->lz_pairs; a1.row_slot = c->lz_slot; a1.count = c->lz_count; a1.epoch = c->lz_epoch;
    a1.zero_ptr = nullptr;
    return a1;
}

// did the lazily evaluated step LZ_LAG epochs before the current one select more than lz_big_rows pairs?  Waits (bounded) for that
// step's target launch to have started -- the host is then at most LZ_LAG steps ahead of the device, which it normally is not.
static int chain2_fill(morl_ctx* c, Chain2Multi& m, const ChainArgs* chains, int n, int S);
static bool lazy_count_was_big(morl_ctx* c, long long* count_out = nullptr) {
    if (count_out) *count_out = -1;
    c->lz_count_missed = false;
    const long long e = (long long)c->lz_epoch - LZ_LAG;
    if (e < 1 || !c->lz_mirror || c->lz_big_rows <= 0) return false;
    if (c->lz_skip_until > c->lz_epoch) { c->lz_count_missed = true; return false; }      // (re-armed later, below)
    volatile unsigned long long* slot = c->lz_mirror + (e & (LZ_SLOTS - 1));
    unsigned long long v = *slot;
    if ((unsigned int)(v >> 32) != (unsigned int)e) {
        // (not there yet: the device is more than LZ_LAG steps behind.  ~2 s bound: a step whose target launch never ran -- an error
        // return between the arg-max and it -- must not hang its successors; they fall back to the small tiles.)  The ordinary case
        // is a device a fraction of a step behind the bound: spin for that long, then give the core away between looks.
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (;;) {
            v = *slot;
            if ((unsigned int)(v >> 32) == (unsigned int)e) { seen = true; break; }
            const auto waited = std::chrono::steady_clock::now() - t0;
            if (waited > std::chrono::seconds(2)) break;
            if (waited > std::chrono::microseconds(300)) std::this_thread::sleep_for(std::chrono::microseconds(50));
#if defined(__x86_64__)
            else __builtin_ia32_pause();
#endif
        }
        c->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!seen) {
            // never reported: the stream of that step is not making progress while this call is in flight (a caller that parks its
            // streams behind events of its own, a step that failed between the arg-max and the target launch).  THIS step takes the
            // small tiles (correct at any count, slow only in the worst case) and says so (bit 3 of morl_ctx_last_step_bf16); the
            // context asks again after 4 LZ_LAG steps -- one bounded wait per 32 steps at worst, never a switch thrown for good.
            fprintf(stderr, "[morl] lazy target rows: the pair count of step %lld was never reported; small tiles for the next %d steps\n",
                    e, 4 * LZ_LAG);
            c->lz_skip_until = c->lz_epoch + 4 * LZ_LAG;
            c->lz_count_missed = true;
            ++c->lz_count_misses;
            return false;
        }
    }
    if (count_out) *count_out = (long long)(unsigned int)(v & 0xffffffffull);
    return (long long)(unsigned int)(v & 0xffffffffull) > c->lz_big_rows;
}

static int lazy_phase1(morl_ctx* c, const EnvelopeTdArgs& p, int td_waves, hipStream_t s) {
    int rc;
    // (the arg-max may already have been taken at the end of the forward launch: morl_envelope_update, BfChain::amax)
    const bool argmax_done = c->lz_argmax_done;
    c->lz_argmax_done = false;
    if (!argmax_done) {
        const EnvelopeTdArgs a1 = lazy_argmax_args(c, p);
        if ((rc = launch_envelope_td(a1, p.B * p.i_groups, td_waves, s, "envelope_argmax"))) return rc;
    }
    // (compact rows: at most one per TD row of this call -- a shard's rows select among ALL W candidates but own B * WI rows)
    const int wi = p.WI > 0 ? p.WI : p.W;
    const float* w_all = c->lz_weights_all ? c->lz_weights_all : p.weights;
    c->lz_weights_all = nullptr;
    ChainArgs t = make_forward_chain(c, c->lz_params_target, c->wt_target, c->lz_next_obs, w_all, p.B, p.W, 0, p.B * wi, false,
                                     c->qt, p.A * p.R);
    t.in_mode = 3;
    t.rows_dev = c->lz_count + (c->lz_epoch & 1);
    t.pairs = c->lz_pairs;
    t.count_mirror = c->lz_mirror_dev + (c->lz_epoch & (LZ_SLOTS - 1));
    t.count_tag = (unsigned int)c->lz_epoch;
    static const bool few_rows = [] { const char* e = getenv("MORL_CHAIN4"); return e ? atoi(e) != 0 : true; }();   // (A/B)
    // many pairs LZ_LAG steps ago (an adversarial batch: every TD row its own pair): the same compact rows on the 64-row f32 tiles
    c->lz_last_big = lazy_count_was_big(c) ? 1 : 0;
    const bool bfn = c->bft_ready && c->bfn_targets && c->bf_ok && c->bf_mode == 1 && !c->lz_last_big;
    c->bft_ready = false;
    c->lz_last_bfn = bfn;
    if (bfn) {
        // few-row split-bf16 chain on the target network's forward stream (mlp_chain_bfn.h): 16-row tiles, the four waves of a tile
        // split every layer's output features -- the same six-product arithmetic as the step's online passes
        BfnMulti m{};
        BfChain& a = m.c[0];
        a.stream = c->bf_stream + (size_t)(c->bf_fwd_blocks + c->bf_bwd_blocks) * BF_BLOCK;
        a.n_steps = c->L; a.k0_steps = c->bf_k0_steps; a.head = 1;
        a.n_stages = 0;
        a.rows = p.B * wi;
        a.in_mode = 3; a.obs = c->lz_next_obs; a.weights = w_all;
        a.B = p.B; a.W = p.W; a.D = c->net.obs_dim; a.R = c->net.reward_dim; a.row_order = 0;
        a.rows_dev = t.rows_dev; a.pairs = t.pairs; a.count_mirror = t.count_mirror; a.count_tag = t.count_tag;
        for (int l = 0; l < c->L; ++l) {
            BfStep& st = a.step[l];
            st.bias = c->lz_params_target + c->offB[l];
            st.N = c->net.dims[l + 1]; st.K = c->net.dims[l];
            st.relu = (l == c->L - 1) ? 0 : 1;
            if (l == c->L - 1) { st.out = c->qt; st.ldout = p.A * p.R; }
        }
        m.n = 1;
        m.tile_start[0] = 0;
        // (16-row tiles: the grid is sized for the worst case -- every TD row its own pair --, the tiles beyond the count exit at once;
        // the rows a step really selects, 1 200 - 3 200, are at most a tile per CU that way)
        m.tile_start[1] = m.tile_start[2] = (a.rows + 15) / 16;
        for (int q = 2; q <= BFN_MAX_MULTI; ++q) m.tile_start[q] = m.tile_start[1];
        m.n_blocks[0] = c->bf_fwd_blocks;
        hipLaunchKernelGGL(mlp_chain_bfn16_kernel, dim3(m.tile_start[1]), dim3(BFN_THREADS), 0, s, m);
        LAUNCH_CHECK("mlp_chain_bfn(lazy targets)");
    } else if (c->lz_last_big) {
        Chain2Multi m{};
        const int S = chain2_fill(c, m, &t, 1, 0);
        hipLaunchKernelGGL(mlp_chain2_kernel<1>, dim3(S), dim3(CH_THREADS), 0, s, m);
        LAUNCH_CHECK("mlp_chain2(lazy targets, many rows)");
    } else if (few_rows && chain4_ok(t)) {
        // 8-row tiles (mlp_chain4.h): twice the workgroups, half the MFMA time per CU and layer
        // (sized for the worst case -- every TD row its own pair --, tiles beyond the count exit at once.  A grid fitted to the count an
        // earlier step reported, the kernel walking further tiles itself, changed nothing: 24.9 us either way, profiles/r05_target_grid_ab.txt)
        hipLaunchKernelGGL(mlp_chain4_kernel, dim3((p.B * wi + C4_TM - 1) / C4_TM), dim3(CH_THREADS), 0, s, t);
        LAUNCH_CHECK("mlp_chain4(lazy targets)");
    } else {
        Chain16Multi m16{};
        const int tiles = chain16_fill(m16, &t, 1);
        hipLaunchKernelGGL(mlp_chain16_kernel, dim3(tiles), dim3(CH_THREADS), 0, s, m16);
        LAUNCH_CHECK("mlp_chain16(lazy targets)");
    }
    return MORL_OK;
}

// ---- one gradient step, in three stages so that a weight-sharded job can put its collectives between them ----------
// stage B: TD rows of `WI` scalarisation vectors (weights_i) against slabs over all `W` candidates:
//          training forward (unless the caller already ran it), envelope arg-max + TD, backward, weight gradients.
//          Everything is normalised by the GLOBAL row count rows_total = B * W_total, so summing the gradients (and
//          the loss) of all shards gives exactly the single-GPU values.  Gradients are left UNCLIPPED in `grads`.
static int update_core(morl_ctx* c, const float* params_online, float* grads, const float* obs, const int32_t* actions,
                       const float* rewards, const float* dones, const float* weights_i, int WI, const float* qo,
                       const float* qt, int W, int i_offset, long long rows_total, int B, const morl_update_cfg* cfg,
                       const morl_update_out* out, bool main_fwd_done, int* splits_out, int* td_groups_out,
                       hipStream_t s) {
    int rc;
    const morl_net_desc& n = c->net;
    const int D = n.obs_dim, R = n.reward_dim, A = n.n_actions, L = c->L;
    const int rows = B * WI;
    // the dW GEMM reads the training pass's layer-0 input from HBM (the W-tiled batch of envelope.py:284-291 is never
    // materialised wider than this [rows][D+R] block)
    // (the fused three-pass launch has already written it from its own input assembly)
    // internal TD-row order: the layer-fused engines keep the rows of a transition together (row b * WI + i), the per-layer
    // engine the reference's i * B + b; everything between the training forward and the weight gradients only needs the SAME
    // order everywhere (a sum over rows), the parity outputs are handed out in reference order
    const int bmajor = c->use_fused ? 1 : 0;
    const int main_ro = bmajor ? 0 : 1;
    c->last_B = B; c->last_WI = WI;
    if (!main_fwd_done && (rc = build_input(obs, weights_i, c->x0m, B, WI, D, R, c->ld0, main_ro, s))) return rc;
    if (!main_fwd_done) {
        c->bits_valid = false;
        if (c->use_fused) {
            if ((rc = chain_forward(c, params_online, c->wt_online, obs, weights_i, B, WI, main_ro, rows, true, c->qm, c->ldq, s))) return rc;
            c->bits_valid = true;   // (the layer-fused forward always emits the sign bits)
            c->bits_bf = false;
        } else {
            if ((rc = net_forward(c, params_online, c->x0m, rows, true, c->qm, c->ldq, s))) return rc;
            c->bits_bf = false;     // (the per-layer engine: no sign bits at all -- and its weight gradients stay on the exact engines)
        }
    }
    // envelope arg-max + TD target + dLoss/dQ: envelope_td_kernel in front of the backward pass, one lane per TD row of a
    // transition, 64 rows per workgroup pass.  (The same stage INSIDE the backward chain's launch was built and measured in round 3:
    // break-even; removed in round 4.)
    int td_groups = std::max(1, std::min(4, (WI + 63) / 64));
    int n_loss = B * td_groups;
    {
        const bool lazy = c->lz_now;
        c->lz_now = false;
        int td_waves = 0;
        EnvelopeTdArgs p = td_args(c, cfg, out, actions, rewards, dones, weights_i, WI, qo, qt, W, i_offset, rows_total, B, &td_waves);
        p.zero_ptr = c->td_zero_ptr; p.zero_n = c->td_zero_n; p.keep_lo = c->td_keep_lo; p.keep_hi = c->td_keep_hi;
        c->td_zero_ptr = nullptr;
        bool td_in_chain = false;
        if (lazy) {
            // 1. + 2. arg-max on the online slab, the target network on the selected rows
            if ((rc = lazy_phase1(c, p, td_waves, s))) return rc;
            // 3. TD target, loss gradient, priorities from the compact target rows
            p.phase = 2; p.best_io = c->lz_best; p.row_slot = c->lz_slot; p.qt = c->qt;
            // ... by the backward chain's own workgroups when their row tiles are whole transitions and nobody asked for the parity
            // outputs (mlp_chain_bf.h, BfTdArgs): no TD launch
            static const bool td_env = [] { const char* e = getenv("MORL_TD_IN_CHAIN"); return e ? atoi(e) != 0 : true; }();   // (A/B)
            if (td_env && c->use_fused && c->bits_bf && L >= 2 && bmajor && WI == W && i_offset == 0 && td_groups == 1 && !p.target && !p.pref &&
                !p.ac && !p.zero_ptr && !p.priority_clear && p.part_floats == 0 && !p.diag_only && p.q_main != nullptr && (c->ldq & 3) == 0) {
                const BfChain probe = bf_backward_chain(c, rows);
                const int tr = bf_tile_rows(c, &probe, 1);
                td_in_chain = tr != BF_TILE_FEW && (W == tr || 2 * W == tr || (4 * W == tr && tr == 64)) && (W & 15) == 0;
            }
        }
        if (!td_in_chain && (rc = launch_envelope_td(p, B * td_groups, td_waves, s, "envelope_td"))) return rc;
        if (td_in_chain) {
            BfChain bwd = bf_backward_chain(c, rows);
            bwd.in_mode = 2;
            BfTdArgs t{};
            t.best_io = p.best_io; t.row_slot = p.row_slot; t.qt = p.qt; t.q_main = p.q_main; t.actions = p.actions; t.rewards = p.rewards;
            t.dones = p.dones; t.weights = p.weights; t.dq = p.dq; t.loss_part = p.loss_part; t.priority = p.priority;
            t.B = B; t.W = W; t.A = p.A; t.R = p.R; t.ldq = p.ldq; t.gamma = p.gamma; t.c_mse = p.c_mse; t.c_aux = p.c_aux;
            if ((rc = bf_launch(c, &bwd, 1, MORL_TIMED_BACKWARD, s, nullptr, &t))) return rc;
            c->td_in_chain_done = true;
        }
    }
    // backward through the hidden layers: g[l-1] = (g[l] @ W_l) * (h[l] > 0)
    if (c->td_in_chain_done) {
        c->td_in_chain_done = false;         // (the backward chain has run: it took the TD stage with it)
    } else if (c->use_fused && c->bits_bf && L >= 2) {
        // (the training forward ran on the bf16 matrix cores and left its sign bits in that kernel's lane layout)
        const BfChain bwd = bf_backward_chain(c, rows);
        if ((rc = bf_launch(c, &bwd, 1, MORL_TIMED_BACKWARD, s))) return rc;
    } else if (c->use_fused) {
        if ((rc = chain_backward(c, params_online, rows, s))) return rc;
    } else
        for (int l = L - 1; l >= 1; --l) {
            GemmProblem g{};
            g.A = c->g[l];
            g.lda = (l == L - 1) ? c->ldq : n.dims[l + 1];
            g.B = params_online + c->offW[l];
            g.ldb = n.dims[l];
            g.C = c->g[l - 1];
            g.ldc = n.dims[l];
            g.mask = c->h[l];
            g.ldmask = n.dims[l];
            g.M = rows; g.N = n.dims[l]; g.K = n.dims[l + 1];
            if ((rc = launch_gemm<true, false, EPI_RELU_MASK>(g, s, "gemm_dx"))) return rc;
        }
    // all dW / db of the step in one split-K launch
    int splits;
    bool per_done = false;
    Dw2Ranges ranges{};
    bool dw2_ok = (c->ld0 & 3) == 0 && (c->ldq & 3) == 0;      // dw_tiles.h streams 16-byte pieces of 16-byte aligned rows
    for (int l = 1; l < L; ++l) dw2_ok = dw2_ok && (n.dims[l] & 3) == 0;
    // the step runs on the bf16 matrix cores (its forward / backward chains did): the weight gradients too, as six split-bf16
    // products per fp32 product (dw_bf.h) -- when every problem fits one of that kernel's three layouts
    bool dwb_ok = c->use_fused && c->bits_bf && c->bf_mode == 1 && dw2_ok && c->dw_mode == 3;
    constexpr bool dwb_env = true;      // (the A/B against dw_tiles.h is morl_ctx_set_exact_f32 / MORL_EXACT_F32=1: profiles/r04_dw_bf_versions.txt)
    c->dw_bf_last = dwb_ok && dwb_env;
    if (dwb_ok && dwb_env) {
        DwbArgs a{};
        a.n = L;
        a.rows = rows;
        a.slab_stride = c->P;
        for (int l = 0; l < L; ++l) {
            DwbProblem& q = a.p[l];
            q.G = c->g[l];
            q.ldg = (l == L - 1) ? c->ldq : n.dims[l + 1];
            q.H = (l == 0) ? c->x0m : c->h[l];
            q.ldh = (l == 0) ? c->ld0 : n.dims[l];
            q.C = c->slabs + c->offW[l];
            q.ldc = n.dims[l];
            q.colsum = c->slabs + c->offB[l];
            q.M = n.dims[l + 1]; q.N = n.dims[l];
            q.gcols = q.ldg; q.hcols = q.ldh;      // (pad columns of dq / x0 are written as zeros by their producers)
            // the kernel's three compile-time shapes (operand tiles beyond the matrix are zero fragments): a narrow output (the Q
            // head), a narrow input (the first layer), or 128 x 128 blocks
            q.shape = (q.M <= 32) ? 2 : (q.N <= 64) ? 1 : 0;
            q.mgroups = (q.M + 16 * DWB_TG[q.shape] - 1) / (16 * DWB_TG[q.shape]);
            q.ngroups = (q.N + 16 * DWB_TH[q.shape] - 1) / (16 * DWB_TH[q.shape]);
        }
        const bool per_rides = cfg->per_tree && i_offset == 0 && out->priority && B <= ST_MAX_B;
        // One 768-work-item workgroup per CU (96 KB of LDS, 3 waves per SIMD), ONE round: row slices sized so that every job costs
        // the same and there are at most as many jobs as CUs -- one CU fewer when the step's PER tree update rides along as an extra
        // workgroup: it is 17 us of serial tree levels, hidden only if it starts with the jobs (as block 257 of 256 it started when
        // the first job ended: + 6.7 us).  Cost of a 32-row chunk by shape (cycles, profiles/r04_dw_bf_probe.txt): 2 630 / 2 080 /
        // 2 100 -- the producers' load -> split -> write turn-around bounds all three, the 128 x 128 shape also waits for its 96
        // MFMAs per consumer -- i.e. 5 : 4 : 4.
        static const int shape_cost[3] = {5, 4, 4};
        const int target = std::max(1, c->num_cus - (per_rides ? 1 : 0));
        const int chunks_total = (rows + DWB_BK - 1) / DWB_BK;
        int jobs = 0;
        for (int budget = 5;; ++budget) {           // cost budget of a job, in fifths of a 128 x 128 chunk
            jobs = 0;
            bool fits = true;
            for (int l = 0; l < L; ++l) {
                DwbProblem& q = a.p[l];
                const int k = std::max(1, budget / shape_cost[q.shape]);          // chunks per job
                q.k_per_split = k * DWB_BK;
                q.splits = (chunks_total + k - 1) / k;
                if (q.splits > c->max_splits) fits = false;
                jobs += q.splits * q.mgroups * q.ngroups;
            }
            if ((fits && jobs <= target) || budget >= 5 * chunks_total) break;   // (one job per (problem, group): nothing left to merge)
        }
        int r = 0;
        jobs = 0;
        splits = 0;
        for (int l = 0; l < L; ++l) {
            DwbProblem& q = a.p[l];
            q.job_start = jobs;
            jobs += q.splits * q.mgroups * q.ngroups;
            splits = std::max(splits, q.splits);
            ranges.end[r] = c->offB[l];                                   ranges.splits[r++] = q.splits;   // W_l
            ranges.end[r] = c->offB[l] + n.dims[l + 1];                   ranges.splits[r++] = q.splits;   // b_l
        }
        ranges.n = r;
        a.jobs = jobs;
        int extra = 0;
        if (per_rides) {                            // the step's PER update rides along
            a.per.tree = cfg->per_tree; a.per.idx = cfg->per_idx; a.per.raw = out->priority;
            a.per.running_max = cfg->per_running_max; a.per.pr_out = nullptr;
            a.per.n_levels = cfg->per_levels; a.per.B = B; a.per.alpha = cfg->per_alpha;
            extra = 1;
            per_done = true;
        }
        int tslot = -1;
        if ((rc = timing_open(c, MORL_TIMED_DW, s, &tslot))) return rc;
        hipLaunchKernelGGL(dw_bf_kernel, dim3(jobs + extra), dim3(DWB_THREADS), 0, s, a);
        LAUNCH_CHECK("dw_bf");
        if ((rc = timing_close(c, tslot, s))) return rc;
    } else if (c->dw_mode == 3 && dw2_ok) {
        // dw_tiles.h: per-problem wave layout; a layout with fewer MFMAs per contraction step gets longer row slices so that
        // every workgroup carries the same matrix-core work, and the whole launch is one round of ~2 workgroups per CU
        Dw2Args a{};
        a.n = L;
        a.rows = rows;
        a.slab_stride = c->P;
        static const int lay_bm[3] = {128, 128, 32}, lay_bn[3] = {128, 64, 128}, lay_cost[3] = {4, 2, 1};
        double unit_tiles = 0.0;            // tiles in units of the 2x2 layout's cost
        for (int l = 0; l < L; ++l) {
            Dw2Problem& q = a.p[l];
            q.G = c->g[l];
            q.ldg = (l == L - 1) ? c->ldq : n.dims[l + 1];
            q.H = (l == 0) ? c->x0m : c->h[l];
            q.ldh = (l == 0) ? c->ld0 : n.dims[l];
            q.C = c->slabs + c->offW[l];
            q.ldc = n.dims[l];
            q.colsum = c->slabs + c->offB[l];
            q.M = n.dims[l + 1]; q.N = n.dims[l];
            q.layout = (q.M <= 32) ? 2 : (q.N <= 64) ? 1 : 0;
            q.tiles_m = (q.M + lay_bm[q.layout] - 1) / lay_bm[q.layout];
            q.tiles_n = (q.N + lay_bn[q.layout] - 1) / lay_bn[q.layout];
            q.gcols = q.ldg;                 // (pad columns of dq / x0 are written as zeros by their producers)
            q.hcols = q.ldh;
            q.c_vec2 = ((((uintptr_t)q.C) & 7u) == 0 && (q.ldc & 1) == 0 && (c->P & 1) == 0) ? 1 : 0;
            unit_tiles += (double)q.tiles_m * q.tiles_n * lay_cost[q.layout] / 4.0;
        }
        int target = 2 * c->num_cus;
        // base slice of the 2x2 layout: unit_tiles * rows / base ~ target jobs, a multiple of the 32-row chunk
        int base = round_up(std::max(1, (int)std::ceil(unit_tiles * rows / (double)target)), DW2_BK);
        for (;;) {      // the split count of every problem must fit the slab buffer
            bool ok = true;
            for (int l = 0; l < L; ++l) {
                const int kps = base * 4 / lay_cost[a.p[l].layout];
                if ((rows + kps - 1) / kps > c->max_splits) ok = false;
            }
            if (ok) break;
            base += DW2_BK;
        }
        int jobs = 0, r = 0;
        splits = 0;
        for (int l = 0; l < L; ++l) {
            Dw2Problem& q = a.p[l];
            q.k_per_split = base * 4 / lay_cost[q.layout];
            q.splits = (rows + q.k_per_split - 1) / q.k_per_split;
            q.job_start = jobs;
            jobs += q.splits * q.tiles_m * q.tiles_n;
            splits = std::max(splits, q.splits);
            ranges.end[r] = c->offB[l];                                   ranges.splits[r++] = q.splits;   // W_l
            ranges.end[r] = c->offB[l] + n.dims[l + 1];                   ranges.splits[r++] = q.splits;   // b_l
        }
        ranges.n = r;
        a.jobs = jobs;
        int extra = 0;
        if (cfg->per_tree && i_offset == 0 && out->priority && B <= ST_MAX_B) {     // the step's PER update rides along
            a.per.tree = cfg->per_tree; a.per.idx = cfg->per_idx; a.per.raw = out->priority;
            a.per.running_max = cfg->per_running_max; a.per.pr_out = nullptr;
            a.per.n_levels = cfg->per_levels; a.per.B = B; a.per.alpha = cfg->per_alpha;
            extra = 1;
            per_done = true;
        }
        a.stagger = 0;
        int tslot = -1;
        if ((rc = timing_open(c, MORL_TIMED_DW, s, &tslot))) return rc;
        hipLaunchKernelGGL(dw_tiles_kernel, dim3(jobs + extra), dim3(DW2_THREADS), 0, s, a);
        LAUNCH_CHECK("dw_tiles");
        if ((rc = timing_close(c, tslot, s))) return rc;
    } else {
    // the generic engine (operand rows that are not 16-byte aligned, or morl_ctx_set_dw_mode(ctx, 2)): one workgroup per CU
    splits = std::max(1, std::min(c->max_splits, (c->num_cus + c->dw_tiles - 1) / c->dw_tiles));
    int kps = round_up((rows + splits - 1) / splits, GEMM_BK);
    splits = (rows + kps - 1) / kps;
    {
        GemmGroup grp{};
        grp.n = L;
        int t = 0;
        for (int l = 0; l < L; ++l) {
            GemmProblem& g = grp.p[l];
            g.A = c->g[l];
            g.lda = (l == L - 1) ? c->ldq : n.dims[l + 1];
            g.B = (l == 0) ? c->x0m : c->h[l];
            g.ldb = (l == 0) ? c->ld0 : n.dims[l];
            g.C = c->slabs + c->offW[l];
            g.ldc = n.dims[l];
            g.colsum = c->slabs + c->offB[l];
            g.M = n.dims[l + 1]; g.N = n.dims[l]; g.K = rows;
            g.k_per_split = kps;
            g.c_split_stride = c->P;
            g.colsum_stride = c->P;
            g.tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
            g.tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
            g.a_vec = vec_ok(g.A, g.lda);
            g.b_vec = vec_ok(g.B, g.ldb);
            grp.tile_start[l] = t;
            t += g.tiles_m * g.tiles_n;
        }
        grp.tile_start[L] = t;
        hipLaunchKernelGGL(gemm_grouped_tn_kernel, dim3(t * splits), dim3(GEMM_THREADS), 0, s, grp);
        LAUNCH_CHECK("gemm_grouped_dw");
    }
    }
    // reduce the split-K slabs into the caller's grad buffer (+ norm partials, this shard's part of the loss)
    const int nblk = std::min(OPT_MAX_BLOCKS, stream_grid(c->P, OPT_THREADS));
    {
        const float lam = cfg->homotopy_lambda > 0.f ? cfg->homotopy_lambda : 0.f;
        if (ranges.n > 0)
            hipLaunchKernelGGL(grad_reduce_ranges_kernel, dim3(nblk), dim3(OPT_THREADS), 0, s, (const float*)c->slabs, ranges,
                               (long long)c->P, grads, (long long)c->P, c->sumsq_part, (const double*)c->loss_part,
                               n_loss, 1.0 / ((double)rows_total * R), 1.0 / (double)rows_total, lam, out->loss);
        else
            hipLaunchKernelGGL(grad_reduce_kernel, dim3(nblk), dim3(OPT_THREADS), 0, s, (const float*)c->slabs, splits,
                               (long long)c->P, grads, (long long)c->P, c->sumsq_part, (const double*)c->loss_part,
                               n_loss, 1.0 / ((double)rows_total * R), 1.0 / (double)rows_total, lam, out->loss);
        LAUNCH_CHECK("grad_reduce");
    }
    if (cfg->per_tree && i_offset == 0 && out->priority && !per_done) {
        if (B > ST_MAX_B) return fail(MORL_ERR_ARG, "PER update inside the step: B=%d > %d", B, ST_MAX_B);
        SumTreeUpdate u{};
        u.tree = cfg->per_tree; u.idx = cfg->per_idx; u.raw = out->priority; u.running_max = cfg->per_running_max;
        u.n_levels = cfg->per_levels; u.B = B; u.alpha = cfg->per_alpha;
        hipLaunchKernelGGL(sumtree_update_kernel, dim3(1), dim3(ST_THREADS), 0, s, u);
        LAUNCH_CHECK("sumtree_update(step)");
    }
    if (out->q_values) {
        const int AR = A * R;
        if (bmajor)
            hipLaunchKernelGGL(copy_rows_bmajor_kernel, dim3(stream_grid((long long)rows * AR, 256)), dim3(256), 0, s,
                               (const float*)c->qm, c->ldq, out->q_values, AR, B, WI, AR);
        else
            hipLaunchKernelGGL(copy_rows_kernel, dim3(stream_grid((long long)rows * AR, 256)), dim3(256), 0, s,
                               (const float*)c->qm, c->ldq, out->q_values, AR, (long long)rows, AR);
        LAUNCH_CHECK("copy_rows");
    }
    if (splits_out) *splits_out = splits;
    if (td_groups_out) *td_groups_out = td_groups;
    return MORL_OK;
}

// stage C: clip_grad_norm_ + Adam on flat buffers.  have_partials: the sum-of-squares partials of grad_reduce are valid
// for `grads` (single GPU); otherwise (gradients were all-reduced) they are recomputed from `grads` first.
// clip + Adam with the step's PER tree update as one extra workgroup of the same launch (the sharded step: the priorities
// are complete only after the all-reduce, so the update cannot ride in the weight-gradient launch as it does on one GPU;
// its 17 us of serial tree levels run beside Adam's 7 instead of behind them)
// (ST_THREADS = 1024 work-items per workgroup: the tree update walks its levels one wave per level)
// the sum of squares of the (all-reduced) gradient with part 1 of the step's PER tree update as one extra workgroup
static __global__ __launch_bounds__(OPT_THREADS) void grad_sumsq_per_kernel(const float* __restrict__ grads, long long P,
                                                                     double* __restrict__ sumsq_part, SumTreeUpdate per,
                                                                     void* __restrict__ per_scratch,
                                                                     const unsigned int* __restrict__ skip_flag) {
    __shared__ __attribute__((aligned(8))) unsigned char lds[ST_LDS_BYTES];
    if (blockIdx.x + 1 == gridDim.x) {
        if (skip_flag != nullptr && *skip_flag != 0u) return;     // (the summed priorities are garbage: see clip_adam_body)
        sumtree_update_part1(per, lds, per_scratch);
        return;
    }
    double ss = 0.0;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)(gridDim.x - 1) * blockDim.x) {
        const float g = grads[p];
        ss += (double)g * (double)g;
    }
    // (the partial of a block in grad_reduce_kernel's order: lanes, then waves)
    double* s_red = reinterpret_cast<double*>(lds);
    ss = wave_sum(ss);
    if (lane_id() == 0) s_red[wave_id()] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < OPT_THREADS / 64; ++w) t += s_red[w];
        sumsq_part[blockIdx.x] = t;
    }
}

static __global__ __launch_bounds__(ST_THREADS) void clip_adam_per_kernel(float* __restrict__ params, float* __restrict__ grads,
                                                                    float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                                    long long P, const double* __restrict__ sumsq_part,
                                                                    int n_part, float max_norm, float one_minus_b1, float b2,
                                                                    float one_minus_b2, float neg_step_size, float bc2_sqrt,
                                                                    float eps, int apply_step, float* __restrict__ grad_norm_out,
                                                                    SumTreeUpdate per, const void* __restrict__ per_scratch,
                                                                    const unsigned int* __restrict__ skip_flag) {
    __shared__ __attribute__((aligned(8))) unsigned char lds[ST_LDS_BYTES];
    if (blockIdx.x + 1 == gridDim.x) {
        if (skip_flag != nullptr && *skip_flag != 0u) return;     // (the summed priorities are garbage too: see clip_adam_body)
        // (per_scratch: part 1 ran as the extra workgroup of the launch in front -- grad_sumsq_per_kernel --, this is part 2)
        if (per_scratch != nullptr) sumtree_update_part2(per, lds, per_scratch);
        else sumtree_update_body(per, lds);
        return;
    }
    clip_adam_body(params, grads, exp_avg, exp_avg_sq, P, sumsq_part, n_part, max_norm, one_minus_b1, b2, one_minus_b2,
                   neg_step_size, bc2_sqrt, eps, apply_step, grad_norm_out, (int)gridDim.x - 1, skip_flag);
}

static int clip_adam_step(morl_ctx* c, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                          const morl_update_cfg* cfg, float* grad_norm_out, bool have_partials, hipStream_t s,
                          const SumTreeUpdate* per = nullptr) {
    c->wt_online_src = nullptr;          // the parameters change: any transposed copy is stale from here on
    c->bf_stream_src = nullptr;
    c->fresh_online = c->fresh_target = c->fresh_bf = c->fresh_bft = nullptr; c->bft_ready = false;
    const unsigned int* skip_flag = c->skip_flag;    // one-shot request of a sharded step (set right before this call)
    c->skip_flag = nullptr;
    const int nblk = std::min(OPT_MAX_BLOCKS, stream_grid(c->P, OPT_THREADS));
    // the PER tree update of a sharded step over BOTH launches that follow the all-reduce (replay_kernels.h: sumtree_update_part1 / 2)
    static const bool per_split_env = [] { const char* e = getenv("MORL_PER_SPLIT"); return e ? atoi(e) != 0 : true; }();   // (A/B)
    const bool per_split = per != nullptr && !have_partials && per_split_env && c->per_scratch != nullptr;
    if (per_split) {
        hipLaunchKernelGGL(grad_sumsq_per_kernel, dim3(nblk + 1), dim3(OPT_THREADS), 0, s, (const float*)grads, (long long)c->P, c->sumsq_part,
                           *per, c->per_scratch, skip_flag);
        LAUNCH_CHECK("grad_sumsq_per");
    } else if (!have_partials) {
        hipLaunchKernelGGL(grad_reduce_kernel, dim3(nblk), dim3(OPT_THREADS), 0, s, (const float*)grads, 1, (long long)c->P,
                           grads, (long long)c->P, c->sumsq_part, (const double*)nullptr, 0, 0.0, 0.0, 0.f, (float*)nullptr);
        LAUNCH_CHECK("grad_sumsq");
    }
    const double b1 = cfg->beta1, b2 = cfg->beta2;
    const int t = std::max(1, cfg->adam_step);
    const double bc1 = 1.0 - std::pow(b1, (double)t);
    const double bc2 = 1.0 - std::pow(b2, (double)t);
    const double step_size = (double)cfg->lr / bc1;
    const double bc2_sqrt = std::sqrt(bc2);
    if (per != nullptr)
        hipLaunchKernelGGL(clip_adam_per_kernel, dim3(stream_grid(c->P, ST_THREADS) + 1), dim3(ST_THREADS), 0, s, params, grads, exp_avg, exp_avg_sq,
                           (long long)c->P, (const double*)c->sumsq_part, nblk, cfg->max_grad_norm, (float)(1.0 - b1), (float)b2,
                           (float)(1.0 - b2), (float)(-step_size), (float)bc2_sqrt, (float)cfg->eps, cfg->apply_step,
                           grad_norm_out, *per, per_split ? (const void*)c->per_scratch : (const void*)nullptr, skip_flag);
    else
        hipLaunchKernelGGL(clip_adam_kernel, dim3(nblk), dim3(OPT_THREADS), 0, s, params, grads, exp_avg, exp_avg_sq,
                           (long long)c->P, (const double*)c->sumsq_part, nblk, cfg->max_grad_norm, (float)(1.0 - b1), (float)b2,
                           (float)(1.0 - b2), (float)(-step_size), (float)bc2_sqrt, (float)cfg->eps, cfg->apply_step,
                           grad_norm_out, skip_flag);
    LAUNCH_CHECK("clip_adam");
    return MORL_OK;
}

extern "C" int morl_envelope_update(morl_ctx* c, float* params_online, const float* params_target, float* grads,
                                    float* exp_avg, float* exp_avg_sq, const float* obs, const float* next_obs,
                                    const int32_t* actions, const float* rewards, const float* dones,
                                    const float* weights, int B, int W, const morl_update_cfg* cfg,
                                    const morl_update_out* out, void* stream) {
    int rc = check_bw(c, B, W);
    if (rc) return rc;
    if (!params_online || !params_target || !grads || !obs || !next_obs || !actions || !rewards || !dones || !weights ||
        !cfg)
        return fail(MORL_ERR_ARG, "NULL array");
    if (cfg->apply_step && (!exp_avg || !exp_avg_sq)) return fail(MORL_ERR_ARG, "Adam state is NULL");
    if (cfg->apply_step && cfg->adam_step < 1) return fail(MORL_ERR_ARG, "adam_step must be >= 1");
    if (cfg->per_tree && (!cfg->per_idx || !cfg->per_running_max || cfg->per_levels < 1 || cfg->per_levels > 40 || !out || !out->priority))
        return fail(MORL_ERR_ARG, "per_tree needs per_idx, per_running_max, per_levels and out->priority");
    // (checked HERE, before the first launch: a failure after the forward / backward launches would leave the caller with an
    // exception and a half-taken step)
    if (cfg->per_tree && B > ST_MAX_B)
        return fail(MORL_ERR_ARG, "PER update inside the step: B=%d > %d (update the tree in chunks with morl_sumtree_update)", B, ST_MAX_B);
    hipStream_t s = (hipStream_t)stream;
    const morl_net_desc& n = c->net;
    const int D = n.obs_dim, R = n.reward_dim, A = n.n_actions;
    const int rows = B * W, AR = A * R;
    static const morl_update_out no_out = {};
    if (!out) out = &no_out;
    c->lz_last = false;
    c->lz_last_bfn = false;
    c->last_dual = false;
    c->last_roll = false;
    c->last_pw = false;
    c->last_fwd_pw = false;
    c->last_step_W = W;
    timing_begin_step(c);

    // stage A: no-grad next-state slabs Qo, Qt [B][W][A][R] (B*W distinct rows instead of the reference's W^2*B)
    bool main_done = false;
    const bool use_bf = bf_wanted(c, rows);
    // lazily from 4 096 TD rows on when the step runs on the bf16 matrix cores, from 8 192 on the f32 chains (measured there in round 3:
    // 8 192 rows 0.255 ms lazily against 0.262 eagerly, 2 048 rows 0.181 against 0.149); MORL_LAZY_MIN_ROWS: one threshold for both
    static const long long lazy_env = [] { const char* e = getenv("MORL_LAZY_MIN_ROWS"); return e ? atoll(e) : -1ll; }();
    const long long lazy_min_rows = lazy_env >= 0 ? lazy_env : (use_bf ? 4096ll : 8192ll);
    if (use_bf) {
        // The two ONLINE passes (next-state slab, training pass) on the bf16 matrix cores as six split-bf16 products each
        // (mlp_chain_bf.h: fp32-class accuracy at 6/16 of the f32-input MFMA's time), one launch.  The TARGET network stays on the
        // exact fp32 tiles: its selected rows (lazy evaluation, the default) or, when the caller asks for the whole slab, its pass.
        if ((rc = refresh_bf_step(c, params_online, params_target, s, rows))) return rc;
        c->lz_now = c->lazy_targets && cfg->envelope && W >= 2 && !out->q_target_next && (c->lazy_targets == 2 || rows >= lazy_min_rows);
        c->lz_last = c->lz_now;
        // an eagerly evaluated FEW-ROW step (a rank's share of a sharded job, small batches): the target network's pass rides in the
        // forward launch as a third few-row chain instead of a launch of its own on the f32 tiles
        const bool few_eager = !c->lz_now && c->bfn_eager3 && bfn_few(c, 2ll * rows) && cfg->slab_parts <= 1;
        if (((c->lz_now && c->bfn_targets) || few_eager) && !c->bft_ready) {
            // (a caller that did not come through morl_envelope_prepare: the target network's forward stream for the few-row chain)
            const BfSplitArgs a = bf_split_args(c, nullptr, params_target);
            hipLaunchKernelGGL(bf_split_kernel, dim3((a.unit_start[a.n] * 64 + 255) / 256), dim3(256), 0, s, a, c->bf_stream);
            LAUNCH_CHECK("bf_split(target)");
            c->bft_ready = true;
        }
        const BfChain two[2] = {bf_forward_chain(c, params_online, next_obs, weights, B, W, rows, false, c->qo, AR),
                                bf_forward_chain(c, params_online, obs, weights, B, W, rows, true, c->qm, c->ldq)};
        if (c->lz_now) { c->lz_params_target = params_target; c->lz_next_obs = next_obs; }
        {
            // (Measured and dropped in round 4, twice: the online next-state pass alone, then arg-max + target rows on a side stream
            // BESIDE the training forward: profiles/r04_tail_stream_ab.txt)
            // When a row tile of the launch is one, two or four whole transitions (W rows each), the workgroups of the next-state pass
            // take their transitions' arg-max themselves, from the head's accumulators (envelope_argmax_tile): no arg-max launch
            BfChain fwd[2] = {two[0], two[1]};
            EnvelopeTdArgs amax_args{};
            static const bool fuse_env = [] { const char* e = getenv("MORL_ARGMAX_IN_CHAIN"); return e ? atoi(e) != 0 : true; }();   // (A/B)
            const int tile_rows = bf_tile_rows(c, two, 2);
            const bool fuse = fuse_env && c->lz_now && tile_rows != BF_TILE_FEW && (W == tile_rows || 2 * W == tile_rows || (4 * W == tile_rows && tile_rows == 64)) &&
                              cfg->slab_parts <= 1 && R <= MORL_MAX_OBJ;
            if (fuse) {
                int td_waves = 0;
                const long long rows_total_ = cfg->rows_total > 0 ? (long long)cfg->rows_total : (long long)rows;
                const EnvelopeTdArgs p1 = td_args(c, cfg, out, actions, rewards, dones, weights, W, c->qo, c->qt, W, 0, rows_total_, B, &td_waves);
                amax_args = lazy_argmax_args(c, p1);
                fwd[0].amax = 1;
            }
            if (few_eager) {
                const BfChain three[3] = {two[0], two[1], bf_forward_chain(c, params_target, next_obs, weights, B, W, rows, false, c->qt, AR, true)};
                c->bft_ready = false;
                c->lz_last_bfn = true;          // (bit 5 of morl_ctx_last_step_bf16: the target network ran on the few-row split-bf16 chain)
                if ((rc = bf_launch(c, three, 3, MORL_TIMED_FORWARD, s))) return rc;
            } else {
                // The training pass's tiles FIRST in the grid when the launch is one round (a tile per CU at most): they are the longer ones
                // (saves, sign bits) and the second half of a grid starts a microsecond behind the first -- 46.0 -> 45.2 us at 256 x 32.
                // NOT when two tiles share a CU (the flagship): the older workgroup wins the matrix pipe, and with the training tile as the
                // older one the no-grad tile is starved and finishes alone -- 77.7 against 71.5 us (profiles/r06_forward_chains_apart.txt;
                // MORL_BF_T_FIRST=0 / 2: never / always, the A/B legs)
                {
                    long long t64 = 0;
                    for (int qn = 0; qn < 2; ++qn) t64 += (fwd[qn].rows + BF_TM - 1) / BF_TM;
                    if (!c->bf_dual && (c->bf_t_first == 2 || (c->bf_t_first == 1 && tile_rows == BF_TM && t64 <= (long long)c->num_cus)))
                        std::swap(fwd[0], fwd[1]);
                }
                if ((rc = bf_launch(c, fwd, 2, MORL_TIMED_FORWARD2, s, fuse ? &amax_args : nullptr))) { c->lz_now = false; return rc; }
                c->lz_argmax_done = fuse;
                if (!c->lz_now && (rc = chain_forward(c, params_target, c->wt_target, next_obs, weights, B, W, 0, rows, false, c->qt, AR, s))) return rc;
            }
        }
        main_done = true;
        c->bits_valid = true;
        c->bits_bf = true;
    } else if (c->use_fused) {
        // layer-fused passes: activations stay in LDS; rows assembled from (obs, weights) inside the kernel
        c->bits_bf = false;
        if ((rc = refresh_transposed(c, params_online, c->wt_online, s, params_target, c->wt_target, true))) return rc;
        if (c->fused_tm == 0) {
            // one launch for the three forward passes: 3 x rows/64 workgroups -> 2 resident per CU
            ChainArgs main_chain = make_forward_chain(c, params_online, c->wt_online, obs, weights, B, W, 0, rows, true, c->qm,
                                                      c->ldq, true);
            main_chain.x0_out = c->x0m;      // layer-0 input of the dW GEMM, written by the pass that assembles it anyway
            main_chain.ldx0 = c->ld0;
            // Lazy target evaluation: the target network is only ever read at a TD row's arg-max (j*, a*), and the rows of a
            // transition agree on a handful of j* -- so this launch carries the online next-state pass and the training pass only,
            // the arg-max runs on the online slab, and the target network is evaluated afterwards on the distinct (b, j*) pairs
            // (update_core; 1 546 of 16 384 rows at the flagship shape).  Not when the caller asks for the whole target slab.
            // Small steps are latency-bound: one more chain launch and a second TD launch cost them more than the target pass
            // they drop (2 048 rows: 0.181 ms lazily, 0.149 eagerly; 8 192 rows: 0.255 against 0.262; MORL_LAZY_MIN_ROWS overrides)
            c->lz_now = c->lazy_targets && cfg->envelope && W >= 2 && !out->q_target_next &&
                        (c->lazy_targets == 2 || rows >= lazy_min_rows);
            c->lz_last = c->lz_now;
            if (c->lz_now) {
                const ChainArgs two[2] = {
                    make_forward_chain(c, params_online, c->wt_online, next_obs, weights, B, W, 0, rows, false, c->qo, AR), main_chain};
                c->timing_kind_override = MORL_TIMED_FORWARD2;
                rc = chain_forward_multi(c, two, 2, s);
                c->timing_kind_override = -1;
                if (rc) { c->lz_now = false; return rc; }
                c->lz_params_target = params_target;
                c->lz_next_obs = next_obs;
            } else if ((rc = chain_forward_x3(
                     c, make_forward_chain(c, params_online, c->wt_online, next_obs, weights, B, W, 0, rows, false, c->qo, AR),
                     make_forward_chain(c, params_target, c->wt_target, next_obs, weights, B, W, 0, rows, false, c->qt, AR),
                     main_chain, s)))
                return rc;
            main_done = true;
            c->bits_valid = true;   // (64- and 32-row tiles write the same word layout)
        } else {
            if ((rc = chain_forward(c, params_online, c->wt_online, next_obs, weights, B, W, 0, rows, false, c->qo, AR, s))) return rc;
            if ((rc = chain_forward(c, params_target, c->wt_target, next_obs, weights, B, W, 0, rows, false, c->qt, AR, s))) return rc;
        }
    } else {
        c->bits_bf = false;          // (per-layer engine, the A/B oracle of the fused ones: nothing of it runs as split-bf16 products)
        if ((rc = build_input(next_obs, weights, c->x0n, B, W, D, R, c->ld0, 0, s))) return rc;
        if ((rc = net_forward(c, params_online, c->x0n, rows, false, c->qo, AR, s))) return rc;
        if ((rc = net_forward(c, params_target, c->x0n, rows, false, c->qt, AR, s))) return rc;
    }
    // The PER tree update (17 us of serial levels) rides as an extra workgroup of a longer launch: the weight gradients when they
    // take longer than it does, the clip + Adam launch for small steps (<= 4 096 rows: the weight-gradient launch would wait for
    // it -- 37 instead of 20 us at 256 x 8)
    constexpr int per_adam_rows = 4096;
    const bool per_with_adam = cfg->per_tree && out->priority && B <= ST_MAX_B && rows <= per_adam_rows && c->dw_mode == 3;
    morl_update_cfg core_cfg = *cfg;
    if (per_with_adam) core_cfg.per_tree = nullptr;
    if (cfg->rows_total != 0 && cfg->rows_total < rows) return fail(MORL_ERR_ARG, "rows_total %lld < B * W", (long long)cfg->rows_total);
    const long long rows_total = cfg->rows_total > 0 ? (long long)cfg->rows_total : (long long)rows;
    if ((rc = update_core(c, params_online, grads, obs, actions, rewards, dones, weights, W, c->qo, c->qt, W, 0,
                          rows_total, B, &core_cfg, out, main_done, nullptr, nullptr, s)))
        return rc;
    SumTreeUpdate per{};
    if (per_with_adam) {
        per.tree = cfg->per_tree; per.idx = cfg->per_idx; per.raw = out->priority; per.running_max = cfg->per_running_max;
        per.n_levels = cfg->per_levels; per.B = B; per.alpha = cfg->per_alpha;
    }
    // (gradients only, unclipped, nobody asks for the norm: nothing left to do -- the batch-sharded step's local part)
    const bool nothing_left = !cfg->apply_step && cfg->max_grad_norm < 0.f && !out->grad_norm && !per_with_adam;
    if (!nothing_left &&
        (rc = clip_adam_step(c, params_online, grads, exp_avg, exp_avg_sq, cfg, out->grad_norm, true, s, per_with_adam ? &per : nullptr)))
        return rc;
    // optional debug / parity outputs
    if (out->q_online_next) HIP_TRY(hipMemcpyAsync(out->q_online_next, c->qo, (size_t)rows * AR * 4, hipMemcpyDeviceToDevice, s));
    if (out->q_target_next) HIP_TRY(hipMemcpyAsync(out->q_target_next, c->qt, (size_t)rows * AR * 4, hipMemcpyDeviceToDevice, s));
    return MORL_OK;
}

// Envelope.update's loop (envelope.py:269-334) in one entry: see include/morl_hip.h.  Everything that can be refused is refused
// before the first launch of the first iteration.
extern "C" int morl_envelope_update_n(morl_ctx* c, const morl_step_io* io, int n, const double* u01, const int64_t* idx_in,
                                      const float* w_src, int adam_step0, float homotopy_lambda, float* loss_out, float* grad_norm_out,
                                      float* priority_out, void* stream) {
    if (!c || !io) return fail(MORL_ERR_ARG, "NULL argument");
    if (n < 1) return fail(MORL_ERR_ARG, "n = %d updates", n);
    int rc = check_bw(c, io->B, io->W);
    if (rc) return rc;
    if (!io->params_online || !io->params_target || !io->grads || !io->exp_avg || !io->exp_avg_sq || !io->records || !io->obs ||
        !io->next_obs || !io->rewards || !io->dones || !io->actions || !io->idx || !io->weights)
        return fail(MORL_ERR_ARG, "NULL field of morl_step_io");
    if (!w_src || !loss_out) return fail(MORL_ERR_ARG, "w_src / loss_out is NULL");
    if (io->tree ? (!u01 || !io->running_max || !priority_out) : !idx_in)
        return fail(MORL_ERR_ARG, io->tree ? "prioritised replay needs u01, running_max and priority_out" : "uniform replay needs idx_in");
    if (io->tree && (io->n_levels < 1 || io->n_levels > 40)) return fail(MORL_ERR_ARG, "n_levels = %d", io->n_levels);
    if (io->tree && io->B > ST_MAX_B) return fail(MORL_ERR_ARG, "PER update inside the step: B=%d > %d", io->B, ST_MAX_B);
    if (io->D != c->net.obs_dim || io->R != c->net.reward_dim)
        return fail(MORL_ERR_ARG, "records of (obs %d, reward %d), the network takes (%d, %d)", io->D, io->R, c->net.obs_dim, c->net.reward_dim);
    if (adam_step0 < 1) return fail(MORL_ERR_ARG, "adam_step0 must be >= 1");
    const int B = io->B, W = io->W, WR = W * io->R;
    for (int k = 0; k < n; ++k) {
        rc = morl_envelope_prepare(c, io->params_online, io->params_target, io->tree, io->n_levels, io->tree ? u01 + (size_t)k * B : nullptr,
                                   io->tree ? nullptr : idx_in + (size_t)k * B, io->records, io->record_floats, io->capacity, B, io->D,
                                   io->R, 1, io->obs, io->next_obs, io->rewards, io->dones, nullptr, io->actions, io->idx,
                                   w_src + (size_t)k * WR, io->weights, WR, stream);
        if (!rc) {
            morl_update_cfg cfg = io->cfg;
            cfg.adam_step = adam_step0 + k;
            cfg.homotopy_lambda = homotopy_lambda;
            cfg.apply_step = 1;
            cfg.main_forward_done = 0; cfg.slab_parts = 0; cfg.rows_total = 0;
            cfg.per_tree = io->tree; cfg.per_idx = io->tree ? io->idx : nullptr; cfg.per_running_max = io->tree ? io->running_max : nullptr;
            cfg.per_levels = io->n_levels;
            morl_update_out out = {};
            out.loss = loss_out + k;
            out.grad_norm = grad_norm_out ? grad_norm_out + k : nullptr;
            out.priority = priority_out;
            rc = morl_envelope_update(c, io->params_online, io->params_target, io->grads, io->exp_avg, io->exp_avg_sq, io->obs, io->next_obs,
                                      io->actions, io->rewards, io->dones, io->weights, B, W, &cfg, &out, stream);
        }
        if (rc) {
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", morl_last_error());
            return fail(rc, "update %d of %d: %s", k, n, msg);
        }
    }
    return MORL_OK;
}

// Weight-sharded variant, stage B only: this rank owns the TD rows of weights [i_offset, i_offset + W_local) and is
// handed the all-gathered slabs qo_all / qt_all [B][W_total][A][R].  Writes this rank's UNCLIPPED gradient
// contribution (already normalised by the global row count) to `grads` and its share of the loss to out->loss; the
// caller all-reduces both and then calls morl_clip_adam.
extern "C" int morl_envelope_update_shard(morl_ctx* c, const float* params_online, float* grads, const float* obs,
                                          const int32_t* actions, const float* rewards, const float* dones,
                                          const float* weights_all, int B, int W_total, int i_offset, int W_local,
                                          const float* qo_all, const float* qt_all, const morl_update_cfg* cfg,
                                          const morl_update_out* out, void* stream) {
    int rc = check_bw(c, B, W_local);
    if (rc) return rc;
    if (!params_online || !grads || !obs || !actions || !rewards || !dones || !weights_all || !qo_all || !qt_all || !cfg)
        return fail(MORL_ERR_ARG, "NULL array");
    if (W_total < 1 || i_offset < 0 || W_local < 1 || i_offset + W_local > W_total)
        return fail(MORL_ERR_ARG, "bad shard [%d, %d) of %d weights", i_offset, i_offset + W_local, W_total);
    if ((long long)W_total * c->net.n_actions * c->net.reward_dim > ENV_MAX_SLAB || W_total * c->net.reward_dim > ENV_MAX_WR)
        return fail(MORL_ERR_ARG, "W_total*A*R exceeds the LDS slab of the envelope kernel (%d floats)", ENV_MAX_SLAB);
    hipStream_t s = (hipStream_t)stream;
    static const morl_update_out no_out = {};
    if (!out) out = &no_out;
    if (cfg->slab_parts > 1 && W_total % cfg->slab_parts)
        return fail(MORL_ERR_ARG, "slab_parts %d does not divide W_total %d", cfg->slab_parts, W_total);
    const bool fwd_done = cfg->main_forward_done != 0;
    if (fwd_done && (c->main_rows != B * W_local))
        return fail(MORL_ERR_STATE, "main_forward_done: morl_envelope_main_forward was not run for %d rows", B * W_local);
    if (!fwd_done && c->use_fused && (rc = refresh_transposed(c, params_online, c->wt_online, s))) return rc;
    c->main_rows = -1;
    // lazily evaluated: arg-max on the gathered ONLINE slab, the target network on the pairs this rank's TD rows selected
    c->lz_now = false;
    c->lz_last = false;
    if (cfg->shard_params_target && cfg->shard_next_obs) {
        if (!cfg->envelope) return fail(MORL_ERR_ARG, "lazy target evaluation is for envelope targets (cfg->envelope = 1)");
        if (!c->use_fused) return fail(MORL_ERR_STATE, "lazy target evaluation needs the layer-fused engine");
        c->lz_now = c->lz_last = true;
        c->lz_params_target = cfg->shard_params_target;
        c->lz_next_obs = cfg->shard_next_obs;
        c->lz_weights_all = weights_all;
    }
    rc = update_core(c, params_online, grads, obs, actions, rewards, dones,
                     weights_all + (size_t)i_offset * c->net.reward_dim, W_local, qo_all, qt_all, W_total, i_offset,
                     (long long)B * W_total, B, cfg, out, fwd_done, nullptr, nullptr, s);
    c->lz_now = false;
    c->lz_weights_all = nullptr;
    return rc;
}

// does the one-call weight-sharded step of B x W_local rows per rank evaluate its targets lazily?  (the unsharded step's rule)
static bool shard_lazy(const morl_ctx* c, int B, int W_local, int envelope) {
    static const long long lazy_min_rows = [] { const char* e = getenv("MORL_LAZY_MIN_ROWS"); return e ? atoll(e) : 8192ll; }();
    return c->use_fused && c->lazy_targets && envelope && (c->lazy_targets == 2 || (long long)B * W_local >= lazy_min_rows);
}

extern "C" int morl_ctx_shard_lazy(morl_ctx* c, int B, int W_local, int envelope) {
    if (!c) return fail(MORL_ERR_ARG, "ctx is NULL");
    return shard_lazy(c, B, W_local, envelope) ? 1 : 0;
}

extern "C" int morl_envelope_slab_online(morl_ctx* c, const float* params_online, const float* params_target, const float* next_obs,
                                         const float* weights_local, int B, int W_local, float* slab_out, void* stream) {
    int rc = check_bw(c, B, W_local);
    if (rc) return rc;
    if (!params_online || !params_target || !next_obs || !weights_local || !slab_out) return fail(MORL_ERR_ARG, "NULL array");
    if (!c->use_fused) return fail(MORL_ERR_STATE, "morl_envelope_slab_online needs the layer-fused engine");
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * W_local, AR = c->net.n_actions * c->net.reward_dim;
    timing_begin_step(c);
    c->last_step_W = W_local;
    if (bf_wanted(c, rows, true)) {
        // the rank's share is large enough for the bf16 matrix cores: split weight streams of the online network (+ the target's
        // K-major copy for the target rows), the next-state pass as one split-bf16 chain; the training pass and the backward follow
        if ((rc = refresh_bf_step(c, params_online, params_target, s))) return rc;
        c->wt_online_src = nullptr;
        c->bf_stream_src = params_online;
        const BfChain one = bf_forward_chain(c, params_online, next_obs, weights_local, B, W_local, rows, false, slab_out, AR);
        return bf_launch(c, &one, 1, MORL_TIMED_FORWARD, s);
    }
    c->bf_stream_src = nullptr;
    if ((rc = refresh_transposed(c, params_online, c->wt_online, s, params_target, c->wt_target, true))) return rc;
    c->wt_online_src = params_online;
    const ChainArgs one = make_forward_chain(c, params_online, c->wt_online, next_obs, weights_local, B, W_local, 0, rows, false, slab_out, AR);
    return chain_forward_multi(c, &one, 1, s);
}

extern "C" int morl_envelope_slabs(morl_ctx* c, const float* params_online, const float* params_target, const float* next_obs,
                                   const float* weights_local, int B, int W_local, float* slabs_out, void* stream) {
    int rc = check_bw(c, B, W_local);
    if (rc) return rc;
    if (!params_online || !params_target || !next_obs || !weights_local || !slabs_out) return fail(MORL_ERR_ARG, "NULL array");
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * W_local, AR = c->net.n_actions * c->net.reward_dim;
    timing_begin_step(c);
    c->last_step_W = W_local;            // (what a staged caller's morl_envelope_prepare expects of its next step)
    float* qo = slabs_out;
    float* qt = slabs_out + (size_t)rows * AR;
    if (c->use_fused) {
        // one transpose launch for both networks, one launch for both passes (2 x rows/64 workgroups share the chip)
        if ((rc = refresh_transposed(c, params_online, c->wt_online, s, params_target, c->wt_target, true))) return rc;
        c->wt_online_src = params_online;
        const ChainArgs two[2] = {
            make_forward_chain(c, params_online, c->wt_online, next_obs, weights_local, B, W_local, 0, rows, false, qo, AR),
            make_forward_chain(c, params_target, c->wt_target, next_obs, weights_local, B, W_local, 0, rows, false, qt, AR)};
        return chain_forward_multi(c, two, 2, s);
    }
    if ((rc = build_input(next_obs, weights_local, c->x0n, B, W_local, c->net.obs_dim, c->net.reward_dim, c->ld0, 0, s))) return rc;
    if ((rc = net_forward(c, params_online, c->x0n, rows, false, qo, AR, s))) return rc;
    return net_forward(c, params_target, c->x0n, rows, false, qt, AR, s);
}

extern "C" int morl_envelope_main_forward(morl_ctx* c, const float* params_online, const float* obs,
                                          const float* weights_local, int B, int W_local, void* stream) {
    int rc = check_bw(c, B, W_local);
    if (rc) return rc;
    if (!params_online || !obs || !weights_local) return fail(MORL_ERR_ARG, "NULL array");
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * W_local;
    if (c->use_fused && bf_wanted(c, rows, true) && c->bf_stream_src == params_online) {
        // the split weight streams this step's morl_envelope_slab_online made are current: the training pass on the bf16 matrix cores
        // (the backward chain and the weight gradients follow it there: update_core looks at bits_bf)
        c->bf_stream_src = nullptr;
        const BfChain one = bf_forward_chain(c, params_online, obs, weights_local, B, W_local, rows, true, c->qm, c->ldq);
        if ((rc = bf_launch(c, &one, 1, MORL_TIMED_FORWARD, s))) return rc;
        c->bits_valid = true;
        c->bits_bf = true;
    } else if (c->use_fused) {
        c->bf_stream_src = nullptr;
        // exactly the third pass of the unsharded step's fused launch: activations and ReLU sign bits saved in the context,
        // the layer-0 input of the dW GEMM written from the kernel's own input assembly
        // the K-major copy made by this step's morl_envelope_slabs is still current unless an optimiser step intervened
        if (c->wt_online_src != params_online && (rc = refresh_transposed(c, params_online, c->wt_online, s))) return rc;
        c->wt_online_src = nullptr;      // one-shot: only the call that directly follows the slabs call of a step reuses it
        ChainArgs one = make_forward_chain(c, params_online, c->wt_online, obs, weights_local, B, W_local, 0, rows, true, c->qm,
                                           c->ldq, true);
        one.x0_out = c->x0m;
        one.ldx0 = c->ld0;
        if ((rc = chain_forward_multi(c, &one, 1, s))) return rc;
        c->bits_valid = true;
        c->bits_bf = false;
    } else {
        if ((rc = build_input(obs, weights_local, c->x0m, B, W_local, c->net.obs_dim, c->net.reward_dim, c->ld0, 1, s))) return rc;
        c->bits_valid = false;
        if ((rc = net_forward(c, params_online, c->x0m, rows, true, c->qm, c->ldq, s))) return rc;
    }
    c->main_rows = rows;
    return MORL_OK;
}

// The sharded step of one rank in one call (include/morl_hip.h): the entry points above and the collectives of morl_comm.hip
// in the order distributed.py used to issue them from the interpreter.
extern "C" int morl_envelope_step_sharded(morl_ctx* c, morl_comm* comm, float* params_online, const float* params_target,
                                          float* grads_x, int64_t n_params, float* exp_avg, float* exp_avg_sq,
                                          const float* obs, const float* next_obs, const int32_t* actions,
                                          const float* rewards, const float* dones, const float* weights_all, int B,
                                          int W_total, int i_offset, int W_local, float* slab_local, float* slab_all,
                                          const morl_update_cfg* cfg, void* stream) {
    if (!c || !comm || !cfg || !grads_x || !slab_local || !slab_all || !weights_all)
        return fail(MORL_ERR_ARG, "NULL argument");
    if (n_params != c->P) return fail(MORL_ERR_ARG, "n_params %lld, the network has %lld", (long long)n_params, (long long)c->P);
    if (W_total < 1 || W_local < 1 || i_offset < 0 || i_offset + W_local > W_total || W_total % W_local || i_offset % W_local)
        return fail(MORL_ERR_ARG, "bad shard [%d, %d) of %d weights", i_offset, i_offset + W_local, W_total);
    if (cfg->per_tree && (!cfg->per_idx || !cfg->per_running_max || cfg->per_levels < 1 || cfg->per_levels > 40))
        return fail(MORL_ERR_ARG, "per_tree needs per_idx, per_running_max and per_levels");
    // everything that can be refused is refused before the first launch / collective: a status after the all-reduce would leave
    // the ranks with the optimiser state advanced and the priorities not (or the other way round)
    if (cfg->apply_step && (!exp_avg || !exp_avg_sq)) return fail(MORL_ERR_ARG, "Adam state is NULL");
    if (cfg->apply_step && cfg->adam_step < 1) return fail(MORL_ERR_ARG, "adam_step must be >= 1");
    if (cfg->per_tree && B > ST_MAX_B) return fail(MORL_ERR_ARG, "PER update inside the step: B=%d > %d", B, ST_MAX_B);
    int rank = 0, world = 1, rc;
    if ((rc = morl_comm_size(comm, &rank, &world))) return rc;
    const int parts = W_total / W_local;
    if (world != parts && world != 1)
        return fail(MORL_ERR_ARG, "%d ranks in the communicator, %d shards of the weight axis", world, parts);
    if (world == parts && rank != i_offset / W_local)
        return fail(MORL_ERR_ARG, "rank %d does not own the weights from %d on", rank, i_offset);
    const int R = c->net.reward_dim, AR = c->net.n_actions * R;
    const int64_t half = (int64_t)B * W_local * AR;            // one network's slab of one rank
    const float* w_loc = weights_all + (size_t)i_offset * R;
    // Lazily evaluated (the unsharded step's rule applied to the rank's rows): only Q_online(s', w_j) is exchanged -- half the
    // all-gather --, the arg-max of the rank's TD rows runs over all gathered candidates, and the rank evaluates the target network
    // on the pairs those rows selected (envelope.py:422-439).  slab_local / slab_all are used as [B][W_local][A][R] / [G][...].
    const bool lazy = shard_lazy(c, B, W_local, cfg->envelope);
    const int64_t part = lazy ? half : 2 * half;
    float* recv = (world == parts) ? slab_all : slab_all + (size_t)(i_offset / W_local) * part;
    if (lazy) rc = morl_envelope_slab_online(c, params_online, params_target, next_obs, w_loc, B, W_local, slab_local, stream);
    else rc = morl_envelope_slabs(c, params_online, params_target, next_obs, w_loc, B, W_local, slab_local, stream);
    if (rc) return rc;
    if ((rc = morl_allgather_q_begin(comm, slab_local, recv, part, stream))) return rc;
    if ((rc = morl_envelope_main_forward(c, params_online, obs, w_loc, B, W_local, stream))) return rc;   // beside the exchange
    if ((rc = morl_comm_wait(comm, stream))) return rc;
    morl_update_cfg shard = *cfg;
    shard.apply_step = 0;
    shard.main_forward_done = 1;
    shard.slab_parts = parts;
    shard.per_tree = nullptr;                                  // priorities are complete only after the all-reduce
    shard.shard_params_target = lazy ? params_target : nullptr;
    shard.shard_next_obs = lazy ? next_obs : nullptr;
    morl_update_out out = {};
    out.loss = grads_x + n_params;
    out.priority = grads_x + n_params + 1;
    if ((rc = morl_envelope_update_shard(c, params_online, grads_x, obs, actions, rewards, dones, weights_all, B, W_total,
                                         i_offset, W_local, slab_all, slab_all + half, &shard, &out, stream)))
        return rc;
    if ((rc = morl_allreduce_grads(comm, grads_x, n_params + 1 + B, stream))) return rc;
    c->skip_flag = morl_host::comm_error_word(comm);
    if (cfg->per_tree) {
        SumTreeUpdate u{};
        u.tree = cfg->per_tree; u.idx = cfg->per_idx; u.raw = grads_x + n_params + 1; u.running_max = cfg->per_running_max;
        u.n_levels = cfg->per_levels; u.B = B; u.alpha = cfg->per_alpha;
        return clip_adam_step(c, params_online, grads_x, exp_avg, exp_avg_sq, cfg, nullptr, false, (hipStream_t)stream, &u);
    }
    return morl_clip_adam(c, params_online, grads_x, exp_avg, exp_avg_sq, cfg, nullptr, stream);
}

// Batch-axis sharding of the step in one call (include/morl_hip.h): the unsharded pipeline on this rank's transitions, one
// all-reduce, the optimiser step.
extern "C" int morl_envelope_step_batch_sharded(morl_ctx* c, morl_comm* comm, float* params_online, const float* params_target,
                                                float* grads_x, int64_t n_params, float* exp_avg, float* exp_avg_sq,
                                                const float* obs, const float* next_obs, const int32_t* actions,
                                                const float* rewards, const float* dones, const float* weights, int B,
                                                int B_total, int b_offset, int W, const morl_update_cfg* cfg, void* stream) {
    if (!c || !comm || !cfg || !grads_x) return fail(MORL_ERR_ARG, "NULL argument");
    if (n_params != c->P) return fail(MORL_ERR_ARG, "n_params %lld, the network has %lld", (long long)n_params, (long long)c->P);
    if (B < 1 || B_total < B || b_offset < 0 || b_offset + B > B_total || B_total % B || b_offset % B)
        return fail(MORL_ERR_ARG, "bad shard [%d, %d) of a batch of %d", b_offset, b_offset + B, B_total);
    if (cfg->per_tree && (!cfg->per_idx || !cfg->per_running_max || cfg->per_levels < 1 || cfg->per_levels > 40))
        return fail(MORL_ERR_ARG, "per_tree needs per_idx, per_running_max and per_levels");
    if (cfg->apply_step && (!exp_avg || !exp_avg_sq)) return fail(MORL_ERR_ARG, "Adam state is NULL");
    if (cfg->apply_step && cfg->adam_step < 1) return fail(MORL_ERR_ARG, "adam_step must be >= 1");
    if (cfg->per_tree && B_total > ST_MAX_B) return fail(MORL_ERR_ARG, "PER update inside the step: B_total=%d > %d", B_total, ST_MAX_B);
    int rank = 0, world = 1, rc;
    if ((rc = morl_comm_size(comm, &rank, &world))) return rc;
    const int parts = B_total / B;
    if (world != parts && world != 1)
        return fail(MORL_ERR_ARG, "%d ranks in the communicator, %d shards of the batch", world, parts);
    if (world == parts && rank != b_offset / B) return fail(MORL_ERR_ARG, "rank %d does not own the transitions from %d on", rank, b_offset);
    hipStream_t s = (hipStream_t)stream;
    float* prio = grads_x + n_params + 1;
    // the other ranks' transitions: zeros -- written by the step's TD launch (one memset launch less)
    if (parts > 1) { c->td_zero_ptr = prio; c->td_zero_n = B_total; c->td_keep_lo = b_offset; c->td_keep_hi = b_offset + B; }
    morl_update_cfg local = *cfg;
    local.apply_step = 0;
    local.per_tree = nullptr;                                  // priorities are complete only after the all-reduce
    local.rows_total = (int64_t)B_total * W;
    const float max_norm = local.max_grad_norm;
    local.max_grad_norm = -1.f;                                // the clip belongs to the SUMMED gradient
    morl_update_out out = {};
    out.loss = grads_x + n_params;
    out.priority = prio + b_offset;
    rc = morl_envelope_update(c, params_online, params_target, grads_x, exp_avg, exp_avg_sq, obs, next_obs, actions, rewards, dones,
                              weights, B, W, &local, &out, stream);
    c->td_zero_ptr = nullptr;                                  // (one-shot: never left behind by a call that failed early)
    if (rc) return rc;
    if ((rc = morl_allreduce_grads(comm, grads_x, n_params + 1 + B_total, stream))) return rc;
    (void)max_norm;
    c->skip_flag = morl_host::comm_error_word(comm);
    if (cfg->per_tree) {
        SumTreeUpdate u{};
        u.tree = cfg->per_tree; u.idx = cfg->per_idx; u.raw = prio; u.running_max = cfg->per_running_max;
        u.n_levels = cfg->per_levels; u.B = B_total; u.alpha = cfg->per_alpha;
        return clip_adam_step(c, params_online, grads_x, exp_avg, exp_avg_sq, cfg, nullptr, false, s, &u);
    }
    return morl_clip_adam(c, params_online, grads_x, exp_avg, exp_avg_sq, cfg, nullptr, stream);
}

// One rank's sharded iteration with its sampling in one entry (include/morl_hip.h)
extern "C" int morl_envelope_rank_step(morl_ctx* c, morl_comm* comm, const morl_step_io* io, int axis, int offset, int share,
                                       const double* u01, const int64_t* idx_in, const float* w_src, int adam_step, float homotopy_lambda,
                                       float* slab_local, float* slab_all, void* stream) {
    if (!c || !comm || !io) return fail(MORL_ERR_ARG, "NULL argument");
    if (axis != 0 && axis != 1) return fail(MORL_ERR_ARG, "axis %d (0: batch, 1: weights)", axis);
    if (!io->params_online || !io->params_target || !io->grads || !io->exp_avg || !io->exp_avg_sq || !io->records || !io->obs ||
        !io->next_obs || !io->rewards || !io->dones || !io->actions || !io->idx || !io->weights)
        return fail(MORL_ERR_ARG, "NULL field of morl_step_io");
    if (!w_src) return fail(MORL_ERR_ARG, "w_src is NULL");
    if (io->tree ? (!u01 || !io->running_max) : !idx_in)
        return fail(MORL_ERR_ARG, io->tree ? "prioritised replay needs u01 and running_max" : "uniform replay needs idx_in");
    if (io->tree && (io->n_levels < 1 || io->n_levels > 40)) return fail(MORL_ERR_ARG, "n_levels = %d", io->n_levels);
    if (io->D != c->net.obs_dim || io->R != c->net.reward_dim)
        return fail(MORL_ERR_ARG, "records of (obs %d, reward %d), the network takes (%d, %d)", io->D, io->R, c->net.obs_dim, c->net.reward_dim);
    const int B = io->B, W = io->W, D = io->D, R = io->R;
    if (share < 1 || offset < 0 || offset + share > (axis == 0 ? B : W))
        return fail(MORL_ERR_ARG, "bad shard [%d, %d) of %d", offset, offset + share, axis == 0 ? B : W);
    if (axis == 1 && (!slab_local || !slab_all)) return fail(MORL_ERR_ARG, "the weight axis needs the slab buffers");
    int rc = check_bw(c, axis == 0 ? share : B, axis == 0 ? W : share);
    if (rc) return rc;
    // (before anything is enqueued: the sharded steps update the PER tree inside their clip + Adam launch, one launch's worth of entries)
    if (io->tree && B > ST_MAX_B) return fail(MORL_ERR_ARG, "PER update inside the rank step: B=%d > %d", B, ST_MAX_B);
    // the prologue launch makes the weight copies of the engine THIS rank's rows will run on
    c->prepare_rows = axis == 0 ? (long long)share * W : (long long)B * share;
    c->prepare_weight_shard = axis == 1;
    if ((rc = morl_envelope_prepare(c, io->params_online, io->params_target, io->tree, io->n_levels, io->tree ? u01 : nullptr,
                                    io->tree ? nullptr : idx_in, io->records, io->record_floats, io->capacity, B, D, R, 1, io->obs,
                                    io->next_obs, io->rewards, io->dones, nullptr, io->actions, io->idx, w_src, io->weights, W * R, stream)))
        return rc;
    morl_update_cfg cfg = io->cfg;
    cfg.adam_step = adam_step;
    cfg.homotopy_lambda = homotopy_lambda;
    cfg.apply_step = 1;
    cfg.main_forward_done = 0; cfg.slab_parts = 0; cfg.rows_total = 0;
    cfg.shard_params_target = nullptr; cfg.shard_next_obs = nullptr;
    cfg.per_tree = io->tree; cfg.per_idx = io->tree ? io->idx : nullptr; cfg.per_running_max = io->tree ? io->running_max : nullptr;
    cfg.per_levels = io->n_levels;
    if (axis == 0)
        return morl_envelope_step_batch_sharded(c, comm, io->params_online, io->params_target, io->grads, c->P, io->exp_avg, io->exp_avg_sq,
                                                io->obs + (size_t)offset * D, io->next_obs + (size_t)offset * D, io->actions + offset,
                                                io->rewards + (size_t)offset * R, io->dones + offset, io->weights, share, B, offset, W, &cfg,
                                                stream);
    return morl_envelope_step_sharded(c, comm, io->params_online, io->params_target, io->grads, c->P, io->exp_avg, io->exp_avg_sq, io->obs,
                                      io->next_obs, io->actions, io->rewards, io->dones, io->weights, B, W, offset, share, slab_local, slab_all,
                                      &cfg, stream);
}

// clip_grad_norm_ + Adam on flat buffers (envelope.py:324-326) -- stage C on its own, for gradients that were
// all-reduced between the stages.
extern "C" int morl_clip_adam(morl_ctx* c, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                              const morl_update_cfg* cfg, float* grad_norm_out, void* stream) {
    if (!c || !params || !grads || !cfg) return fail(MORL_ERR_ARG, "NULL argument");
    if (cfg->apply_step && (!exp_avg || !exp_avg_sq)) return fail(MORL_ERR_ARG, "Adam state is NULL");
    return clip_adam_step(c, params, grads, exp_avg, exp_avg_sq, cfg, grad_norm_out, false, (hipStream_t)stream);
}

extern "C" int morl_polyak(const float* src, float* dst, float tau, int64_t n, void* stream) {
    if (!src || !dst) return fail(MORL_ERR_ARG, "NULL array");
    if (n < 0) return fail(MORL_ERR_ARG, "n < 0");
    if (n == 0) return MORL_OK;
    hipLaunchKernelGGL(polyak_kernel, dim3(stream_grid(n, OPT_THREADS)), dim3(OPT_THREADS), 0, (hipStream_t)stream, src,
                       dst, (long long)n, tau, (float)(1.0 - (double)tau));
    LAUNCH_CHECK("polyak");
    return MORL_OK;
}

extern "C" int morl_pareto_mask(const double* points, int N, int R, int remove_duplicates, uint8_t* mask_out,
                                void* stream) {
    if (N == 0) return MORL_OK;
    if (!points || !mask_out) return fail(MORL_ERR_ARG, "NULL array");
    if (N < 0 || R < 1 || R > MORL_MAX_OBJ) return fail(MORL_ERR_ARG, "bad sizes N=%d R=%d (R <= %d)", N, R, MORL_MAX_OBJ);
    hipStream_t s = (hipStream_t)stream;
    // j ranges: aim at ~16 waves per SIMD over the chip (1024 SIMDs), ranges of whole unroll groups, at least 64 points
    const int bx = (N + PARETO_THREADS - 1) / PARETO_THREADS;
    const int want = std::max(1, (16 * 1024) / (bx * (PARETO_THREADS / 64)));
    int chunk = (N + want - 1) / want;
    chunk = std::max(64, (chunk + 63) / 64 * 64);
    const int by = (N + chunk - 1) / chunk;
    const dim3 grid(bx, by), block(PARETO_THREADS);
    HIP_TRY(hipMemsetAsync(mask_out, 1, (size_t)N, s));            // keep unless some j range clears it
    switch (R) {      // one instantiation per objective count: the comparison chain is fully unrolled
#define MORL_PARETO_CASE(r) case r: hipLaunchKernelGGL(pareto_mask_kernel<r>, grid, block, 0, s, points, N, remove_duplicates, chunk, mask_out); break;
        MORL_PARETO_CASE(1) MORL_PARETO_CASE(2) MORL_PARETO_CASE(3) MORL_PARETO_CASE(4)
        MORL_PARETO_CASE(5) MORL_PARETO_CASE(6) MORL_PARETO_CASE(7) MORL_PARETO_CASE(8)
#undef MORL_PARETO_CASE
        default: return fail(MORL_ERR_ARG, "R=%d", R);
    }
    LAUNCH_CHECK("pareto_mask");
    return MORL_OK;
}

// ---- front metrics (performance_indicators.py:15-25, 71-91) ---------------------------------------------------------------
extern "C" int64_t morl_metrics_workspace_doubles(int N, int R) {
    if (N < 0 || R < 1) return -1;
    return (int64_t)std::max(R - 1, 0) * N + HV_MAX_BLOCKS;
}

extern "C" int morl_hypervolume(const double* points, int N, int R, const double* ref_point, double* workspace,
                                double* hv_out, void* stream) {
    if (!ref_point || !hv_out || (N > 0 && (!points || !workspace))) return fail(MORL_ERR_ARG, "NULL array");
    if (N < 0 || R < 1 || R > MORL_MAX_OBJ) return fail(MORL_ERR_ARG, "bad sizes N=%d R=%d (R <= %d)", N, R, MORL_MAX_OBJ);
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        HIP_TRY(hipMemsetAsync(hv_out, 0, sizeof(double), s));
        return MORL_OK;
    }
    double boxes = 1.0;
    for (int d = 0; d < R - 1; ++d) boxes *= (double)N;
    if (boxes * (double)N > 4e11)
        return fail(MORL_ERR_ARG, "hypervolume of %d points in %d objectives needs %.3g point tests (limit 4e11)", N, R,
                    boxes * (double)N);
    const long long n_boxes = (long long)boxes;
    double* coords = workspace;
    double* part = workspace + (size_t)(R - 1) * N;
    if (R > 1) {
        hipLaunchKernelGGL(hv_sort_kernel, dim3(R - 1), dim3(HV_THREADS), 0, s, points, N, R, ref_point, coords);
        LAUNCH_CHECK("hv_sort");
    }
    const int blocks = (int)std::max(1ll, std::min<long long>(HV_MAX_BLOCKS, (n_boxes + HV_THREADS - 1) / HV_THREADS));
    hipLaunchKernelGGL(hv_boxes_kernel, dim3(blocks), dim3(HV_THREADS), 0, s, points, N, R, ref_point, (const double*)coords,
                       n_boxes, part);
    LAUNCH_CHECK("hv_boxes");
    hipLaunchKernelGGL(metric_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)part, blocks, 1.0, hv_out);
    LAUNCH_CHECK("hv_finish");
    return MORL_OK;
}

extern "C" int morl_expected_utility(const double* front, int N, int R, const double* weights, int M, double* workspace,
                                     double* eum_out, void* stream) {
    if (!front || !weights || !workspace || !eum_out) return fail(MORL_ERR_ARG, "NULL array");
    if (N < 1 || M < 1 || R < 1 || R > MORL_MAX_OBJ || (M + 3) / 4 > HV_MAX_BLOCKS)
        return fail(MORL_ERR_ARG, "bad sizes N=%d M=%d R=%d (M <= %d, R <= %d)", N, M, R, 4 * HV_MAX_BLOCKS, MORL_MAX_OBJ);
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (M + HV_THREADS / 64 - 1) / (HV_THREADS / 64);
    hipLaunchKernelGGL(eum_kernel, dim3(blocks), dim3(HV_THREADS), 0, s, front, N, R, weights, M, workspace);
    LAUNCH_CHECK("eum");
    hipLaunchKernelGGL(metric_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, blocks, 1.0 / (double)M,
                       eum_out);
    LAUNCH_CHECK("eum_finish");
    return MORL_OK;
}

extern "C" int morl_sumtree_sample(const double* tree, int n_levels, const double* u01, int B, int64_t* idx,
                                   void* stream) {
    if (!tree || !u01 || !idx) return fail(MORL_ERR_ARG, "NULL array");
    if (n_levels < 1 || n_levels > 40 || B < 1) return fail(MORL_ERR_ARG, "bad sizes");
    hipLaunchKernelGGL(sumtree_sample_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, tree, n_levels,
                       u01, B, idx);
    LAUNCH_CHECK("sumtree_sample");
    return MORL_OK;
}

extern "C" int morl_sumtree_set(double* tree, int n_levels, const int64_t* ptr, const double* value, int n,
                                double* running_max, void* stream) {
    if (!tree || !ptr || !running_max) return fail(MORL_ERR_ARG, "NULL array");
    if (n_levels < 1 || n_levels > 40 || n < 0) return fail(MORL_ERR_ARG, "bad sizes");
    if (n == 0) return MORL_OK;
    hipLaunchKernelGGL(sumtree_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tree, n_levels, ptr, value, n,
                       (const double*)running_max);
    LAUNCH_CHECK("sumtree_set");
    return MORL_OK;
}

extern "C" int morl_sumtree_update(double* tree, int n_levels, const int64_t* idx, const float* raw, int B,
                                   double alpha, double* running_max, double* pr_out, void* stream) {
    if (!tree || !idx || !raw || !running_max) return fail(MORL_ERR_ARG, "NULL array");
    if (n_levels < 1 || n_levels > 40) return fail(MORL_ERR_ARG, "bad n_levels");
    if (B < 1 || B > ST_MAX_B) return fail(MORL_ERR_ARG, "B=%d outside [1,%d]", B, ST_MAX_B);
    SumTreeUpdate u{};
    u.tree = tree; u.idx = idx; u.raw = raw; u.running_max = running_max; u.pr_out = pr_out;
    u.n_levels = n_levels; u.B = B; u.alpha = (float)alpha;
    hipLaunchKernelGGL(sumtree_update_kernel, dim3(1), dim3(ST_THREADS), 0, (hipStream_t)stream, u);
    LAUNCH_CHECK("sumtree_update");
    return MORL_OK;
}

extern "C" int morl_sumtree_update_clamped(double* tree, int n_levels, const int64_t* idx, const float* raw, int B,
                                           double alpha, double clamp_min, double* running_max, double* pr_out, void* stream) {
    if (!tree || !idx || !raw || !running_max) return fail(MORL_ERR_ARG, "NULL array");
    if (n_levels < 1 || n_levels > 40) return fail(MORL_ERR_ARG, "bad n_levels");
    if (B < 1 || B > ST_MAX_B) return fail(MORL_ERR_ARG, "B=%d outside [1,%d]", B, ST_MAX_B);
    if (!(alpha >= 0.0) || !(clamp_min > 0.0)) return fail(MORL_ERR_ARG, "alpha %g must be >= 0 and clamp_min %g > 0", alpha, clamp_min);
    SumTreeUpdate u{};
    u.tree = tree; u.idx = idx; u.raw = raw; u.running_max = running_max; u.pr_out = pr_out;
    u.n_levels = n_levels; u.B = B; u.alpha = (float)alpha; u.clamp_min = (float)clamp_min;
    hipLaunchKernelGGL(sumtree_update_kernel, dim3(1), dim3(ST_THREADS), 0, (hipStream_t)stream, u);
    LAUNCH_CHECK("sumtree_update_clamped");
    return MORL_OK;
}
