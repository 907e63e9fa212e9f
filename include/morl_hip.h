/*
 * morl_hip.h -- C ABI of libmorl_hip.so: the MI355X (gfx950) native hot path of the
 * multi-objective TD update of LucasAlegre/morl-baselines.
 *
 * The reference has no FFI/plugin boundary (it is 100 % Python on PyTorch); the boundary it does
 * have is the Python class API (MOPolicy.update / Envelope.envelope_target / ReplayBuffer.sample,
 * common/pareto.py functions).  Each entry point below replaces the arithmetic of one of those
 * reference call sites (cited per function as path:line under /root/reference/morl_baselines)
 * and is bound from Python with ctypes (INTEGRATION.md shows the stub a maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, a negative morl_status otherwise; the message of the
 *     last failure on the calling thread is available from morl_last_error().
 *   - every pointer marked "device" is caller-owned device memory (e.g. a torch tensor's
 *     data_ptr()); the library never frees or retains it beyond the call.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, no
 *     call synchronises the host unless stated.
 *   - plain C types only: no torch / HIP types in any signature.
 *   - a morl_ctx owns only scratch workspace (activations, split-K slabs); one ctx per agent,
 *     not thread-safe per ctx.
 */
#ifndef MORL_HIP_H
#define MORL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MORL_MAX_LAYERS 8   /* linear layers per network */
#define MORL_MAX_OBJ 8      /* reward dimension R */
#define MORL_ABI_VERSION 13

typedef enum morl_status {
    MORL_OK = 0,
    MORL_ERR_ARG = -1,      /* bad argument (shape, alignment, null) */
    MORL_ERR_HIP = -2,      /* a HIP runtime call failed */
    MORL_ERR_STATE = -3,    /* ctx too small for the request */
    MORL_ERR_ALLOC = -4
} morl_status;

typedef struct morl_ctx morl_ctx;

/* Weight-conditioned Q network: cat(obs, w) -> [Linear, ReLU] * (n_layers-1) -> Linear -> (A, R).
 * dims[0] = D + R, dims[n_layers] = A * R.  Flat parameter layout (what the torch nn.Parameters
 * are views of): for l = 0..n_layers-1: W_l (dims[l+1] x dims[l], row-major = nn.Linear.weight),
 * then b_l (dims[l+1]).  Replaces QNet / mlp: multi_policy/envelope/envelope.py:33-77,
 * common/networks.py:10-48. */
typedef struct morl_net_desc {
    int32_t n_layers;
    int32_t dims[MORL_MAX_LAYERS + 1];
    int32_t obs_dim;     /* D */
    int32_t reward_dim;  /* R */
    int32_t n_actions;   /* A */
} morl_net_desc;

/* Hyper-parameters of one Envelope gradient step (envelope.py:88-118 constructor arguments). */
typedef struct morl_update_cfg {
    float gamma;
    float homotopy_lambda;   /* 0 => plain MSE (envelope.py:309) */
    float max_grad_norm;     /* < 0 => no clipping (max_grad_norm=None) */
    double lr, beta1, beta2, eps; /* float64 like torch's Python-side scalars (adam.py) */
    int32_t adam_step;       /* 1-based index of the step being taken */
    int32_t envelope;        /* 1: envelope target (envelope.py:404-440), 0: DDQN target (:442-463) */
    int32_t apply_step;      /* 0: stop after gradients (parity tests) */
    /* morl_envelope_update_shard only: */
    int32_t main_forward_done; /* 1: morl_envelope_main_forward has already run for this batch (it can overlap the all-gather) */
    int32_t slab_parts;      /* 0 / 1: qo_all, qt_all are [B][W_total][A][R].  G > 1: they point at part 0 of an all-gathered
                              * buffer [G][2][B][W_total/G][A][R] (what all_gather_into_tensor builds from every rank's
                              * morl_envelope_slabs output): qt_all = qo_all + B*(W_total/G)*A*R, parts 2*B*(W_total/G)*A*R
                              * floats apart -- read in place, no re-layout pass */
    /* prioritized replay (envelope.py:329-334 -> prioritized_buffer.py:187-195), optional: per_tree != NULL makes
     * morl_envelope_update apply  priority = (|td . w| + running_max)^per_alpha  to the device sum tree itself (the same
     * arithmetic as morl_sumtree_update), as an extra workgroup of the weight-gradient launch instead of a launch of its
     * own.  per_idx: device int64 [B] sampled leaf indices; per_running_max: device double [1]. */
    double* per_tree;
    const int64_t* per_idx;
    double* per_running_max;
    int32_t per_levels;
    float per_alpha;
    /* batch-axis sharding (a rank updates on its B transitions of a larger batch): the TD rows of the whole job, by which
     * the loss and its gradient are normalised.  0 => B * W (this call is the whole batch). */
    int64_t rows_total;
    /* morl_envelope_update_shard only -- lazy target evaluation of the weight-sharded step (envelope.py:422-439: the arg-max reads
     * the ONLINE slab alone, the target network only at each TD row's (j*, a*)): both non-NULL => qo_all is the all-gathered
     * ONLINE slab [G][B][W_total/G][A][R] (morl_envelope_slab_online of every rank; parts B*(W_total/G)*A*R floats apart when
     * slab_parts = G > 1), qt_all is ignored, and the rank evaluates the target network itself on the distinct (transition,
     * weight) pairs its own TD rows selected -- any of the W_total weights -- from shard_next_obs [B][D] and weights_all. */
    const float* shard_params_target;
    const float* shard_next_obs;
} morl_update_cfg;

/* Optional device outputs of morl_envelope_update (any may be NULL). */
typedef struct morl_update_out {
    float* loss;          /* [1]   critic loss (envelope.py:307-313)                               */
    float* grad_norm;     /* [1]   total L2 norm before clipping (envelope.py:324-325)               */
    float* priority;      /* [B]   |td . w| of the weight-0 rows (envelope.py:330-331)               */
    float* target;        /* [W*B][R] envelope / DDQN target, row r = i*B + b (envelope.py:293-297)  */
    int32_t* pref;        /* [W*B] arg-max weight index j* (envelope.py:426), envelope only          */
    int32_t* ac;          /* [W*B] arg-max action a* (envelope.py:424 / :455)                        */
    float* q_online_next; /* [B][W][A][R] Q_online(s'_b, w_j)  (de-duplicated rows)                  */
    float* q_target_next; /* [B][W][A][R] Q_target(s'_b, w_j)                                        */
    float* q_values;      /* [W*B][A][R] Q_online(s_b, w_i), row r = i*B + b (envelope.py:300)       */
} morl_update_out;

const char* morl_last_error(void);
int morl_abi_version(void);
/* 1 when the library was built by hipcc for gfx950, 0 for the host-emulated test build (tests/hipsim). */
int morl_is_device_build(void);

/* ---- context ------------------------------------------------------------------------------ */
int morl_ctx_create(morl_ctx** out, const morl_net_desc* net, int max_batch, int max_weights);
int morl_ctx_destroy(morl_ctx* ctx);
/* Select the MLP engine: 0 = per-layer GEMMs, 1 = layer-fused chain (row tile chosen per launch),
 * 2 / 3 = layer-fused chain with the row tile forced to 64 / 32.  The fused engine needs widths <= 256 and hidden
 * widths % 4 == 0.  Returns the engine now active (0..3), < 0 on error. */
int morl_ctx_set_fused(morl_ctx* ctx, int enable);
/* Weight-gradient engine: 0 = wave-level tiles streaming both operands HBM -> registers (dw_wave.h), 1 = 128x128
 * double-buffered LDS tiles, two workgroups per CU, 2 = the single-buffered tiles of the per-layer engine, 3 = per-problem wave
 * layouts with a three-stage operand pipeline (dw_tiles.h; default). */
int morl_ctx_set_dw_mode(morl_ctx* ctx, int mode);
/* Per-launch timing of the dominant kernel (the layer-fused MLP chain): every = n > 0 brackets the chain launches of every
 * n-th Envelope step (counted from this call; the first one is timed) with HIP event pairs on the caller's stream, 0 turns it
 * off, -1 brackets ONE chain launch of every step (the launches of a step take turns, so that every kind of launch is sampled
 * equally often at a quarter of the event records), -2 the same on every second step.  An event record costs a few microseconds of stream time, so sampling keeps the measurement from perturbing the step
 * it measures.  morl_ctx_read_timing blocks until the recorded launches have finished, returns their number and summed
 * duration (ms) and clears the record.  Used by bench.py for the roofline figure. */
int morl_ctx_set_timing(morl_ctx* ctx, int every);
int morl_ctx_read_timing(morl_ctx* ctx, int* n_launches, double* total_ms);
/* The same record split by launch kind: n_launches / total_ms are arrays of MORL_TIMED_KINDS entries -- the forward chain launch
 * (the step's three passes, or the slabs / training-forward launches of a sharded step), the backward-dX chain launch and the
 * weight-gradient launch (dw_tiles; bracketed too, so in the rotating mode the three take turns).  morl_ctx_read_timing returns
 * the two chain kinds summed.  Either call clears the record. */
#define MORL_TIMED_FORWARD 0     /* three passes in the launch (or the launches of a sharded step) */
#define MORL_TIMED_BACKWARD 1
#define MORL_TIMED_DW 2
#define MORL_TIMED_FORWARD2 3    /* the two-pass forward launch of a lazily evaluated step (online next-state + training pass) */
#define MORL_TIMED_KINDS 4
int morl_ctx_read_timing_kinds(morl_ctx* ctx, int* n_launches, double* total_ms);
/* Parity-test aid: out [rows][dims[layer]] <- the post-ReLU activations of hidden layer `layer` (1 .. n_layers - 1) that the last
 * training forward on this context saved for the weight-gradient GEMM; out > 0 is the ReLU mask its backward pass applied.
 * tests/flip_aware.py feeds these masks to the oracle, so that a unit whose pre-activation is within rounding of zero (and lands
 * on the other side of the ReLU than in torch's GEMM) is accounted for exactly instead of by a blanket tolerance. */
int morl_ctx_debug_hidden(morl_ctx* ctx, int layer, int rows, float* out, void* stream);
/* number of float parameters of `net` in the flat layout */
int64_t morl_param_count(const morl_net_desc* net);

/* ---- replay buffer: common/buffer.py:68-96 (gather part of ReplayBuffer.sample) -------------
 * records: device, row-major [capacity][record_floats] with one transition per row laid out as
 *   obs[D] | next_obs[D] | reward[R] | done[1] | action[Ad]   (all float; record_floats = 2D+R+1+Ad)
 * idx: device int64 [B].  Outputs are device arrays shaped as the reference returns them; actions are written as
 * float [B][Ad] and / or int32 [B][Ad] (either pointer may be NULL, not both). */
int morl_gather_batch(const float* records, int record_floats, int64_t capacity, const int64_t* idx, int B,
                      int D, int R, int action_dim, float* obs, float* next_obs, float* rewards, float* dones,
                      float* actions_f, int32_t* actions_i, void* stream);

/* ---- one training batch in one launch: PrioritizedReplayBuffer.sample (common/prioritized_buffer.py:149-185: SumTree.sample
 * :30-54 + the gathers) or ReplayBuffer.sample (common/buffer.py:68-96) ------------------------------------------------------
 * tree != NULL: idx[b] = sum-tree descent with u01[b] (float64, reference order); tree == NULL: idx[b] = idx_in[b].
 * idx_out (device int64 [B], may be NULL) receives the indices; the record gather is that of morl_gather_batch.
 * u01 / idx_in / aux_src may point to PINNED HOST memory mapped into the device address space (see morl_host_device_pointer):
 * they are read inside the kernel instead of by separate copy launches.  aux_src != NULL: aux_floats floats are also moved
 * aux_src -> aux_dst (device) -- the step's sampled weight vectors ride along (envelope.py:281-283). */
int morl_sample_gather(const double* tree, int n_levels, const double* u01, const int64_t* idx_in, const float* records,
                       int record_floats, int64_t capacity, int B, int D, int R, int action_dim, float* obs,
                       float* next_obs, float* rewards, float* dones, float* actions_f, int32_t* actions_i,
                       int64_t* idx_out, const float* aux_src, float* aux_dst, int aux_floats, void* stream);
/* The whole prologue of an Envelope step in one launch: morl_sample_gather's work beside the K-major shadow copies of both
 * networks' weights that the layer-fused kernels stream (made for exactly these two parameter buffers; the step's
 * morl_envelope_update / morl_envelope_slabs on the same context then skips its own shadow-weight launch -- one-shot, dropped by
 * every optimiser step of the library).  Arguments after params_target as morl_sample_gather. */
int morl_envelope_prepare(morl_ctx* ctx, const float* params_online, const float* params_target, const double* tree,
                          int n_levels, const double* u01, const int64_t* idx_in, const float* records, int record_floats,
                          int64_t capacity, int B, int D, int R, int action_dim, float* obs, float* next_obs, float* rewards,
                          float* dones, float* actions_f, int32_t* actions_i, int64_t* idx_out, const float* aux_src,
                          float* aux_dst, int aux_floats, void* stream);
/* Lazy target evaluation (default on; MORL_LAZY_TARGETS=0): envelope_target (envelope.py:404-440) evaluates the TARGET network on
 * every (s'_b, w_j) row, but a TD row (i, b) only reads it at its own arg-max (j*, a*) -- and the W rows of a transition agree on a
 * handful of j* (1 546 distinct (b, j*) pairs of 16 384 at the flagship shape).  morl_envelope_update on the layer-fused engine
 * therefore runs: forward launch with the online next-state pass and the training pass -> arg-max over the online slab, the
 * distinct (b, j*) pairs taking compact rows in the same launch -> the target network on THOSE rows (16-row tiles, device-side
 * row count) -> TD target / loss gradient.  Same values as the eager form (a row's Q does not depend on which rows share its
 * tile); the eager form runs when the caller asks for out->q_target_next, for the DDQN target, on the
 * per-layer engine and for small steps (latency-bound: measured slower lazily): fewer than 4 096 TD rows on the bf16 matrix cores, 8 192
 * on the f32 chains and in the weight-sharded rank step (MORL_LAZY_MIN_ROWS overrides).
 * _set_lazy_targets: 0 always eager, 1 (default) lazy from that row count on, 2 lazy at every size (what smoke() and the small
 * fixtures use to put the default pipeline of large steps under the oracle); returns the previous setting; _lazy_target_rows the rows the last lazy step evaluated -- its distinct pairs
 * (with more than 64 weight vectors a transition's TD rows span several workgroups, and a pair selected from two of them is
 * listed, and evaluated, once per workgroup); 0 if the last step ran eagerly.  Synchronises `stream`. */
int morl_ctx_set_lazy_targets(morl_ctx* ctx, int enable);
/* Arithmetic of the big launches of morl_envelope_update (steps of >= 4 096 TD rows, a weight-sharded rank's from 8 192 -- MORL_BF_MIN_ROWS -- on networks whose hidden
 * layers are 256 wide, head <= 32 columns, input <= 64): by default the two ONLINE forward passes (QNet.forward, envelope.py:300,
 * :420) and the dX half of the backward pass (:323) run on the bf16 matrix cores, every fp32 product evaluated as six products of
 * three-way bf16 splits accumulated in fp32 (csrc/mlp_chain_bf.h: exact split of every finite fp32, fp32-class result: max
 * |Q - Q_float64| 2.4e-7 against 2.3e-7 for the k-ordered fp32 chain) -- the f32-input MFMA runs at 1/16 of the bf16 rate.
 * enable = 1 (or MORL_EXACT_F32=1) keeps every GEMM on the f32-input MFMA (rounds 1-3; the A/B leg).  The target network's rows,
 * the weight gradients and every other entry point are f32-input MFMA in both settings.  Returns the previous setting. */
int morl_ctx_set_exact_f32(morl_ctx* ctx, int enable);
/* What the last morl_envelope_update on this context ran on (what bench.py prices its roofline against): bit 0 = forward passes and
 * dX backward as split-bf16 products, bit 1 = the weight gradients too; 0 = everything on the f32-input MFMA.  Bit 2: its lazily
 * evaluated target rows ran on the 64-row f32 tiles instead of the 8-row ones -- the adaptive fall-back for batches whose TD
 * rows select many (transition, weight) pairs: the launch of lazily evaluated step e is sized by the pair count step e - 8
 * reported (more than MORL_LAZY_BIG_ROWS = 6 144 -> large tiles; same compact rows, same values), so a worst-case batch (every
 * TD row its own pair) costs what the eager target pass costs, not a 16 384-row pass through 8-row tiles.  Reading that count
 * makes the host wait when it is more than 8 lazily evaluated steps ahead of the device (it is then throttled to the device's
 * pace, as by any bounded queue); morl_ctx_backpressure_seconds returns the host time spent so.  That wait is bounded (2 s): a
 * count that never arrives (a stream parked behind the caller's own events, a failed step) makes THAT step -- and the 32 after it,
 * then the context asks again -- take the 8-row tiles: bit 3 = the last step was sized without its count, bit 4 = this happened
 * at least once on this context (bench.py and the tests assert it did not); bit 5 = the target network of the last step -- its
 * lazily evaluated rows (MORL_BFN_TARGETS=1) or the whole pass of an eagerly evaluated few-row step -- ran on the few-row split-bf16
 * chain (mlp_chain_bfn.h: six split products, fp32-class) rather than on the f32 tiles; bit 6 = the two online forward passes ran as
 * pairs of 64-row tiles sharing every weight fragment (mlp_chain_bf2.h, MORL_BF_DUAL=1: the same bits as the separate tiles); bit 7 =
 * the backward chain ran with rolling epilogues over a pair-major stream (mlp_chain_bf_roll.h, MORL_BF_ROLL=1: the same bits); bit 8 =
 * the backward chain's weight ring was issued by a producer wave (mlp_chain_bf_pw_kernel / mlp_chain_bf32_pw_kernel: the default for
 * one-round 64-row launches and for 32-row launches); bit 9 = the forward launch's too (mlp_chain_bf_fwd_pw_kernel: one-round launches). */
int morl_ctx_last_step_bf16(morl_ctx* ctx);
int morl_ctx_backpressure_seconds(morl_ctx* ctx, double* seconds);
int morl_ctx_lazy_target_rows(morl_ctx* ctx, int* rows, void* stream);
/* The shadow copies made by morl_envelope_prepare are consumed ONLY by the gradient step that directly follows it
 * (morl_envelope_update / morl_envelope_slabs / the one-call sharded steps); every other entry point re-makes its copies.  A
 * caller that writes the parameter buffers in place between prepare and that step (morl_polyak, load_state_dict, copy_) calls
 * this to drop the prepared copies explicitly. */
int morl_ctx_invalidate_shadows(morl_ctx* ctx);
/* device-side address of a pinned (page-locked, mapped) host allocation; MORL_ERR_HIP if the memory is not mapped */
int morl_host_device_pointer(void* host_ptr, void** device_ptr);
/* Same gather for records with arbitrary extra fields (CAPQL ReplayMemory.sample, multi_policy/capql/capql.py:56-63,
 * whose transitions carry the episode's weight vector): outs[f][b][0..widths[f]) = records[idx[b]][offsets[f] ..).
 * offsets / widths / outs are HOST arrays of n_fields (<= 8) entries; outs[f] are device pointers. */
int morl_gather_fields(const float* records, int record_floats, int64_t capacity, const int64_t* idx, int B,
                       int n_fields, const int32_t* offsets, const int32_t* widths, float* const* outs, void* stream);

/* ---- QNet.forward: envelope.py:60-77 ----------------------------------------------------------
 * Evaluates Q(obs_b, w_k) for all B x W pairs.  row_order 0: row = b*W + k ("next-state slab"
 * [B][W][A][R]); row_order 1: row = k*B + b (the reference's tiled TD-row order, envelope.py:284-291).
 * q_out: device [B*W][A*R]. */
int morl_qnet_forward(morl_ctx* ctx, const float* params, const float* obs, const float* weights, int B, int W,
                      int row_order, float* q_out, void* stream);

/* ---- envelope arg-max: envelope.py:422-439 (and DDQN :454-462 when diag_only) -----------------
 * qo, qt: device [B][W][A][R]; weights: device [W][R].  For TD row (i, b):
 * (j*, a*) = first arg-max over (j, a) of w_i . qo[b][j][a][:] (products and sums separately rounded in
 * objective order, as the reference's batched einsum); target[i*B+b][:] = qt[b][j*][a*][:].
 * diag_only = 1 restricts j to i (DDQN).  pref / ac may be NULL. */
int morl_envelope_reduce(const float* qo, const float* qt, const float* weights, int B, int W, int A, int R,
                         int diag_only, float* target, int32_t* pref, int32_t* ac, void* stream);

/* Same arg-max for ARBITRARY rows, i.e. exactly Envelope.envelope_target(obs, w, sampled_w) (envelope.py:404-440):
 * row n has its own scalarisation vector row_weights[n][:] and its own slabs qo[n] / qt[n] ([n_rows][W][A][R],
 * Q(obs_n, sampled_w_j)).  target [n_rows][R]; pref / ac [n_rows] may be NULL. */
int morl_envelope_reduce_rows(const float* qo, const float* qt, const float* row_weights, int n_rows, int W, int A,
                              int R, float* target, int32_t* pref, int32_t* ac, void* stream);

/* ---- Envelope.max_action (envelope.py:389-402) for n (observation, weight) rows in ONE call: Q(obs_r, w_r) through the
 * network, scalarisation w_r . Q[a] as the fma chain torch's unbatched einsum evaluates to, first arg-max over the actions.
 * obs [n][D], w [n][R], actions_out int32 [n] (all device).  n <= max_batch * max_weights of the context. */
int morl_envelope_greedy_actions(morl_ctx* ctx, const float* params, const float* obs, const float* w, int n,
                                 int32_t* actions_out, void* stream);

/* ---- one Envelope gradient step: envelope.py:269-334 -------------------------------------------
 * All arrays device.  params_online / grads / exp_avg / exp_avg_sq: flat [P]; params_target: flat [P]
 * (read-only).  obs/next_obs [B][D], actions int32 [B], rewards [B][R], dones [B], weights [W][R].
 * Computes targets with the de-duplicated B*W-row formulation (bit-identical to the reference's
 * W^2*B-row formulation, SURVEY.md headline fact 3), MSE (+ homotopy) loss, gradients (written to
 * `grads`, post-clip as torch leaves them), clip_grad_norm_ and torch's single-tensor Adam. */
int morl_envelope_update(morl_ctx* ctx, float* params_online, const float* params_target, float* grads,
                         float* exp_avg, float* exp_avg_sq, const float* obs, const float* next_obs,
                         const int32_t* actions, const float* rewards, const float* dones, const float* weights,
                         int B, int W, const morl_update_cfg* cfg, const morl_update_out* out, void* stream);

/* ---- Envelope.update's whole loop in ONE library entry: `for g in range(self.gradient_updates)` of envelope.py:269-334 -----
 * Everything of a step that does not change from one call to the next lives in a block the caller fills once and keeps
 * (device pointers of the agent's persistent buffers, the replay store, the batch tensors the gather writes and the update reads);
 * a call then only carries what the host draws per iteration -- in the reference's order per generator: the B unit uniforms of
 * PrioritizedReplayBuffer.sample (prioritized_buffer.py:40; `u01` [n][B] doubles) or the B indices of ReplayBuffer.sample
 * (buffer.py:82; `idx_in` [n][B] int64) and the W x R sampled weights (envelope.py:279-283; `w_src` [n][W*R] floats), all
 * DEVICE-VISIBLE addresses (device memory or mapped pinned host memory, read in place by the gather launch) -- plus the index of
 * the first Adam step and the current homotopy weight.  Iteration k = morl_envelope_prepare (tree descent / index read + record
 * gather + the w copy + the step's weight copies) then morl_envelope_update with adam_step0 + k; with a tree, iteration k + 1
 * samples through the priorities iteration k wrote (envelope.py:329-334), exactly as n separate rounds would.
 * loss_out [n], grad_norm_out [n] or NULL, priority_out [B] (|td . w| of the LAST iteration; needed when io->tree != NULL). */
typedef struct morl_step_io {
    float* params_online;        /* flat [P], updated in place */
    const float* params_target;  /* flat [P] */
    float* grads;                /* flat [P] */
    float* exp_avg;              /* flat [P] */
    float* exp_avg_sq;           /* flat [P] */
    double* tree;                /* PER sum tree (levels concatenated root-first), or NULL: uniform replay */
    double* running_max;         /* [1] (with tree) */
    const float* records;        /* [capacity][record_floats] AoS replay records: obs | next_obs | reward | done | action */
    int64_t capacity;
    float* obs;                  /* [B][D]   batch tensors: written by the gather, read by the update */
    float* next_obs;             /* [B][D] */
    float* rewards;              /* [B][R] */
    float* dones;                /* [B] */
    int32_t* actions;            /* [B] */
    int64_t* idx;                /* [B] sampled indices (output) */
    float* weights;              /* [W][R] the step's sampled weights (device copy of w_src's current slice) */
    int32_t n_levels;            /* of the tree */
    int32_t record_floats;
    int32_t D, R;
    int32_t B, W;
    morl_update_cfg cfg;         /* gamma, max_grad_norm, lr, beta1, beta2, eps, envelope, per_alpha; the per-iteration fields
                                  * (adam_step, homotopy_lambda, apply_step, per_* pointers, rows_total) are set by the entry */
} morl_step_io;
int morl_envelope_update_n(morl_ctx* ctx, const morl_step_io* io, int n, const double* u01, const int64_t* idx_in,
                           const float* w_src, int adam_step0, float homotopy_lambda, float* loss_out, float* grad_norm_out,
                           float* priority_out, void* stream);

/* ---- weight-axis sharding of the same step (no counterpart in the reference, which is single-device) ------------
 * A rank that owns the TD rows of weights [i_offset, i_offset + W_local) of W_total:
 *   1. morl_envelope_slabs: Q_online / Q_target(s'_b, w_j) for its W_local weights in one launch pair -> local slabs
 *      [2][B][W_local][A][R]; the caller all-gathers them (cfg->slab_parts describes the gathered layout) and, while that
 *      collective is in flight, runs morl_envelope_main_forward (the training forward does not need the slabs);
 *   2. morl_envelope_update_shard: training forward of its rows, envelope arg-max over ALL W_total candidates, TD,
 *      backward.  `grads` receives this rank's UNCLIPPED contribution, already normalised by the global row count
 *      B * W_total, out->loss its share of the loss (out->priority: |td . w| of weight 0's rows on the rank with
 *      i_offset == 0, zeros on the others, so that the sum over the ranks is the priority; target / pref / ac / q_values are
 *      local [W_local*B] rows);
 *   3. the caller all-reduces (sums) grads and the loss; morl_clip_adam applies clip_grad_norm_ + Adam identically
 *      on every rank. */
/* next-state slabs of this rank: slabs_out [2][B][W_local][A][R] (online, then target network), rows (b, j) of
 * (next_obs_b, weights_local_j) -- envelope.py:416-420 on the de-duplicated rows. */
int morl_envelope_slabs(morl_ctx* ctx, const float* params_online, const float* params_target, const float* next_obs,
                        const float* weights_local, int B, int W_local, float* slabs_out, void* stream);
/* training forward Q_online(s_b, w_i) of this rank's TD rows (envelope.py:300), activations kept in the context for
 * morl_envelope_update_shard(cfg->main_forward_done = 1).  weights_local = this rank's W_local vectors.  (Reuses the
 * transposed weights of this step's morl_envelope_slabs when params_online is the same buffer and no morl_clip_adam /
 * morl_envelope_update ran in between; a caller that rewrites the parameters itself between the two calls must not rely on that.) */
int morl_envelope_main_forward(morl_ctx* ctx, const float* params_online, const float* obs, const float* weights_local,
                               int B, int W_local, void* stream);
/* The lazily evaluated form of step 1: the ONLINE network's slab only, slab_out [B][W_local][A][R] -- half the all-gather, no
 * target pass; params_target: the same launch makes the weight copy the target rows of morl_envelope_update_shard(cfg->
 * shard_params_target) stream.  morl_ctx_shard_lazy tells whether the one-call step (morl_envelope_step_sharded) of B x W_local
 * rows per rank takes this form (the same rule as the unsharded step: from MORL_LAZY_MIN_ROWS = 8 192 rows on, envelope targets
 * only, morl_ctx_set_lazy_targets) -- a staged caller asks it to stay bit-identical to the one-call step. */
int morl_envelope_slab_online(morl_ctx* ctx, const float* params_online, const float* params_target, const float* next_obs,
                              const float* weights_local, int B, int W_local, float* slab_out, void* stream);
int morl_ctx_shard_lazy(morl_ctx* ctx, int B, int W_local, int envelope);
int morl_envelope_update_shard(morl_ctx* ctx, const float* params_online, float* grads, const float* obs,
                               const int32_t* actions, const float* rewards, const float* dones,
                               const float* weights_all, int B, int W_total, int i_offset, int W_local,
                               const float* qo_all, const float* qt_all, const morl_update_cfg* cfg,
                               const morl_update_out* out, void* stream);
int morl_clip_adam(morl_ctx* ctx, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                   const morl_update_cfg* cfg, float* grad_norm_out, void* stream);

/* ---- the collectives of the sharded step (the reference has no counterpart: common/morl_algorithm.py:42
 * places everything on one device).  One process per GPU, RCCL over xGMI, bound at run time (an instance the process already
 * loaded -- PyTorch's -- is reused).  Nothing synchronises the host.
 *   morl_comm_unique_id   rank 0 draws the 128-byte id and hands it to the other ranks (any side channel)
 *   morl_comm_init        every rank, collectively; blocks until all `world` ranks joined (current HIP device = the rank's GPU).
 *                         world == 1 with an all-zero id makes a loopback communicator that never loads RCCL
 *   morl_allgather_q_begin  all-gather of the ranks' morl_envelope_slabs outputs into recv [world][count_per_rank], issued on
 *                         the communicator's own stream behind everything already enqueued on `stream`; what the caller
 *                         enqueues on `stream` afterwards (morl_envelope_main_forward) runs beside the exchange
 *   morl_comm_wait        `stream` waits for that all-gather (before morl_envelope_update_shard)
 *   morl_allreduce_grads  in-place sum over the ranks of the flat [gradient | loss | priorities] buffer, on `stream`
 *   morl_comm_init_custom a communicator over the CALLER's transport instead of RCCL: two call-backs that enqueue the all-gather
 *                         (recv [world][count_per_rank] complete) and the in-place sum in stream order on the stream they are
 *                         handed and return 0.  RCCL is one transport, the one-rank loopback another; torch.distributed is
 *                         bound this way for the gloo CPU tests (world 2 / 4 through the same one-call rank step) and for
 *                         MORL_COMM=torch on the GPU.  `user` is passed back untouched. */
#define MORL_COMM_ID_BYTES 128
typedef struct morl_comm morl_comm;
typedef int (*morl_allgather_fn)(void* user, const float* send, float* recv, int64_t count_per_rank, void* stream);
typedef int (*morl_allreduce_fn)(void* user, float* buf, int64_t count, void* stream);
int morl_comm_unique_id(void* id_out);
int morl_comm_init(morl_comm** out, const void* unique_id, int rank, int world);
int morl_comm_init_custom(morl_comm** out, int rank, int world, morl_allgather_fn allgather, morl_allreduce_fn allreduce,
                          void* user);
/* Single-hop transport over peer-mapped memory (hipIpc): the step's messages are latency-bound, so instead of a
 * ring every rank writes its contribution straight into the memory of the rank that needs it, all links at once -- all-gather =
 * push + collect, all-reduce = push (reduce-scatter) + reduce in rank order + pull (csrc/morl_comm.hip).  Two-phase set-up, at most
 * 8 ranks: _create allocates this rank's shared region (sized for all-reduces of max_allreduce_floats and all-gathers of
 * max_allgather_floats per rank) and returns its 64-byte handle; the caller all-gathers the world's handles through any side
 * channel; _connect maps the peers.  Waits are bounded (3 s).  A wait that runs out is never silent: it sets an error word that (a)
 * the clip + Adam launch of the one-call sharded steps reads on the device -- the optimiser state and the PER tree are left alone
 * for that step --, (b) morl_comm_poll reads through a host-mapped mirror WITHOUT synchronising (call it before every step) and (c)
 * morl_comm_check reads after a device synchronisation.  The region is fine-grained (uncached) device memory; if that cannot be
 * allocated _create fails (no fall-back to coarse-grained memory, where a peer's writes need not be visible to a running kernel). */
#define MORL_COMM_IPC_HANDLE_BYTES 64
int morl_comm_ipc_create(morl_comm** out, int rank, int world, int64_t max_allreduce_floats, int64_t max_allgather_floats,
                         void* handle_out);
int morl_comm_ipc_connect(morl_comm* comm, const void* all_handles);
int morl_comm_check(morl_comm* comm);
int morl_comm_poll(morl_comm* comm);
int morl_comm_destroy(morl_comm* comm);
int morl_comm_size(const morl_comm* comm, int* rank, int* world);
int morl_allgather_q_begin(morl_comm* comm, const float* send, float* recv, int64_t count_per_rank, void* stream);
int morl_comm_wait(morl_comm* comm, void* stream);
int morl_allreduce_grads(morl_comm* comm, float* buf, int64_t count, void* stream);
/* The whole sharded step of one rank in ONE call (what distributed.py's update() runs per gradient step): slabs of its
 * W_local weights -> all-gather on the communicator's stream, beside the training forward of its rows -> TD / backward /
 * weight gradients of its rows -> all-reduce of grads_x = [P gradient | 1 loss | B priorities] -> clip_grad_norm_ + Adam
 * (cfg->apply_step, lr, betas, eps, adam_step, max_grad_norm) -> the PER priority update from the summed priorities when
 * cfg->per_tree is set.  A strong-scaled rank's kernels take ~20 us each, so the host work between them is what bounds the
 * step: one library entry instead of seven keeps the interpreter out of it.  slab_local [2][B][W_local][A][R] is this rank's
 * send buffer, slab_all [W_total / W_local][2][B][W_local][A][R] the gathered slabs (read in place); n_params = P.  After the
 * call grads_x[P] is the job's loss and grads_x[P + 1 ...] the B priorities (defined on every rank).  The communicator has
 * W_total / W_local ranks; a communicator of ONE rank with W_local < W_total runs the step of that one rank of the larger
 * job alone (a measurement aid for single-GPU boxes: the other parts of slab_all are read as the caller left them). */
int morl_envelope_step_sharded(morl_ctx* ctx, morl_comm* comm, float* params_online, const float* params_target,
                               float* grads_x, int64_t n_params, float* exp_avg, float* exp_avg_sq, const float* obs,
                               const float* next_obs, const int32_t* actions, const float* rewards, const float* dones,
                               const float* weights_all, int B, int W_total, int i_offset, int W_local, float* slab_local,
                               float* slab_all, const morl_update_cfg* cfg, void* stream);

/* The other way to shard the same step: over the BATCH axis.  Every rank draws the same B_total transitions and W weight
 * vectors, rank r keeps transitions [b_offset, b_offset + B) and runs the WHOLE unsharded pipeline on them (all W weights: the
 * envelope arg-max of a TD row only looks at slabs of its own transition, so nothing has to be gathered), normalised by the
 * job's B_total * W rows; ONE collective -- the all-reduce of grads_x = [P gradient | 1 loss | B_total priorities] (a rank
 * writes the priorities of its own transitions, zeros elsewhere) -- then clip + Adam and the PER update from the B_total summed
 * priorities.  The arrays are the rank's slices ([B] rows); cfg->per_idx the B_total sampled leaves.  Compared with the
 * weight-axis form above: no all-gather, the three forward passes stay one launch, the same rows per rank.  A communicator of
 * one rank with B < B_total runs one rank of the larger job alone (measurement aid, as above). */
int morl_envelope_step_batch_sharded(morl_ctx* ctx, morl_comm* comm, float* params_online, const float* params_target,
                                     float* grads_x, int64_t n_params, float* exp_avg, float* exp_avg_sq, const float* obs,
                                     const float* next_obs, const int32_t* actions, const float* rewards, const float* dones,
                                     const float* weights, int B, int B_total, int b_offset, int W,
                                     const morl_update_cfg* cfg, void* stream);

/* One rank's WHOLE sharded Envelope.update iteration in one entry, sampling included: morl_envelope_prepare (tree descent /
 * index read + record gather + the w copy + the step's weight copies, into io's batch tensors) followed by
 * morl_envelope_step_batch_sharded (axis 0: this rank keeps transitions [offset, offset + share) of io->B) or
 * morl_envelope_step_sharded (axis 1: weights [offset, offset + share) of io->W; slab_local / slab_all as there).  `io` is the
 * persistent block of morl_envelope_update_n with io->grads = grads_x = [P gradient | 1 loss | io->B priorities] (the buffer the
 * all-reduce sums); u01 / idx_in / w_src as there, for one iteration.  What the sharded agents of distributed.py call per step. */
int morl_envelope_rank_step(morl_ctx* ctx, morl_comm* comm, const morl_step_io* io, int axis, int offset, int share,
                            const double* u01, const int64_t* idx_in, const float* w_src, int adam_step, float homotopy_lambda,
                            float* slab_local, float* slab_all, void* stream);

/* ---- polyak_update: common/networks.py:120-139 ------------------------------------------------- */
int morl_polyak(const float* src, float* dst, float tau, int64_t n, void* stream);

/* ---- get_non_pareto_dominated_inds: common/pareto.py:34-57 --------------------------------------
 * points: device float64 [N][R]; mask_out: device uint8 [N] (1 = keep).  Bit-exact boolean result.
 * remove_duplicates: bit 0 as the reference's flag; bit 1 (measurement aid of bench_front.py): run without the wave-uniform
 * early exits, i.e. execute all N^2 pair tests -- same mask. */
int morl_pareto_mask(const double* points, int N, int R, int remove_duplicates, uint8_t* mask_out, void* stream);

/* ---- front metrics: common/performance_indicators.py -------------------------------------------------------------
 * hypervolume(ref_point, points) (:15-25; the reference negates both and calls pymoo's minimisation HV -- the quantity
 * is the volume dominated by the points and dominating ref_point) and expected_utility(front, weights_set) (:71-91,
 * utility = dot).  All arrays are device float64: points / front [N][R], ref_point [R], weights [M][R]; results are
 * one device double each.  workspace: morl_metrics_workspace_doubles(N, R) doubles (the larger of the two calls' needs;
 * M <= 4096 weight vectors).  Exact up to the rounding of a fixed-order fp64 sum; fronts of up to 512 points are staged in
 * LDS, larger ones are read in place; N^R point tests above 4e11 are refused. */
int64_t morl_metrics_workspace_doubles(int N, int R);
int morl_hypervolume(const double* points, int N, int R, const double* ref_point, double* workspace, double* hv_out,
                     void* stream);
int morl_expected_utility(const double* front, int N, int R, const double* weights, int M, double* workspace,
                          double* eum_out, void* stream);

/* ---- PER sum-tree: common/prioritized_buffer.py:12-82 (device-resident) -------------------------
 * tree: device float64, levels concatenated root first; level l has 2^l nodes and starts at offset
 * 2^l - 1; n_levels = ceil(log2(capacity)) + 1.
 * sample:  idx[k] = descent for query 0 + (root - 0) * u01[k]          (SumTree.sample :30-54)
 * set:     sequential tree.set(ptr[k], value[k]) (value < 0 => use *running_max)  (SumTree.set :56-67,
 *          PrioritizedReplayBuffer.add :126-147)
 * update:  running_max = max(running_max, max(pr)); batch_set(idx, pr) keeping the first occurrence of
 *          duplicated indices and adding in ascending index order (:69-82, :187-195).
 *          pr[k] = powf(raw[k] + (float)running_max_before, alpha)  (envelope.py:333, numpy float32 arithmetic);
 *          alpha < 0: pr[k] = raw[k] (the caller already formed the priorities). */
int morl_sumtree_sample(const double* tree, int n_levels, const double* u01, int B, int64_t* idx, void* stream);
int morl_sumtree_set(double* tree, int n_levels, const int64_t* ptr, const double* value, int n,
                     double* running_max, void* stream);
/* priority = max(raw, clamp_min) ** alpha (GPIPD.update: priority.clip(min=self.min_priority) ** self.alpha,
 * multi_policy/gpi_pd/gpi_pd.py:507-526), then as morl_sumtree_update */
int morl_sumtree_update_clamped(double* tree, int n_levels, const int64_t* idx, const float* raw, int B, double alpha,
                                double clamp_min, double* running_max, double* pr_out, void* stream);
int morl_sumtree_update(double* tree, int n_levels, const int64_t* idx, const float* raw, int B, double alpha,
                        double* running_max, double* pr_out, void* stream);

/* ================================================================================================
 * Continuous-action actor-critic updates: CAPQL, MOSAC (the MORL/D subproblem learner) and
 * GPI-PD / GPI-LS with continuous actions (TD3 style).
 *
 * Replaces the arithmetic of
 *   CAPQL.update                       multi_policy/capql/capql.py:321-349        (+ Policy :100-158, QNetwork :161-173)
 *   MOSAC.update                       single_policy/ser/mosac_continuous_action.py:430-489 (+ actor :60-123, critic :28-57)
 *   MORLD.__update_others              multi_policy/morld/morld.py:423-433  (population > 1: every subproblem's
 *                                      MOSAC.update advanced by the same launches)
 *   GPIPDContinuousAction.update       multi_policy/gpi_pd/gpi_pd_continuous_action.py:373-434 (+ Policy :34-58,
 *                                      QNetwork :61-73 with LayerNorm / Dropout from common/networks.py:10-48)
 *
 * A context advances `population` independent learners of identical shape per call; every array below has
 * the learner index as its slowest axis ([pop] ...).  population == 1 for CAPQL / GPI-PD / a single MOSAC.
 *
 * Flat parameter layouts (fp32; the torch nn.Parameters of the host classes are views into them):
 *   Q-net   in = D + Ad (+ R when the critic is weight-conditioned: CAPQL, TD3), for each hidden layer l:
 *           W_l [h_l][in_l], b_l [h_l], then LayerNorm gamma_l [h_l], beta_l [h_l] when q_layer_norm; finally
 *           W_out [R][h_last], b_out [R].                          = nn.Sequential.parameters() order.
 *   policy  in = D (+ R when weight-conditioned), hidden layers W_l, b_l (no LayerNorm), then the heads as ONE
 *           linear layer: W_head [heads*Ad][h_last], b_head [heads*Ad]; heads = 2 (rows 0..Ad-1 = mean, rows
 *           Ad..2Ad-1 = log_std) for CAPQL / MOSAC, heads = 1 (mean) for TD3.
 *   SACD    critics: in = D, out = A * R (Q of every action); actor: in = D, out = A logits.  batch.actions holds the
 *           taken action index as a float ([pop][rows], ReplayBuffer's float32 action column); batch.w is [pop][R];
 *           no noise inputs (the expectation over actions is exact); the actor is updated once per call.
 *   critics of a learner are contiguous: q[pop][num_q][Pq]; one Adam state over all of them (the reference
 *   chains the Q-nets' parameters into one optimiser).
 * Random draws are INPUTS (device arrays the host fills from its generator): re-parameterisation noise eps,
 * TD3 target-policy noise.  Dropout keep-masks are generated on the device from cfg.dropout_seed unless explicit
 * masks are passed (parity tests).
 * ================================================================================================ */
#define MORL_AC_CAPQL 0
#define MORL_AC_MOSAC 1
#define MORL_AC_TD3 2
#define MORL_AC_SACD 3   /* MOSAC with discrete actions: single_policy/ser/mosac_discrete_action.py:440-503 */

typedef struct morl_ac_ctx morl_ac_ctx;

typedef struct morl_ac_desc {
    int32_t algo;                    /* MORL_AC_* */
    int32_t obs_dim, act_dim, reward_dim;   /* SACD: act_dim = number of discrete actions */
    int32_t n_hidden;                /* len(net_arch), 1 .. MORL_MAX_LAYERS-1 */
    int32_t hidden[MORL_MAX_LAYERS]; /* net_arch, shared by the policy and the Q-nets (as in the reference) */
    int32_t num_q;                   /* Q-networks per learner (reference default 2; MOSAC: exactly 2) */
    int32_t q_layer_norm;            /* LayerNorm(eps 1e-5, affine) after every hidden Linear of the Q-nets */
    float q_drop_rate;               /* Dropout p after every hidden Linear of the Q-nets (before LayerNorm) */
    int32_t population;              /* learners advanced per call */
    int32_t max_rows;                /* largest number of batch rows of one learner (2 * batch_size for GPI-PD) */
} morl_ac_desc;

/* see morl_ac_cfg.grad_hook */
typedef int (*morl_grad_hook)(void* user, int which, float* grads, int64_t count, void* stream);

typedef struct morl_ac_cfg {
    float gamma, tau;
    float alpha;                     /* entropy coefficient when it is not learnt (CAPQL; MOSAC autotune = 0) */
    double q_lr, policy_lr, alpha_lr;
    double beta1, beta2, eps;        /* torch.optim.Adam defaults 0.9 / 0.999 / 1e-8 */
    int32_t q_step;                  /* 1-based Adam step of the critic optimiser taken by this call */
    int32_t policy_step;             /* 1-based Adam step of the FIRST actor iteration of this call */
    int32_t do_policy;               /* run the actor update (delayed policy updates) */
    int32_t policy_iters;            /* MOSAC: policy_freq inner actor iterations; otherwise 1 */
    int32_t do_target;               /* polyak the target critics (MOSAC: target_net_freq) */
    int32_t autotune;                /* MOSAC: learn log_alpha */
    float target_entropy;            /* MOSAC: -prod(action_shape) */
    float policy_noise, noise_clip;  /* TD3 target policy smoothing */
    int32_t n_per;                   /* TD3: rows whose |TD| priorities are written (0 = none) */
    uint64_t dropout_seed;           /* changes every call */
    /* Data-parallel learners (several processes, each with its own rows of the batch): when grad_hook is set, the critic
     * gradients (which = 0, out->q_grads, required) and the actor gradients (which = 1, out->pol_grads, required when
     * do_policy) are handed to the hook BEFORE their Adam step; the hook reduces them in place over the job (e.g. an RCCL
     * all-reduce AVG enqueued so that `stream` waits for it) and returns 0.  Every process then takes the identical step.
     * Not available with a learnt entropy coefficient (autotune) or the discrete-action SAC. */
    morl_grad_hook grad_hook;
    void* grad_hook_user;
} morl_ac_cfg;

/* caller-owned device state of the learners */
typedef struct morl_ac_state {
    float* q;            /* [pop][num_q][Pq] */
    float* q_target;     /* [pop][num_q][Pq] */
    float* q_exp_avg;    /* [pop][num_q][Pq] */
    float* q_exp_avg_sq;
    float* pol;          /* [pop][Pp] */
    float* pol_exp_avg;
    float* pol_exp_avg_sq;
    float* pol_target;   /* [pop][Pp]  TD3 only */
    float* log_alpha;    /* [pop]  MOSAC autotune only */
    float* log_alpha_exp_avg;
    float* log_alpha_exp_avg_sq;
    const float* action_scale;  /* [Ad]  (high - low) / 2 */
    const float* action_bias;   /* [Ad]  (high + low) / 2 */
    /* optional device-resident Adam step counters (learners of a MORL/D population have taken different numbers of
     * steps): when non-NULL the bias corrections use q_steps[p] + 1 / pol_steps[p] + 1 + iteration instead of
     * cfg.q_step / cfg.policy_step, and the counters are advanced by the call itself. */
    int32_t* q_steps;           /* [pop] */
    int32_t* pol_steps;         /* [pop] */
} morl_ac_state;

typedef struct morl_ac_batch {
    int32_t rows;               /* batch rows per learner */
    int32_t active;             /* learners advanced by this call, 1..population (0 = population); the state / batch /
                                   output pointers then address the FIRST of `active` consecutive learners */
    const float* obs;           /* [pop][rows][D] */
    const float* actions;       /* [pop][rows][Ad] */
    const float* rewards;       /* [pop][rows][R] */
    const float* next_obs;      /* [pop][rows][D] */
    const float* dones;         /* [pop][rows] */
    const float* w;             /* CAPQL / TD3: per-row weights [pop][rows][R]; MOSAC: [pop][R] */
    const float* eps_next;      /* [pop][rows][Ad]  N(0,1): next-action sample (CAPQL / MOSAC), target noise (TD3) */
    const float* eps_pi;        /* [policy_iters][pop][rows][Ad]  (CAPQL / MOSAC) */
    const float* eps_alpha;     /* [policy_iters][pop][rows][Ad]  (MOSAC autotune) */
    const uint8_t* drop_masks;  /* optional explicit keep masks, see morl_ac_mask_bytes(); NULL = device RNG */
} morl_ac_batch;

/* optional device outputs (NULL = not wanted) */
typedef struct morl_ac_out {
    float* critic_loss;   /* [pop]  CAPQL / TD3: (1/num_q) sum_n mse_n ; MOSAC: qf1_loss + qf2_loss */
    float* q_losses;      /* [pop][num_q]  per-critic mse */
    float* policy_loss;   /* [pop]  last actor iteration */
    float* alpha_loss;    /* [pop]  MOSAC autotune, last iteration */
    float* alpha;         /* [pop]  MOSAC: exp(log_alpha) after the call */
    float* priority;      /* [pop][n_per]  TD3: |q_0 - target| * 0.05 . w  (before clip / pow) */
    float* target_q;      /* [pop][rows][R]  (MOSAC: [pop][rows] scalarised) */
    float* q_grads;       /* [pop][num_q][Pq]  critic gradients of this call */
    float* pol_grads;     /* [pop][Pp]  actor gradients of the last iteration */
} morl_ac_out;

int64_t morl_ac_q_param_count(const morl_ac_desc* d);
int64_t morl_ac_policy_param_count(const morl_ac_desc* d);
/* bytes of batch.drop_masks when every learner is active: for phase in (target critics, critics, critics at pi):
 * for net in [active][num_q]: for hidden layer l: rows * h_l keep flags (row-major) */
int64_t morl_ac_mask_bytes(const morl_ac_desc* d, int rows);
int morl_ac_create(morl_ac_ctx** out, const morl_ac_desc* d);
/* Dense-layer engine of the actor-critic path (process-wide tuning knob; results are bit-identical either way):
 * 0 = pick per launch (wave-level 32 x 32 MFMA tiles while the launch has < 128 LDS tiles of 128 x 128, the LDS-tiled
 * engine above that), 1 = always LDS tiles, 2 = always wave-level tiles. */
int morl_ac_set_gemm_mode(int mode);
int morl_ac_destroy(morl_ac_ctx* ctx);

/* One gradient update of every learner (the body of the reference's update() loop), asynchronous on `stream`. */
int morl_ac_update(morl_ac_ctx* ctx, const morl_ac_state* st, const morl_ac_batch* batch, const morl_ac_cfg* cfg,
                   const morl_ac_out* out, void* stream);
/* The reference's `for _ in range(self.gradient_updates)` loop (capql.py:322, mosac_continuous_action.py:431,
 * gpi_pd_continuous_action.py:375 -- GPI-PD runs 20 updates per environment step; the reference's own jitted precedent is
 * multi_policy/gpi_ls_jax/gpi_ls_jax.py:342-477, a fori_loop over the gradient updates) in ONE call: update k consumes
 * batches[k] / cfgs[k] (its Adam step index, dropout seed, do_policy / do_target flags) and writes outs[k] (outs may be NULL).
 * The host draws the n batches (indices, noise) beforehand on the reference's RNG streams; the n x ~20 launches are enqueued by
 * one library entry.  Stops at the first failing update (the message names it); updates before it have been enqueued. */
int morl_ac_update_n(morl_ac_ctx* ctx, const morl_ac_state* st, int n, const morl_ac_batch* batches, const morl_ac_cfg* cfgs,
                     const morl_ac_out* outs, void* stream);

/* Policy forward for acting / evaluation.  obs [pop][rows][D]; w as in morl_ac_batch (NULL for MOSAC);
 * mode 0: deterministic action (CAPQL Policy.get_action, TD3 Policy.forward without noise; MOSAC: the tanh mean),
 * mode 1: sampled action with eps [pop][rows][Ad] (MOSACActor.get_action; CAPQL Policy.sample; TD3: noise).
 * use_target != 0 reads st->pol_target (TD3).  actions_out [pop][rows][Ad]; logp_out [pop][rows] or NULL.
 * SACD: actions_out receives the actor LOGITS [pop][rows][A] (Categorical sampling stays with the caller). */
int morl_ac_policy_forward(morl_ac_ctx* ctx, const morl_ac_state* st, const float* obs, const float* w, int rows,
                           int mode, const float* eps, int use_target, const morl_ac_cfg* cfg, float* actions_out,
                           float* logp_out, void* stream);

/* Critic forward (eval mode: no dropout): q_out [pop][num_q][rows][R] for inputs obs / actions (/ w);
 * SACD: q_out [pop][num_q][rows][A*R], actions ignored. */
int morl_ac_q_forward(morl_ac_ctx* ctx, const morl_ac_state* st, const float* obs, const float* actions,
                      const float* w, int rows, int use_target, float* q_out, void* stream);

/* ================================================================================================
 * GPI-PD / GPI-LS with discrete actions: multi_policy/gpi_pd/gpi_pd.py
 *   QNet (:41-76)            sf = relu(Linear(D, h0)(obs)); wf = relu(Linear(R, h0)(w)); net(sf * wf) with
 *                            net = [Linear, Dropout(p), LayerNorm, ReLU] * (len(arch) - 1) -> Linear(A * R)
 *   GPIPD.update (:416-520)  min-over-ensemble TD target, GPI envelope target over a weight set (gpi_pd),
 *                            non-standard Huber (common/networks.py:90-100), per-net clip, Adam, PER errors
 *   gpi_action / max_action (:564-582, :608-617), _envelope_target (:662-690), _reset_priorities (:619-660)
 * Flat parameter layout of ONE net (= QNet.parameters() order): weights_features.0.weight [h0][R], .bias [h0],
 * state_features.0.weight [h0][D], .bias [h0], then net.* as in morl_ac (W, b[, gamma, beta] per hidden layer, W_out
 * [A*R][h_last], b_out).  The num_nets ensemble members are contiguous: q[num_nets][P]; one Adam state over all.
 * ================================================================================================ */
typedef struct morl_gpi_ctx morl_gpi_ctx;

typedef struct morl_gpi_desc {
    int32_t obs_dim, reward_dim, n_actions;
    int32_t n_hidden;                  /* len(net_arch) >= 2 */
    int32_t hidden[MORL_MAX_LAYERS];   /* net_arch; hidden[0] is the width of the two feature embeddings */
    int32_t num_nets;                  /* ensemble size (reference default 2), 1..4 */
    int32_t layer_norm;
    float drop_rate;
    int32_t max_rows;                  /* most rows of one call (2 * batch_size for update) */
    int32_t max_support;               /* most weight vectors of one envelope target / GPI action */
} morl_gpi_desc;

typedef struct morl_gpi_cfg {
    float gamma;
    float min_priority;                /* the Huber threshold (gpi_pd.py:476) */
    float max_grad_norm;               /* < 0: no clipping; else clip_grad_norm_ of every net separately */
    double lr, beta1, beta2, eps;
    int32_t adam_step;                 /* 1-based */
    int32_t gpi_pd;                    /* also form the envelope target and the gtd errors */
    int32_t n_per;                     /* rows whose PER errors are written */
    int32_t apply_step;                /* 0: gradients only */
    uint64_t dropout_seed;
} morl_gpi_cfg;

typedef struct morl_gpi_out {          /* optional device outputs */
    float* critic_loss;                /* scalar */
    float* td_error;                   /* [n_per]  |w . max_n |psi_n - target||  (before clip / pow) */
    float* gtd_error;                  /* [n_per]  same with the envelope target */
    float* target_q;                   /* [rows][R] */
    float* target_q_envelope;          /* [rows][R] */
    float* grads;                      /* [num_nets][P]  (after clipping) */
    float* grad_norm;                  /* [num_nets]  pre-clip norms (only when clipping) */
} morl_gpi_out;

int64_t morl_gpi_param_count(const morl_gpi_desc* d);
/* keep-mask bytes of one dropout phase over `rows` input rows: for net: for hidden layer l >= 1: rows * h_l flags;
 * morl_gpi_update consumes phases (target nets on rows, target nets on rows * K when gpi_pd, nets on rows) in order */
int64_t morl_gpi_mask_bytes(const morl_gpi_desc* d, int rows);
int morl_gpi_create(morl_gpi_ctx** out, const morl_gpi_desc* d);
int morl_gpi_destroy(morl_gpi_ctx* ctx);
/* the per-update arguments of morl_gpi_update as a struct (morl_gpi_update_n) */
typedef struct morl_gpi_batch {
    const float* obs;                  /* [rows][D] */
    const int32_t* actions;            /* [rows] */
    const float* rewards;              /* [rows][R] */
    const float* next_obs;             /* [rows][D] */
    const float* dones;                /* [rows] */
    const float* w;                    /* [rows][R] */
    const float* sampled_w;            /* [K][R] (gpi_pd) */
    const uint8_t* drop_masks;         /* optional explicit keep masks */
    int32_t rows, K;
} morl_gpi_batch;
/* One gradient update (the loop body of GPIPD.update).  w: per-row weights [rows][R]; sampled_w [K][R] (gpi_pd). */
int morl_gpi_update(morl_gpi_ctx* ctx, float* q, const float* q_target, float* exp_avg, float* exp_avg_sq,
                    const float* obs, const int32_t* actions, const float* rewards, const float* next_obs,
                    const float* dones, const float* w, int rows, const float* sampled_w, int K,
                    const uint8_t* drop_masks, const morl_gpi_cfg* cfg, const morl_gpi_out* out, void* stream);
/* GPIPD.update's `for g in range(self.gradient_updates)` loop (gpi_pd.py:418-419, 20 by default) in one call: update k =
 * morl_gpi_update on batches[k] / cfgs[k] -> outs[k] (outs may be NULL).  See morl_ac_update_n. */
int morl_gpi_update_n(morl_gpi_ctx* ctx, float* q, const float* q_target, float* exp_avg, float* exp_avg_sq, int n,
                      const morl_gpi_batch* batches, const morl_gpi_cfg* cfgs, const morl_gpi_out* outs, void* stream);
/* The same loop with prioritised replay (GPIPD's default, gpi_pd.py:416-420 + 507-526): iteration k samples B transitions
 * through the sum tree with the unit uniforms u01[k][.] (drawn by the host in the reference's order), gathers them into the
 * buffers batches[k] points to (rows [0, B), and again as rows [B, 2 B) when `doubled`: len(weight_support) > 1; the host has
 * filled batches[k].w / sampled_w), runs morl_gpi_update, and writes priority = max(|td|, min_priority) ** alpha of the B
 * transitions back into the tree -- what iteration k + 1 samples through.  outs[k] must carry td_error (or gtd_error when
 * use_gtd); idx [n][B] receives the sampled indices.  Record layout / capacity / D / R / action_dim as morl_sample_gather. */
typedef struct morl_gpi_per {
    double* tree;
    double* running_max;
    const double* u01;                 /* [n][B], device or mapped pinned host */
    const float* records;
    int64_t* idx;                      /* [n][B] out */
    int64_t capacity;
    int32_t n_levels, record_floats, D, R, action_dim, B;
    int32_t doubled, use_gtd;
    float alpha, min_priority;
} morl_gpi_per;
int morl_gpi_update_n_per(morl_gpi_ctx* ctx, float* q, const float* q_target, float* exp_avg, float* exp_avg_sq, int n,
                          const morl_gpi_per* per, const morl_gpi_batch* batches, const morl_gpi_cfg* cfgs,
                          const morl_gpi_out* outs, void* stream);
/* the continuous-action learner's loop (gpi_pd_continuous_action.py:373-417) the same way: float actions, priority =
 * max(outs[k].priority, min_priority) ** alpha; one learner (batches[k].active <= 1); use_gtd is ignored */
int morl_ac_update_n_per(morl_ac_ctx* ctx, const morl_ac_state* st, int n, const morl_gpi_per* per, const morl_ac_batch* batches,
                         const morl_ac_cfg* cfgs, const morl_ac_out* outs, void* stream);
/* Q(obs_row, w_row) of `n_nets` consecutive nets starting at `params`, eval mode (no dropout):
 * q_out [n_nets][rows][A*R].  w_per_row = 0: one weight vector for every row. */
int morl_gpi_q_forward(morl_gpi_ctx* ctx, const float* params, int n_nets, const float* obs, const float* w,
                       int w_per_row, int rows, float* q_out, void* stream);
/* GPI action: arg max_i max_a w . Q_0(obs, a, support_i); M = 0 selects max_action (min over the ensemble at w).
 * action_out / policy_out: device int32 (policy_out may be NULL). */
int morl_gpi_action(morl_gpi_ctx* ctx, const float* q, const float* obs, const float* support, int M, const float* w,
                    int32_t* action_out, int32_t* policy_out, void* stream);
/* GPI actions of n observations at once (the Dyna rollouts, gpi_pd.py:377-387): actions_out[i] = action of
 * arg max_k max_a w . Q_0(obs_i, a, support_k); n * M <= max_rows * max_support.  Eval mode (no dropout). */
int morl_gpi_actions(morl_gpi_ctx* ctx, const float* q, const float* obs, int n, const float* support, int M,
                     const float* w, int32_t* actions_out, void* stream);
/* The same for n (observation, weight) PAIRS -- evaluation episodes of many weight vectors stepped in lock-step
 * (common/evaluation.py:118-144 calls agent.eval once per weight and step): row i acts under w_rows[i].
 * M > 0: GPIPD.gpi_action (gpi_pd.py:564-582) per row over the support; M = 0: GPIPD.max_action (:608-617) per row
 * (element-wise min over the ensemble).  n * max(M, 1) <= max_rows * max_support. */
int morl_gpi_actions_rows(morl_gpi_ctx* ctx, const float* q, const float* obs, const float* w_rows, int n,
                          const float* support, int M, int32_t* actions_out, void* stream);
/* _reset_priorities errors of `rows` transitions: |w . (r + (1-d) gamma max_next - Q_0(s, w)[a])| with max_next the
 * envelope target over `support` (gpi_pd != 0; rows * M <= max_rows * max_support) or the double-Q target. */
int morl_gpi_priorities(morl_gpi_ctx* ctx, const float* q, const float* q_target, const float* obs,
                        const int32_t* actions, const float* rewards, const float* next_obs, const float* dones,
                        int rows, const float* w, const float* support, int M, int gpi_pd, float gamma,
                        float* gtd_out, void* stream);

/* ================================================================================================
 * Probabilistic dynamics ensemble of the Dyna part of GPI-PD: common/model_based/probabilistic_ensemble.py
 *   forward (:88-129)            h = (x - mu) / sigma; [EnsembleLayer, ReLU] * len(arch); EnsembleLayer -> (mean, logvar)
 *                                logvar = max_logvar - softplus(max_logvar - logvar); = min_logvar + softplus(. - min_logvar)
 *   _compute_loss (:156-169)     mean of F.gaussian_nll_loss(mean, y, exp(logvar)) + 0.01 * (sum max_logvar - sum min_logvar)
 *   fit's optimiser step (:206-212, :248-252)  Adam(lr) with per-layer L2 weight decay, max / min_logvar without
 *   _compute_mse_losses (:171-176)
 * Parameters of ONE member in the flat buffer (E members contiguous, [E][Pm]): for every layer W [out][in]
 * (= the reference's W[e] transposed: EnsembleLayer stores [in][out]) then b [out].  max_logvar / min_logvar [out_dim]
 * are shared by the members and live in their own small arrays (with their own Adam moments).
 * ================================================================================================ */
typedef struct morl_ens_ctx morl_ens_ctx;

typedef struct morl_ens_desc {
    int32_t input_dim, output_dim;      /* model output is 2 * output_dim (mean | logvar) */
    int32_t n_hidden;
    int32_t hidden[MORL_MAX_LAYERS];
    int32_t ensemble_size;
    int32_t max_rows;                   /* rows per member of one call */
} morl_ens_desc;

typedef struct morl_ens_cfg {
    double lr, beta1, beta2, eps;
    int32_t adam_step;                  /* 1-based */
    float weight_decay[MORL_MAX_LAYERS];/* L2 coefficient of layer l (W and b), probabilistic_ensemble.py:206 */
} morl_ens_cfg;

int64_t morl_ens_param_count(const morl_ens_desc* d);
int morl_ens_create(morl_ens_ctx** out, const morl_ens_desc* d);
int morl_ens_destroy(morl_ens_ctx* ctx);
/* One optimiser step of fit().  x [E][rows][in], y [E][rows][out] (the members' bootstrap batches); mu / sigma [in]
 * (NULL: no input normalisation); logvar_bounds [2][out] = max_logvar | min_logvar with moments lv_m / lv_v [2][out];
 * loss_out: device scalar (the value _compute_loss returns) or NULL. */
int morl_ens_train_step(morl_ens_ctx* ctx, float* params, float* exp_avg, float* exp_avg_sq, float* logvar_bounds,
                        float* lv_m, float* lv_v, const float* mu, const float* sigma, const float* x, const float* y,
                        int rows, const morl_ens_cfg* cfg, float* loss_out, void* stream);
/* Forward of every member.  x: [rows][in] shared by the members (x_per_member = 0) or [E][rows][in].
 * mean_out / logvar_out [E][rows][out] (logvar already bounded; logvar_out may be NULL). */
int morl_ens_forward(morl_ens_ctx* ctx, const float* params, const float* logvar_bounds, const float* mu,
                     const float* sigma, const float* x, int x_per_member, int rows, float* mean_out, float* logvar_out,
                     void* stream);
/* Per-member holdout MSE (_compute_mse_losses): x [rows][in] shared, y [rows][out]; mse_out [E]. */
int morl_ens_mse(morl_ens_ctx* ctx, const float* params, const float* logvar_bounds, const float* mu, const float* sigma,
                 const float* x, const float* y, int rows, float* mse_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MORL_HIP_H */
