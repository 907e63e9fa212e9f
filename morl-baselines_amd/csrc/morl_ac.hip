// libmorl_hip.so, second translation unit: the continuous-action actor-critic updates (include/morl_hip.h,
// "Continuous-action actor-critic updates").  Host side: validates, owns the activation workspace ("tapes"),
// enqueues the batched kernels of ac_kernels.h on the caller's stream.  Nothing here synchronises the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "morl_hip.h"
#include "morl_host.h"
#include "morl_device.h"
#include "gemm_f32.h"
#include "mlp_chain2.h"
#include "mlp_chain16.h"
#include "optim_kernels.h"
#include "ac_kernels.h"

using namespace morl;
using morl_host::dmalloc;
using morl_host::fail;
using morl_host::round_up;
using morl_host::stream_grid;
using morl_host::vec_ok;

namespace {

// mlp() of common/networks.py:10-48 as a parameter map
struct Mlp {
    int L = 0;                                   // linear layers
    int dims[MORL_MAX_LAYERS + 1] = {};
    int ld[MORL_MAX_LAYERS + 1] = {};            // row stride of the activation at each level (multiple of 4)
    bool ln = false;
    float drop = 0.f;
    int64_t offW[MORL_MAX_LAYERS] = {}, offB[MORL_MAX_LAYERS] = {}, offG[MORL_MAX_LAYERS] = {};
    int64_t P = 0;

    void finish() {
        int64_t o = 0;
        for (int l = 0; l < L; ++l) {
            offW[l] = o; o += (int64_t)dims[l + 1] * dims[l];
            offB[l] = o; o += dims[l + 1];
            if (ln && l < L - 1) { offG[l] = o; o += 2 * (int64_t)dims[l + 1]; }
        }
        P = o;
        for (int l = 0; l <= L; ++l) ld[l] = round_up(dims[l], 4);
    }
};

// activations of one batched forward pass (G nets x cap rows), kept for the backward
struct Tape {
    int G = 0, xG = 0;
    int cap = 0;                                 // rows capacity (row-block stride of every buffer)
    float* x = nullptr;                          // [xG][cap][ld0]
    float* h[MORL_MAX_LAYERS] = {};              // post-ReLU activations of the hidden layers
    float* zx[MORL_MAX_LAYERS] = {};             // Linear output -> xhat (LayerNorm / Dropout nets only)
    float* rstd[MORL_MAX_LAYERS] = {};
    unsigned long long* mask[MORL_MAX_LAYERS] = {};      // Dropout keep bits, [G][cap][ceil(N / 64)] words per hidden layer
    float* dh[MORL_MAX_LAYERS] = {};             // dLoss/dh of the hidden layers as the layer-fused backward of a LayerNorm net leaves
                                                 // it for ac_ln_grad_kernel (the per-layer path reads it from g[] before the post-op)
    float* out = nullptr;                        // [G][cap][ld[L]]
    float* g[MORL_MAX_LAYERS] = {};              // dLoss/dz of every linear layer
    float* dx = nullptr;                         // [G][cap][ld0]
    // layer-fused passes (mlp_chain2.h): (h[l] > 0) packed by the chain's tiling, [G][ceil(cap / 64)][256] words per layer
    unsigned long long* bits[MORL_MAX_LAYERS] = {};
    bool bits_valid = false;                     // written by the last forward of this tape
};

}  // namespace

struct morl_ac_ctx {
    morl_ac_desc d{};
    Mlp q, pol;
    int heads = 2, QG = 0, PG = 0, cap = 0;
    bool w_input = true;                         // weight vector is a network input (CAPQL, TD3)
    Tape tq_a, tq_b, tp_a, tp_b;
    float* act = nullptr;                        // [PG][cap][Ad]
    float* logp_next = nullptr;                  // [PG][cap]
    float* logp_pi = nullptr;
    float* act_pi = nullptr;                     // [PG][cap][Ad] (a ~ pi(s) when its head runs next to pi(s')'s)
    float* xq_b = nullptr;                       // tq_b.x as allocated; xq_pi: the critics' input rows (s, pi(s), w) of the actor phase
    float* xq_pi = nullptr;                      //   when pi(s) is sampled before the critic phase has finished with (s, a, w)
    float* save_y = nullptr;                     // [PG][cap][Ad]
    float* save_std = nullptr;
    float* gq = nullptr;                         // [QG][Pq]
    float* gp = nullptr;                         // [PG][Pp]
    float* alpha_dev = nullptr;                  // [PG]
    float* adam_corr = nullptr;                  // [2][PG][2] bias-correction scalars of the critics' / the actor's fused Adam step
    // K-major shadow copies of the weight matrices used by the forward GEMMs of an update (ac_kernels.h: MlpLayout)
    float* wt_q = nullptr;                       // [QG][Pq]
    float* wt_qt = nullptr;                      // [QG][Pq]  target critics
    float* wt_pol = nullptr;                     // [PG][Pp]
    float* wt_polt = nullptr;                    // [PG][Pp]  target actor (TD3)
    std::vector<void*> allocs;
};

static int alloc_f(std::vector<void*>& allocs, float** p, size_t n) {
    int rc = dmalloc((void**)p, std::max<size_t>(n, 4) * sizeof(float));
    if (rc) return rc;
    allocs.push_back(*p);
    return hipMemsetAsync(*p, 0, std::max<size_t>(n, 4) * sizeof(float), nullptr) == hipSuccess
               ? MORL_OK : fail(MORL_ERR_HIP, "hipMemsetAsync failed");
}
static int alloc_f(morl_ac_ctx* c, float** p, size_t n) { return alloc_f(c->allocs, p, n); }

static int alloc_tape(std::vector<void*>& c, const Mlp& m, Tape& t, int G, int xG, int cap_rows, bool post) {
    int rc;
    const size_t cap = (size_t)cap_rows;
    t.G = G; t.xG = xG; t.cap = cap_rows;
    if ((rc = alloc_f(c, &t.x, (size_t)xG * cap * m.ld[0]))) return rc;
    if ((rc = alloc_f(c, &t.dx, (size_t)G * cap * m.ld[0]))) return rc;
    for (int l = 0; l < m.L; ++l) {
        const size_t n = (size_t)G * cap * m.ld[l + 1];
        if ((rc = alloc_f(c, &t.g[l], n))) return rc;
        if (l == m.L - 1) { if ((rc = alloc_f(c, &t.out, n))) return rc; }
        else {
            if ((rc = alloc_f(c, &t.h[l], n))) return rc;
            {
                float* bw = nullptr;     // 64-bit words, allocated as pairs of floats
                if ((rc = alloc_f(c, &bw, (size_t)G * ((cap + 63) / 64) * CH_THREADS * 2))) return rc;
                t.bits[l] = reinterpret_cast<unsigned long long*>(bw);
            }
            if (post) {
                if ((rc = alloc_f(c, &t.zx[l], n))) return rc;
                if (m.ln && (rc = alloc_f(c, &t.dh[l], n))) return rc;
                if ((rc = alloc_f(c, &t.rstd[l], (size_t)G * cap))) return rc;
                float* mk = nullptr;
                if ((rc = alloc_f(c, &mk, (size_t)G * cap * ((m.dims[l + 1] + 63) / 64) * 2 + 2))) return rc;
                t.mask[l] = reinterpret_cast<unsigned long long*>(((uintptr_t)mk + 7) & ~(uintptr_t)7);
            }
        }
    }
    return MORL_OK;
}

static int fill_nets(const morl_ac_desc* d, Mlp& q, Mlp& pol, int& heads, bool& w_input) {
    if (!d) return fail(MORL_ERR_ARG, "desc is NULL");
    if (d->algo < MORL_AC_CAPQL || d->algo > MORL_AC_SACD) return fail(MORL_ERR_ARG, "unknown algo %d", d->algo);
    if (d->algo == MORL_AC_SACD && (d->num_q != 2 || d->act_dim > SACD_MAX_A || d->q_layer_norm || d->q_drop_rate > 0.f))
        return fail(MORL_ERR_ARG, "discrete MOSAC: two plain critics, at most %d actions", SACD_MAX_A);
    if (d->n_hidden < 1 || d->n_hidden > MORL_MAX_LAYERS - 1) return fail(MORL_ERR_ARG, "n_hidden %d out of range", d->n_hidden);
    if (d->obs_dim < 1 || d->act_dim < 1 || d->reward_dim < 1 || d->reward_dim > MORL_MAX_OBJ)
        return fail(MORL_ERR_ARG, "bad dims D=%d Ad=%d R=%d", d->obs_dim, d->act_dim, d->reward_dim);
    if (d->num_q < 1 || d->num_q > 4) return fail(MORL_ERR_ARG, "num_q %d not in 1..4", d->num_q);
    if (d->algo == MORL_AC_MOSAC && d->num_q != 2) return fail(MORL_ERR_ARG, "MOSAC uses exactly two critics");
    if (d->q_drop_rate < 0.f || d->q_drop_rate >= 1.f) return fail(MORL_ERR_ARG, "drop rate %g", (double)d->q_drop_rate);
    const bool sacd = d->algo == MORL_AC_SACD;
    w_input = d->algo != MORL_AC_MOSAC && !sacd;
    heads = (d->algo == MORL_AC_TD3 || sacd) ? 1 : 2;
    q = Mlp();
    pol = Mlp();
    q.L = pol.L = d->n_hidden + 1;
    q.dims[0] = sacd ? d->obs_dim : d->obs_dim + d->act_dim + (w_input ? d->reward_dim : 0);
    pol.dims[0] = d->obs_dim + (w_input ? d->reward_dim : 0);
    for (int l = 0; l < d->n_hidden; ++l) {
        if (d->hidden[l] < 1 || d->hidden[l] > 64 * POST_MAXJ) return fail(MORL_ERR_ARG, "hidden[%d] = %d", l, d->hidden[l]);
        q.dims[l + 1] = pol.dims[l + 1] = d->hidden[l];
    }
    q.dims[q.L] = sacd ? d->act_dim * d->reward_dim : d->reward_dim;
    pol.dims[pol.L] = heads * d->act_dim;
    q.ln = d->q_layer_norm != 0;
    q.drop = d->q_drop_rate;
    q.finish();
    pol.finish();
    return MORL_OK;
}

extern "C" int64_t morl_ac_q_param_count(const morl_ac_desc* d) {
    Mlp q, p; int h; bool w;
    return fill_nets(d, q, p, h, w) ? -1 : q.P;
}
extern "C" int64_t morl_ac_policy_param_count(const morl_ac_desc* d) {
    Mlp q, p; int h; bool w;
    return fill_nets(d, q, p, h, w) ? -1 : p.P;
}
extern "C" int64_t morl_ac_mask_bytes(const morl_ac_desc* d, int rows) {
    Mlp q, p; int h; bool w;
    if (fill_nets(d, q, p, h, w)) return -1;
    int64_t per_net = 0;
    for (int l = 0; l < q.L - 1; ++l) per_net += (int64_t)rows * q.dims[l + 1];
    return 3 * (int64_t)d->population * d->num_q * per_net;
}

extern "C" int morl_ac_destroy(morl_ac_ctx* c) {
    if (!c) return MORL_OK;
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    return MORL_OK;
}

extern "C" int morl_ac_create(morl_ac_ctx** out, const morl_ac_desc* d) {
    if (!out) return fail(MORL_ERR_ARG, "out is NULL");
    *out = nullptr;
    Mlp q, p; int heads; bool w_input;
    int rc = fill_nets(d, q, p, heads, w_input);
    if (rc) return rc;
    if (d->population < 1 || d->max_rows < 1) return fail(MORL_ERR_ARG, "population %d / max_rows %d", d->population, d->max_rows);
    morl_ac_ctx* c = new (std::nothrow) morl_ac_ctx();
    if (!c) return fail(MORL_ERR_ALLOC, "out of host memory");
    c->d = *d; c->q = q; c->pol = p; c->heads = heads; c->w_input = w_input;
    c->PG = d->population; c->QG = d->population * d->num_q; c->cap = d->max_rows;
    const bool post = q.ln || q.drop > 0.f;
    const size_t cap = (size_t)c->cap, Ad = (size_t)d->act_dim;
    if ((rc = alloc_tape(c->allocs, c->q, c->tq_a, c->QG, c->PG, c->cap, post)) ||
        (rc = alloc_tape(c->allocs, c->q, c->tq_b, c->QG, c->PG, c->cap, post)) ||
        (rc = alloc_tape(c->allocs, c->pol, c->tp_a, c->PG, c->PG, c->cap, false)) ||
        (rc = alloc_tape(c->allocs, c->pol, c->tp_b, c->PG, c->PG, c->cap, false)) ||
        (rc = alloc_f(c, &c->act, c->PG * cap * Ad)) || (rc = alloc_f(c, &c->act_pi, c->PG * cap * Ad)) ||
        (rc = alloc_f(c, &c->xq_pi, c->PG * cap * (size_t)q.ld[0])) || (rc = alloc_f(c, &c->logp_next, c->PG * cap)) ||
        (rc = alloc_f(c, &c->logp_pi, c->PG * cap)) || (rc = alloc_f(c, &c->save_y, c->PG * cap * Ad)) ||
        (rc = alloc_f(c, &c->save_std, c->PG * cap * Ad)) || (rc = alloc_f(c, &c->gq, (size_t)c->QG * q.P)) ||
        (rc = alloc_f(c, &c->gp, (size_t)c->PG * p.P)) || (rc = alloc_f(c, &c->alpha_dev, c->PG)) || (rc = alloc_f(c, &c->adam_corr, (size_t)4 * c->PG)) ||
        (rc = alloc_f(c, &c->wt_q, (size_t)c->QG * q.P)) || (rc = alloc_f(c, &c->wt_qt, (size_t)c->QG * q.P)) ||
        (rc = alloc_f(c, &c->wt_pol, (size_t)c->PG * p.P)) || (rc = alloc_f(c, &c->wt_polt, (size_t)c->PG * p.P))) {
        morl_ac_destroy(c);
        return rc;
    }
    if (hipDeviceSynchronize() != hipSuccess) { morl_ac_destroy(c); return fail(MORL_ERR_HIP, "workspace init failed"); }
    c->xq_b = c->tq_b.x;
    *out = c;
    return MORL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// batched layer launches
// ---------------------------------------------------------------------------------------------------------------------
// Engine choice per launch: the LDS-tiled 128 x 128 engine needs >= ~half a chip of tiles to be worth its barriers;
// below that the work is latency-bound and the wave-level 32 x 32 tiles (gemm_wave.h) spread it over many more waves
// (split-K over the four waves of a workgroup once K > 32: same sums in a different, still deterministic order).
static int g_ac_gemm_mode = 0;      // 0 auto, 1 always LDS tiles, 2 always wave tiles (morl_ac_set_gemm_mode)
static bool use_wave_tiles(long long tiles128) {
    if (g_ac_gemm_mode == 1) return false;
    if (g_ac_gemm_mode == 2) return true;
    constexpr long long below = 128;
    return tiles128 < below;
}

template <bool A_KC, bool B_KC, int EPI>
static int launch_bgemm(GemmBatched b, int G, hipStream_t s, const char* name) {
    GemmProblem& g = b.p;
    g.tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
    g.tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
    g.a_vec = vec_ok(g.A, g.lda) && (b.sA % 4 == 0);
    g.b_vec = vec_ok(g.B, g.ldb) && (b.sB % 4 == 0);
    g.k_per_split = round_up(g.K, GEMM_BK);
    if (use_wave_tiles((long long)g.tiles_m * g.tiles_n * G)) {
        g.tiles_m = (g.M + 31) / 32;
        g.tiles_n = (g.N + 31) / 32;
        if (g.K > 32)      // deep contraction: the four waves of a workgroup split K (one memory round trip per wave)
            hipLaunchKernelGGL((gemm_wave4_batched_kernel<A_KC, B_KC, EPI>), dim3(g.tiles_m * g.tiles_n, 1, G), dim3(256), 0, s, b);
        else
            hipLaunchKernelGGL((gemm_wave_batched_kernel<A_KC, B_KC, EPI>), dim3((g.tiles_m * g.tiles_n + 3) / 4, 1, G), dim3(256), 0, s, b);
    } else {
        hipLaunchKernelGGL((gemm_batched_kernel<A_KC, B_KC, EPI>), dim3(g.tiles_m * g.tiles_n, 1, G), dim3(GEMM_THREADS), 0, s, b);
    }
    LAUNCH_CHECK(name);
    return MORL_OK;
}

extern "C" int morl_ac_set_gemm_mode(int mode) {
    if (mode < 0 || mode > 2) return fail(MORL_ERR_ARG, "gemm mode %d not in 0..2", mode);
    g_ac_gemm_mode = mode;
    return MORL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// layer-fused passes: a plain ReLU MLP (no LayerNorm, no active Dropout, widths <= 256) runs ALL its layers in one launch of
// the persistent chain kernel (mlp_chain2.h) -- the G networks of a tape are one batched chain, a paired pass a second one
// in the same launch.  Round 1 ran one GEMM launch per layer: a single-learner update was 35-52 dependent ~8 us launches.
// ---------------------------------------------------------------------------------------------------------------------
// Measured on MI355X (round 2, one learner, 128-row batch; ms per update, per-layer engines -> fused passes): with the
// 16-row tiles of mlp_chain16.h CAPQL 0.197 -> 0.164, MOSAC 0.342 -> 0.276, GPI-PD continuous 0.261 -> 0.250.  (With 32-row
// tiles the fused pass LOSES: a pass is then 4-8 workgroups whose tiles carry a whole 256 x 256 layer each, 23 us per pass
// against 3 x 8 us for the wave-tile GEMMs that spread a layer over 32 workgroups -- CAPQL 0.214, MOSAC 0.367.)
// (the per-layer engines stay for networks the chain does not take: LayerNorm / Dropout, widths beyond 256)
static constexpr bool g_ac_chain = true;

static bool chain_shape_ok(const Mlp& m) {
    if (!g_ac_chain || m.ln || m.L < 2 || m.L > MORL_MAX_LAYERS) return false;
    if (m.dims[0] > CH_MAXW) return false;
    for (int l = 1; l < m.L; ++l)
        if (m.dims[l] > CH_MAXW || m.dims[l] <= 32 || (m.dims[l] & 3)) return false;     // hidden layers: wide steps, 16-byte rows
    if (m.dims[m.L] > CH_MAXW || (m.dims[m.L] > 32 && (m.dims[m.L] & 3))) return false;
    return true;
}

constexpr long long AC_C16_MAX_ROWS = 4096;     // row count up to which a chain of these three-layer nets takes the 16-row tiles
// ONE decision for "does a chain of this many rows (x networks) run on the 16-row tiles": ac_chain_launch (which kernel) and
// chain_nmajor (does the pass have K-major shadow weights at all) both ask here, so that the two
// cannot make them disagree (an N-major chain has no shadow copy for the 16-row kernel to stream)
static bool ac_rows_take_chain16(long long rows_x_nets) {
    static const bool small_rows = [] { const char* e = getenv("MORL_CHAIN16"); return e ? atoi(e) != 0 : true; }();
    constexpr long long limit = AC_C16_MAX_ROWS;
    return small_rows && rows_x_nets <= limit;
}
// The actor's backward chain with the policy head's backward pass in its input stage (ChainArgs::in_mode == 4): each tile computes
// its 16 rows of dLoss/d(head pre-activations) itself (ac_head_bwd_row: the function the separate ac_head_bwd_kernel runs) -- one
// launch and one first-touch round trip fewer per actor step.
struct HeadBwdHook {
    static constexpr bool active = true;
    HeadBwdArgs a;
    __device__ __forceinline__ void operator()(int g, int row, float (&v)[16]) const { ac_head_bwd_row(a, g, row, v); }
};
static __global__ __launch_bounds__(CH_THREADS, 2) void mlp_chain16_headbwd_kernel(Chain16Multi m, HeadBwdArgs hb) {
    __shared__ __attribute__((aligned(16))) float sAct[C16_TM * C2_LDK + 16];
    const ChainArgs& p = m.p[0];                                 // (one chain: the actor's backward pass)
    kernarg_warm<sizeof(ChainArgs)>();
    int lt = (int)blockIdx.x, g = 0;
    const int tpn = (p.rows + C16_TM - 1) / C16_TM;              // tiles per network
    if (p.nb > 1) { g = lt / tpn; lt -= g * tpn; }
    mlp_chain16_body<false, HeadBwdHook>(p, lt * C16_TM, sAct, g, HeadBwdHook{hb});
}

static int ac_chain_launch(const ChainArgs* chains, int n, hipStream_t s, const HeadBwdArgs* head_bwd = nullptr) {
    static const int num_cus = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return prop.multiProcessorCount;
        return 256;
    }();
    long long most = 0;
    for (int q = 0; q < n; ++q) {
        most = std::max(most, (long long)chains[q].rows * std::max(1, chains[q].nb));
        if ((chains[q].fast == 2) != (chains[0].fast == 2))
            return fail(MORL_ERR_STATE, "ac_chain_launch: chains of one launch disagree on the weight stream (fast %d vs %d)",
                        chains[0].fast, chains[q].fast);
    }
    if (chains[0].fast != 2 && ac_rows_take_chain16(most)) {
        for (int q = 0; q < n; ++q)
            for (int l = 0; l < chains[q].n_steps; ++l)
                if (chains[q].step[l].N > 32 && !chains[q].step[l].Bmat) return fail(MORL_ERR_STATE, "ac_chain_launch: 16-row chain without a K-major operand");
        Chain16Multi m16{};
        const int tiles = chain16_fill(m16, chains, n);
#ifdef C16_PROF
        // development build (-DC16_PROF): per-wave phase cycle sums of the launches of two updates, printed once
        static long long* prof_dev = nullptr;
        static int launch_no = 0;
        if (!prof_dev) hipMalloc(&prof_dev, (size_t)1024 * 4 * 8 * 8);
        for (int q = 0; q < n; ++q) m16.p[q].prof = prof_dev;
#endif
        if (head_bwd) hipLaunchKernelGGL(mlp_chain16_headbwd_kernel, dim3(tiles), dim3(CH_THREADS), 0, s, m16, *head_bwd);
        else hipLaunchKernelGGL(mlp_chain16_kernel, dim3(tiles), dim3(CH_THREADS), 0, s, m16);
        LAUNCH_CHECK("ac_chain16");
#ifdef C16_PROF
        if (++launch_no > 600 && launch_no <= 612 && tiles <= 1024) {
            hipStreamSynchronize(s);
            std::vector<long long> h((size_t)tiles * 4 * 8);
            hipMemcpy(h.data(), prof_dev, h.size() * 8, hipMemcpyDeviceToHost);
            double a[8] = {0};
            for (size_t w = 0; w < (size_t)tiles * 4; ++w)
                for (int i = 0; i < 8; ++i) a[i] += (double)h[w * 8 + i] / (tiles * 4);
            fprintf(stderr, "C16_PROF launch %d: %d tiles, steps %d; clock64 cycles per wave: input %.0f | first-step mfma %.0f | wide mfma %.0f | "
                            "barrier %.0f | epilogue+sync %.0f | narrow %.0f | final copy %.0f | total %.0f\n",
                    launch_no, tiles, chains[0].n_steps, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
        }
#endif
        return MORL_OK;
    }
    if (head_bwd) return fail(MORL_ERR_STATE, "ac_chain_launch: the head's backward pass rides on the 16-row tiles only");
    Chain2Multi m{};
    m.n = n;
    int units = 0;
    for (int q = 0; q < n; ++q) {
        m.p[q] = chains[q];
        m.unit_start[q] = units;
        units += std::max(1, chains[q].nb) * ((chains[q].rows + 63) / 64);
    }
    for (int q = n; q <= CH_MAX_MULTI; ++q) m.unit_start[q] = units;
    const int S = std::max(1, std::min(2 * num_cus, 2 * units));
    m.full_rounds = units / S;
    m.tail_base = m.full_rounds * S;
    m.tail_units = units - m.tail_base;
    m.tail_halves = (2 * m.tail_units <= S) ? 1 : 0;
    m.stagger = 1;
    constexpr bool xcd_env = true;
    for (int q = 0; q < n; ++q) m.xcd_contig |= (xcd_env && chains[q].nb > 1) ? 1 : 0;
    if (chains[0].fast == 2) hipLaunchKernelGGL(mlp_chain2_n_kernel, dim3(S), dim3(CH_THREADS), 0, s, m);
    else hipLaunchKernelGGL(mlp_chain2_kernel<1>, dim3(S), dim3(CH_THREADS), 0, s, m);
    LAUNCH_CHECK("ac_chain");
    return MORL_OK;
}

// Populations (every pass of the update in the 32 / 64-row regime of the chain): the forward chain streams the nn.Linear
// matrices as they are (N-major weight stream, mlp_chain2.h) and the first-layer dX step of the backward chain reads them
// K-major -- no K-major shadow copies, no transposes at the start of an update and behind every Adam step (98 of the 840 us of a
// 64-learner MORL/D update), and the stream itself is faster than the generic K-major one (half the load instructions).  Single
// learners keep the shadow copies: the same stream on the 16-row chain issues as many loads as the K-major one, uncoalesced, and
// measured slower than the transposes it saves (CAPQL 0.168 vs 0.165 ms, MOSAC 0.290 vs 0.275 ms).
static const bool g_ac_nmajor = [] { const char* e = getenv("MORL_AC_NMAJOR"); return e ? atoi(e) != 0 : true; }();   // (A/B runs)
static bool chain_nmajor(const Mlp& m, int rows, int G) {
    return g_ac_nmajor && chain_shape_ok(m) && !ac_rows_take_chain16((long long)rows * G);
}

// forward chain of the G networks of `t`: input rows t.x (shared by x_div networks), every hidden activation saved to t.h[]
// (the weight-gradient GEMM reads them) together with its sign bits, output to t.out
static ChainArgs ac_forward_chain(const Mlp& m, const float* params, const float* wt, int64_t pstride, Tape& t, int rows, int x_div) {
    ChainArgs a{};
    a.n_steps = m.L;
    a.rows = rows;
    a.in_mode = 1;
    a.src = t.x; a.ldsrc = m.ld[0]; a.K0 = m.dims[0];
    a.nb = t.G; a.sSrc = (long long)t.cap * m.ld[0]; a.src_div = x_div;
    a.fast = wt ? 0 : 2;                       // no shadow copy: N-major weight stream (populations, see chain_nmajor)
    const long long units = (t.cap + 63) / 64;
    for (int l = 0; l < m.L; ++l) {
        ChainStep& st = a.step[l];
        const bool last = (l == m.L - 1);
        st.Bmat = wt ? wt + m.offW[l] : nullptr;   // K-major shadow copy [in][out]
        st.ldb = m.dims[l + 1];
        st.Bt = params + m.offW[l];            // nn.Linear layout [out][in] = N-major
        st.ldbt = m.dims[l];
        st.K = m.dims[l];
        st.N = m.dims[l + 1];
        st.bias = params + m.offB[l];
        st.relu = last ? 0 : 1;
        st.sW = pstride;
        st.out = last ? t.out : t.h[l];
        st.ldout = m.ld[l + 1];
        st.sOut = (long long)t.cap * m.ld[l + 1];
        if (!last) { st.bits_out = t.bits[l]; st.sBits = units * CH_THREADS; }
    }
    return a;
}

// ---- LayerNorm / Dropout networks on the 16-row chain (mlp_chain16.h: mlp_chain16_post_kernel) ------------------------------------
// The hidden layers' post-ops (Dropout -> LayerNorm -> ReLU, common/networks.py:10-48) run on the tile's rows in LDS, so a pass is ONE
// launch instead of a GEMM + a post-op (+ a LayerNorm-gradient) launch per layer: GPI-PD's critics (gpi_pd.py:41-76,
// gpi_pd_continuous_action.py:60-73).  MORL_AC_LN_CHAIN: bit 0 forward passes, bit 1 backward passes; 0 = the per-layer launches.
// Default 1 -- measured on MI355X (profiles/r05_ln_chain_ab.json): the FORWARD chain wins (GPI-PD discrete 0.2845 -> 0.2639 ms per
// update, continuous 0.2243 -> 0.2194), the backward chain does not (its four 256-wide dX steps and three post-op stages are one
// serial 73 us chain on 16 - 32 workgroups where the per-layer launches spread each step over the chip: 0.2845 -> 0.3010): built,
// held to the same fixtures (tests/test_ln_chain.py runs both), not the default.
static const int g_ac_ln_chain = [] { const char* e = getenv("MORL_AC_LN_CHAIN"); return e ? atoi(e) : 1; }();
static bool chain_post_ok(const Mlp& m, long long rows_x_nets) {
    if (!g_ac_chain || !(m.ln || m.drop > 0.f) || m.L < 2 || m.L > MORL_MAX_LAYERS) return false;
    if (m.dims[0] > CH_MAXW) return false;
    for (int l = 1; l < m.L; ++l)
        if (m.dims[l] > CH_MAXW || m.dims[l] <= 32 || (m.dims[l] & 3)) return false;
    if (m.dims[m.L] > CH_MAXW || (m.dims[m.L] > 32 && (m.dims[m.L] & 3))) return false;
    return ac_rows_take_chain16(rows_x_nets);          // (the post-op stages exist on the 16-row tiles only)
}

struct DropSpec {
    bool active = false;              // train-mode dropout
    const uint8_t* ext = nullptr;     // explicit masks of this phase: [G][per_net bytes]
    int64_t ext_net_bytes = 0;
    unsigned long long seed = 0;
};

// forward of G nets (params + g * pstride) on t.x (shared by groups of x_div nets); result in t.out
// params2 / t2 / ds2: a second, independent pass of the same network shape (other parameters, other rows) whose layer GEMMs
// ride in the same launches (GemmBatched::split); its Dropout / LayerNorm post-ops, if any, are launched per tape
static int mlp_forward(const Mlp& m, const float* params, int64_t pstride, Tape& t, int rows, int x_div,
                       const DropSpec& ds, hipStream_t s, const float* params2 = nullptr, Tape* t2 = nullptr,
                       const DropSpec* ds2 = nullptr, const float* wt = nullptr, const float* wt2 = nullptr) {
    // wt / wt2: K-major shadow copies of params / params2 (same flat layout): the weight operand is then read along the
    // output index like the dX GEMMs do, without the LDS panel transposes of the contraction-contiguous form
    const long long cap = t.cap;
    if ((wt != nullptr) != (t2 ? wt2 != nullptr : wt != nullptr)) return fail(MORL_ERR_STATE, "paired passes: both or no shadow copy");
    if (t2 && (t2->cap != t.cap || t2->G != t.G)) return fail(MORL_ERR_STATE, "paired passes need equally shaped tapes");
    t.bits_valid = false;
    if (t2) t2->bits_valid = false;
    const bool plain = !(ds.active && m.drop > 0.f) && !(t2 && ds2 && ds2->active && m.drop > 0.f);
    if ((wt != nullptr || chain_nmajor(m, rows, t.G)) && plain && chain_shape_ok(m)) {
        ChainArgs ch[2] = {ac_forward_chain(m, params, wt, pstride, t, rows, x_div), ChainArgs{}};
        if (t2) ch[1] = ac_forward_chain(m, params2, wt2, pstride, *t2, rows, x_div);
        int rc = ac_chain_launch(ch, t2 ? 2 : 1, s);
        if (rc) return rc;
        t.bits_valid = true;
        if (t2) t2->bits_valid = true;
        return MORL_OK;
    }
    // LayerNorm / Dropout networks with a K-major shadow copy, few enough rows for the 16-row tiles: the whole pass (both passes of
    // a pair) as ONE chain launch whose hidden steps carry their post-op (mlp_chain16.h: c16_post_fwd)
    if ((g_ac_ln_chain & 1) && wt != nullptr && chain_post_ok(m, (long long)rows * t.G)) {
        Chain16PostMulti pm{};
        pm.n = t2 ? 2 : 1;
        int tiles = 0;
        for (int pass = 0; pass < pm.n; ++pass) {
            Tape& tt = pass ? *t2 : t;
            const DropSpec& dd = pass ? *ds2 : ds;
            const float* pp = pass ? params2 : params;
            ChainArgs ch = ac_forward_chain(m, pp, pass ? wt2 : wt, pstride, tt, rows, x_div);
            ChainPostSet& ps = pm.ps[pass];
            const bool drop = dd.active && m.drop > 0.f;
            ps.pstride = pstride; ps.ext_gstride = dd.ext_net_bytes; ps.cap = tt.cap;
            ps.drop_p = m.drop; ps.inv_keep = 1.0f / (1.0f - m.drop);
            int64_t ext_off = 0;
            for (int l = 0; l < m.L - 1; ++l) {
                ChainStep& st = ch.step[l];
                st.relu = 0;                       // (the post-op applies it, after Dropout / LayerNorm)
                st.bits_out = nullptr;             // (the backward of these nets reads h, not sign bits)
                ChainPost& a = ps.st[l];
                a.active = (m.ln || drop) ? 1 : 0;
                if (!a.active) { st.relu = 1; continue; }
                a.xhat = tt.zx[l]; a.rstd = tt.rstd[l]; a.mask = drop ? tt.mask[l] : nullptr;
                a.ext_mask = (drop && dd.ext) ? dd.ext + ext_off : nullptr;
                a.gamma = m.ln ? pp + m.offG[l] : nullptr;
                a.seed = dd.seed * 0x100000001B3ull + (unsigned long long)(l + 1) * 0x9E3779B97F4A7C15ull;
                a.gstride = cap * m.ld[l + 1]; a.ld = m.ld[l + 1];
                a.drop = drop ? 1 : 0;
                ext_off += (int64_t)rows * m.dims[l + 1];
            }
            pm.p[pass] = ch;
            pm.tile_start[pass] = tiles;
            tiles += std::max(1, ch.nb) * ((ch.rows + C16_TM - 1) / C16_TM);
        }
        for (int q = pm.n; q <= 2; ++q) pm.tile_start[q] = tiles;
        hipLaunchKernelGGL(mlp_chain16_post_kernel<1>, dim3(tiles), dim3(CH_THREADS), 0, s, pm);
        LAUNCH_CHECK("ac_chain16_post_fwd");
        return MORL_OK;
    }
    int64_t ext_off = 0;
    for (int l = 0; l < m.L; ++l) {
        const bool last = (l == m.L - 1);
        const bool drop = !last && ds.active && m.drop > 0.f;
        const bool post = !last && (m.ln || drop);
        GemmBatched b{};
        GemmProblem& g = b.p;
        g.A = (l == 0) ? t.x : t.h[l - 1];
        g.lda = m.ld[l];
        b.sA = cap * m.ld[l];
        b.a_div = (l == 0) ? x_div : 1;
        g.B = (wt ? wt : params) + m.offW[l];
        g.ldb = wt ? m.dims[l + 1] : m.dims[l];
        b.sB = pstride;
        g.bias = params + m.offB[l];
        b.sBias = pstride;
        g.C = last ? t.out : (post ? t.zx[l] : t.h[l]);
        g.ldc = m.ld[l + 1];
        b.sC = cap * m.ld[l + 1];
        g.M = rows; g.N = m.dims[l + 1]; g.K = m.dims[l];
        const bool drop2 = t2 && !last && ds2 && ds2->active && m.drop > 0.f;
        const bool post2 = t2 && !last && (m.ln || drop2);
        int G = t.G;
        if (t2) {
            // the epilogue (bias vs bias + ReLU) is chosen per launch: both passes must agree on whether a post-op follows
            if (post != post2) return fail(MORL_ERR_STATE, "paired passes disagree on the post-op of layer %d", l);
            b.split = t.G;
            b.A2 = (l == 0) ? t2->x : t2->h[l - 1];
            b.B2 = (wt2 ? wt2 : params2) + m.offW[l];
            b.bias2 = params2 + m.offB[l];
            b.C2 = last ? t2->out : (post ? t2->zx[l] : t2->h[l]);
            G = 2 * t.G;
        }
        int rc;
        if (wt) rc = (last || post) ? launch_bgemm<true, false, EPI_BIAS>(b, G, s, "ac_gemm_fwd_t")
                                    : launch_bgemm<true, false, EPI_BIAS_RELU>(b, G, s, "ac_gemm_fwd_relu_t");
        else rc = (last || post) ? launch_bgemm<true, true, EPI_BIAS>(b, G, s, "ac_gemm_fwd")
                                 : launch_bgemm<true, true, EPI_BIAS_RELU>(b, G, s, "ac_gemm_fwd_relu");
        if (rc) return rc;
        PostArgs both[2];
        for (int pass = 0; pass < (t2 ? 2 : 1) && post; ++pass) {
            Tape& tt = pass ? *t2 : t;
            const DropSpec& dd = pass ? *ds2 : ds;
            const float* pp = pass ? params2 : params;
            const bool dr = pass ? drop2 : drop;
            PostArgs a{};
            a.z = tt.zx[l]; a.h = tt.h[l]; a.rstd = tt.rstd[l]; a.mask = tt.mask[l];
            a.ext_mask = (dr && dd.ext) ? dd.ext + ext_off : nullptr;
            a.ext_gstride = dd.ext_net_bytes;
            a.gamma = m.ln ? pp + m.offG[l] : nullptr;
            a.pstride = pstride;
            a.gstride = cap * m.ld[l + 1];
            a.cap = tt.cap; a.N = m.dims[l + 1]; a.ld = m.ld[l + 1]; a.rows = rows;
            a.ln = m.ln ? 1 : 0; a.drop = dr ? 1 : 0;
            a.drop_p = m.drop; a.inv_keep = 1.0f / (1.0f - m.drop);
            a.seed = dd.seed * 0x100000001B3ull + (unsigned long long)(l + 1) * 0x9E3779B97F4A7C15ull;
            both[pass] = a;
        }
        if (post && t2) {
            hipLaunchKernelGGL(ac_post_fwd_pair_kernel, dim3((rows + 3) / 4, t.G, 2), dim3(256), 0, s, both[0], both[1]);
            LAUNCH_CHECK("ac_post_fwd_pair");
        } else if (post) {
            hipLaunchKernelGGL(ac_post_fwd_kernel, dim3((rows + 3) / 4, t.G), dim3(256), 0, s, both[0]);
            LAUNCH_CHECK("ac_post_fwd");
        }
        if (!last) ext_off += (int64_t)rows * m.dims[l + 1];
    }
    return MORL_OK;
}

// backward of the same pass: t.g[L-1] holds dLoss/d(out).  grads ([G][P], fully overwritten) may be NULL (no parameter
// gradients wanted); need_dx -> t.dx = dLoss/d(input rows).  `dropped` = the forward ran with train-mode dropout.
// fuse / fused: the caller offers the optimiser step of these gradients (AdamFuse with everything but the layer offsets filled
// in); *fused says whether the weight-gradient launch took it -- it does when it runs on the split-K wave tiles (single learners
// and small populations), where a workgroup's tile holds finished gradient entries; otherwise `grads` is written as usual
static int mlp_backward(const Mlp& m, const float* params, int64_t pstride, Tape& t, int rows, int x_div,
                        bool dropped, float* grads, bool need_dx, hipStream_t s, int64_t grad_stride = -1,
                        const float* wt = nullptr, const AdamFuse* fuse = nullptr, bool* fused = nullptr,
                        const HeadBwdArgs* head_bwd = nullptr) {
    if (fused) *fused = false;
    // head_bwd: t.g[L-1] is still to be computed from the critics' dLoss/d(action) (the actor's head).  In the chain's input stage
    // when the pass runs as a 16-row chain and a row's entries fit one column block; by its own launch otherwise
    static const bool head_in_chain = [] { const char* e = getenv("MORL_AC_HEADBWD_IN_CHAIN"); return e ? atoi(e) != 0 : true; }();   // (A/B)
    bool head_pending = head_bwd != nullptr;
    const bool head_rides = head_pending && head_in_chain && t.bits_valid && chain_shape_ok(m) && !(dropped && m.drop > 0.f) &&
                            m.dims[m.L] <= 16 && m.L - 1 >= 1 && ac_rows_take_chain16((long long)rows * t.G);
    if (head_pending && !head_rides) {
        hipLaunchKernelGGL(ac_head_bwd_kernel, dim3((head_bwd->G * rows + 255) / 256), dim3(256), 0, s, *head_bwd);
        LAUNCH_CHECK("ac_head_bwd");
        head_pending = false;
    }
    const long long cap = t.cap;
    if (grad_stride < 0) grad_stride = m.P;      // floats between the gradient blocks of consecutive nets
    // dX chain in one launch when the forward was the layer-fused one (its ReLU sign bits are in the tape): g[L-1] -> ... ->
    // g[0] (-> dx), every g[l] written for the weight-gradient GEMM below.  A narrow input layer (<= 32 columns) needs the
    // K-major shadow copy as its N-major operand; without one that last dX step stays a GEMM launch.
    int chain_down_to = -1;                      // layers [chain_down_to, L-1] had their dX computed by the chain
    if (t.bits_valid && chain_shape_ok(m) && !(dropped && m.drop > 0.f)) {
        // (the 32 / 64-row chain reads a narrow step's operand K-major when there is no shadow copy)
        const int last_l = need_dx ? ((m.dims[0] > 32 || wt != nullptr || chain_nmajor(m, rows, t.G)) ? 0 : 1) : 1;
        if (m.L - 1 >= last_l) {
            ChainArgs a{};
            a.rows = rows;
            a.in_mode = head_pending ? 4 : 1;
            a.src = t.g[m.L - 1]; a.ldsrc = m.ld[m.L]; a.K0 = m.dims[m.L];
            a.nb = t.G; a.sSrc = cap * m.ld[m.L]; a.src_div = 1;
            a.fast = 0;
            const long long units = (t.cap + 63) / 64;
            int k = 0;
            for (int l = m.L - 1; l >= last_l; --l, ++k) {
                ChainStep& st = a.step[k];
                st.Bmat = params + m.offW[l];          // [out][in]: K-major for g_l @ W_l
                st.ldb = m.dims[l];
                st.Bt = wt ? wt + m.offW[l] : nullptr; // [in][out]: N-major (narrow steps only)
                st.ldbt = m.dims[l + 1];
                st.K = m.dims[l + 1];
                st.N = m.dims[l];
                st.sW = pstride;
                if (l > 0) { st.bits_in = t.bits[l - 1]; st.sBits = units * CH_THREADS; }
                st.out = (l == 0) ? t.dx : t.g[l - 1];
                st.ldout = m.ld[l];
                st.sOut = cap * m.ld[l];
            }
            a.n_steps = k;
            int rc = ac_chain_launch(&a, 1, s, head_pending ? head_bwd : nullptr);
            if (rc) return rc;
            head_pending = false;
            chain_down_to = last_l;
        }
    }
    if (head_pending) return fail(MORL_ERR_STATE, "mlp_backward: the head's backward pass was left to a chain that did not run");
    // LayerNorm / Dropout networks: the dX steps with the hidden layers' post-op derivatives between them as ONE chain launch
    // (mlp_chain16.h: c16_post_bwd), then the LayerNorm affine gradients of all layers in one launch (from the dLoss/dh copies
    // the chain left in t.dh[])
    if ((g_ac_ln_chain & 2) && chain_post_ok(m, (long long)rows * t.G) && (!m.ln || !grads || t.dh[0] != nullptr) && m.L - 1 >= 1) {
        const bool drop = dropped && m.drop > 0.f;
        const int last_l = need_dx ? 0 : 1;
        Chain16PostMulti pm{};
        pm.n = 1;
        ChainArgs& a = pm.p[0];
        a.rows = rows;
        a.in_mode = 1;
        a.src = t.g[m.L - 1]; a.ldsrc = m.ld[m.L]; a.K0 = m.dims[m.L];
        a.nb = t.G; a.sSrc = cap * m.ld[m.L]; a.src_div = 1;
        a.fast = 0;
        ChainPostSet& ps = pm.ps[0];
        ps.pstride = pstride; ps.cap = t.cap; ps.drop_p = m.drop; ps.inv_keep = 1.0f / (1.0f - m.drop);
        int k = 0;
        for (int l = m.L - 1; l >= last_l; --l, ++k) {
            ChainStep& st = a.step[k];
            st.Bmat = params + m.offW[l];          // [out][in]: K-major for g_l @ W_l
            st.ldb = m.dims[l];
            st.Bt = wt ? wt + m.offW[l] : nullptr;
            st.ldbt = m.dims[l + 1];
            st.K = m.dims[l + 1];
            st.N = m.dims[l];
            st.sW = pstride;
            st.out = (l == 0) ? t.dx : t.g[l - 1];
            st.ldout = m.ld[l];
            st.sOut = cap * m.ld[l];
            ChainPost& po = ps.st[k];
            po.active = (l > 0 && (m.ln || drop)) ? 1 : 0;
            if (l > 0 && !po.active) return fail(MORL_ERR_STATE, "mlp_backward: post-op chain on a layer without a post-op");
            if (po.active) {
                const int hl = l - 1;              // hidden layer whose post-op is differentiated
                po.xhat = t.zx[hl]; po.rstd = t.rstd[hl]; po.mask = drop ? t.mask[hl] : nullptr; po.h = t.h[hl];
                po.gamma = m.ln ? params + m.offG[hl] : nullptr;
                po.dh_out = (m.ln && grads) ? t.dh[hl] : nullptr;
                po.gstride = cap * m.ld[l]; po.ld = m.ld[l];
                po.drop = drop ? 1 : 0;
            }
        }
        a.n_steps = k;
        pm.tile_start[0] = 0;
        const int tiles = std::max(1, a.nb) * ((rows + C16_TM - 1) / C16_TM);
        pm.tile_start[1] = pm.tile_start[2] = tiles;
        if (m.dims[0] <= 32 && need_dx && !wt) return fail(MORL_ERR_STATE, "mlp_backward: a narrow first layer needs the K-major shadow copy");
        hipLaunchKernelGGL(mlp_chain16_post_kernel<2>, dim3(tiles), dim3(CH_THREADS), 0, s, pm);
        LAUNCH_CHECK("ac_chain16_post_bwd");
        if (m.ln && grads) {
            LnGradMulti lg{};
            lg.n = m.L - 1;
            int max_n = 0;
            for (int hl = 0; hl < m.L - 1; ++hl) {
                LnGradArgs& q = lg.a[hl];
                q.d = t.dh[hl]; q.h = t.h[hl]; q.xhat = t.zx[hl];
                q.dgamma = grads + m.offG[hl];
                q.pstride = grad_stride; q.gstride = cap * m.ld[hl + 1];
                q.N = m.dims[hl + 1]; q.ld = m.ld[hl + 1]; q.rows = rows;
                max_n = std::max(max_n, q.N);
            }
            hipLaunchKernelGGL(ac_ln_grad_multi_kernel, dim3((max_n + 63) / 64, t.G, lg.n), dim3(64 * COLRED_WAVES), 0, s, lg);
            LAUNCH_CHECK("ac_ln_grad_multi");
        }
        chain_down_to = last_l;
    }
    for (int l = m.L - 1; l >= 0; --l) {
        if (l == 0 && !need_dx) break;
        if (chain_down_to >= 0 && l >= chain_down_to) continue;
        const bool drop = l > 0 && dropped && m.drop > 0.f;
        const bool post = l > 0 && (m.ln || drop);
        GemmBatched b{};
        GemmProblem& g = b.p;
        g.A = t.g[l];
        g.lda = m.ld[l + 1];
        b.sA = cap * m.ld[l + 1];
        b.a_div = 1;
        g.B = params + m.offW[l];
        g.ldb = m.dims[l];
        b.sB = pstride;
        g.C = (l == 0) ? t.dx : t.g[l - 1];
        g.ldc = m.ld[l];
        b.sC = cap * m.ld[l];
        g.M = rows; g.N = m.dims[l]; g.K = m.dims[l + 1];
        int rc;
        if (l > 0 && !post) {
            g.mask = t.h[l - 1];
            g.ldmask = m.ld[l];
            b.sMask = cap * m.ld[l];
            rc = launch_bgemm<true, false, EPI_RELU_MASK>(b, t.G, s, "ac_gemm_dx_relu");
        } else {
            rc = launch_bgemm<true, false, EPI_STORE>(b, t.G, s, "ac_gemm_dx");
        }
        if (rc) return rc;
        if (post) {
            const int hl = l - 1;                           // hidden layer whose post-op is differentiated
            if (m.ln && grads) {
                LnGradArgs a{};
                a.d = t.g[hl]; a.h = t.h[hl]; a.xhat = t.zx[hl];
                a.dgamma = grads + m.offG[hl];
                a.pstride = grad_stride; a.gstride = cap * m.ld[l];
                a.N = m.dims[l]; a.ld = m.ld[l]; a.rows = rows;
                hipLaunchKernelGGL(ac_ln_grad_kernel, dim3((a.N + 63) / 64, t.G), dim3(64 * COLRED_WAVES), 0, s, a);
                LAUNCH_CHECK("ac_ln_grad");
            }
            PostBwdArgs a{};
            a.d = t.g[hl]; a.h = t.h[hl]; a.xhat = t.zx[hl]; a.rstd = t.rstd[hl]; a.mask = t.mask[hl];
            a.gamma = m.ln ? params + m.offG[hl] : nullptr;
            a.pstride = pstride; a.gstride = cap * m.ld[l];
            a.cap = t.cap; a.N = m.dims[l]; a.ld = m.ld[l]; a.rows = rows;
            a.ln = m.ln ? 1 : 0; a.drop = drop ? 1 : 0;
            a.inv_keep = 1.0f / (1.0f - m.drop);
            hipLaunchKernelGGL(ac_post_bwd_kernel, dim3((rows + 3) / 4, t.G), dim3(256), 0, s, a);
            LAUNCH_CHECK("ac_post_bwd");
        }
    }
    if (grads) {
        GemmGroupBatched grp{};
        grp.n = m.L;
        grp.sC = grad_stride;
        int tiles = 0;
        for (int l = 0; l < m.L; ++l) {
            GemmProblem& g = grp.p[l];
            g.A = t.g[l];
            g.lda = m.ld[l + 1];
            grp.sA[l] = cap * m.ld[l + 1];
            g.B = (l == 0) ? t.x : t.h[l - 1];
            g.ldb = m.ld[l];
            grp.sB[l] = cap * m.ld[l];
            grp.b_div[l] = (l == 0) ? x_div : 1;
            g.C = grads + m.offW[l];
            g.ldc = m.dims[l];
            g.colsum = grads + m.offB[l];
            g.M = m.dims[l + 1]; g.N = m.dims[l]; g.K = rows;
            g.k_per_split = round_up(rows, GEMM_BK);
            g.tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
            g.tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
            g.a_vec = vec_ok(g.A, g.lda) && (grp.sA[l] % 4 == 0);
            g.b_vec = vec_ok(g.B, g.ldb) && (grp.sB[l] % 4 == 0);
            grp.tile_start[l] = tiles;
            tiles += g.tiles_m * g.tiles_n;
        }
        grp.tile_start[m.L] = tiles;
        if (use_wave_tiles((long long)tiles * t.G)) {
            tiles = 0;
            for (int l = 0; l < m.L; ++l) {
                GemmProblem& g = grp.p[l];
                g.tiles_m = (g.M + 31) / 32;
                g.tiles_n = (g.N + 31) / 32;
                grp.tile_start[l] = tiles;
                tiles += g.tiles_m * g.tiles_n;
            }
            grp.tile_start[m.L] = tiles;
            if (rows > 32 && fuse && fused && grad_stride == m.P) {
                AdamFuse f = *fuse;
                for (int l = 0; l < m.L; ++l) { f.offW[l] = m.offW[l]; f.offB[l] = m.offB[l]; }
                f.grads = grads;
                if (m.ln)
                    for (int l = 0; l < m.L - 1; ++l) { f.extra_off[f.n_extra] = m.offG[l]; f.extra_len[f.n_extra++] = 2 * m.dims[l + 1]; }
                hipLaunchKernelGGL(gemm_wave4_grouped_tn_batched_adam_kernel, dim3(tiles + (f.n_extra ? 1 : 0), 1, t.G), dim3(256), 0, s,
                                   grp, f);
                *fused = true;
            } else if (rows > 32)
                hipLaunchKernelGGL(gemm_wave4_grouped_tn_batched_kernel, dim3(tiles, 1, t.G), dim3(256), 0, s, grp);
            else
                hipLaunchKernelGGL(gemm_wave_grouped_tn_batched_kernel, dim3((tiles + 3) / 4, 1, t.G), dim3(256), 0, s, grp);
        } else {
            hipLaunchKernelGGL(gemm_grouped_tn_batched_kernel, dim3(tiles, 1, t.G), dim3(GEMM_THREADS), 0, s, grp);
        }
        LAUNCH_CHECK("ac_gemm_dw");
    }
    return MORL_OK;
}

static int concat(morl_ac_ctx* c, float* dst, int ld, int G, int rows, const float* s0, int w0, const float* s1, int w1,
                  const float* s2, int w2, hipStream_t s) {
    ConcatArgs a{};
    const float* src[3] = {s0, s1, s2};
    const int wd[3] = {w0, w1, w2};
    int n = 0;
    for (int k = 0; k < 3; ++k)
        if (src[k] && wd[k] > 0) {
            a.src[n] = src[k]; a.width[n] = wd[k];
            a.gstride[n] = (long long)rows * wd[k]; a.rstride[n] = wd[k];
            ++n;
        }
    a.n_src = n;
    a.dst = dst; a.ld = ld; a.dst_gstride = (long long)c->cap * ld; a.rows = rows; a.G = G;
    hipLaunchKernelGGL(ac_concat_kernel, dim3(stream_grid((long long)G * rows * ld, 256)), dim3(256), 0, s, a);
    LAUNCH_CHECK("ac_concat");
    return MORL_OK;
}

// one Adam step on `G` learner segments of `seg` floats each; t = steps[g] + step_add, or step_add when steps == NULL
static MlpLayout layout_of(const Mlp& m) {
    MlpLayout t{};
    t.L = m.L;
    t.P = m.P;
    for (int l = 0; l < m.L; ++l) { t.offW[l] = m.offW[l]; t.K[l] = m.dims[l]; t.N[l] = m.dims[l + 1]; }
    return t;
}

static int host_tiles(const MlpLayout& t) {
    int n = 0;
    for (int l = 0; l < t.L; ++l) n += ((t.N[l] + TR_T - 1) / TR_T) * ((t.K[l] + TR_T - 1) / TR_T);
    return n;
}

// K-major shadow copies of up to four parameter sets in one launch (ac_kernels.h: ac_transpose_multi_kernel)
static int launch_transposes(const TransposeMulti& tm, hipStream_t s) {
    int tiles = 0, nets = 0;
    for (int k = 0; k < tm.n; ++k) { tiles = std::max(tiles, host_tiles(tm.lay[k])); nets = std::max(nets, tm.nets[k]); }
    if (tm.n < 1 || nets < 1) return MORL_OK;
    long long longest = 0;
    for (int k = 0; k < tm.n; ++k) longest = std::max(longest, tm.lay[k].P * tm.nets[k]);
    static const long long scatter_max = [] { const char* e = getenv("MORL_AC_SCATTER_MAX"); return e ? atoll(e) : (1ll << 20); }();   // (tests)
    if (longest <= scatter_max) {
        hipLaunchKernelGGL(ac_transpose_scatter_kernel, dim3(stream_grid(longest, 256, 1024), 1, tm.n), dim3(256), 0, s, tm);
        LAUNCH_CHECK("ac_transpose");
        return MORL_OK;
    }
    if (nets > 65535) return fail(MORL_ERR_ARG, "%d nets in one transpose launch", nets);
    hipLaunchKernelGGL(ac_transpose_multi_kernel, dim3(tiles + 1, nets, tm.n), dim3(256), 0, s, tm);
    LAUNCH_CHECK("ac_transpose");
    return MORL_OK;
}

// wt / net: also refresh the K-major shadow copy of the stepped parameters (seg = a whole number of `net`-shaped nets)
static int adam(float* params, float* grads, float* m, float* v, long long seg, int G, double lr, const int* steps,
                int step_add, const morl_ac_cfg* cfg, hipStream_t s, float* wt = nullptr, const Mlp* net = nullptr,
                const MlpLayout* lay = nullptr) {
    const int nblk = std::min(256, stream_grid(seg, 256));
    // a few nets: the Adam kernel scatters the stepped values into the shadow copy itself (one launch fewer); a population:
    // the scattered 4-byte stores cost more than the step (93 against 37 us at 64 learners), so the copy is re-made by the
    // tiled transpose kernel behind it
    const MlpLayout layout = wt ? (lay ? *lay : layout_of(*net)) : MlpLayout{};
    static const long long scatter_max = [] { const char* e = getenv("MORL_AC_SCATTER_MAX"); return e ? atoll(e) : (1ll << 20); }();   // (tests)
    const bool scatter = wt && seg * G <= scatter_max;
    hipLaunchKernelGGL(ac_adam_kernel, dim3(nblk, G), dim3(256), 0, s, params, (const float*)grads, m, v, seg, steps, step_add,
                       lr, cfg->beta1, cfg->beta2, (float)cfg->eps, scatter ? wt : nullptr, layout);
    LAUNCH_CHECK("ac_adam");
    if (wt && !scatter) {
        TransposeMulti tm{};
        tm.n = 1;
        tm.src[0] = params; tm.dst[0] = wt; tm.lay[0] = layout; tm.nets[0] = (int)(seg / layout.P) * G;
        return launch_transposes(tm, s);
    }
    return MORL_OK;
}

static int polyak(const float* src, float* dst, long long n, float tau, hipStream_t s) {
    hipLaunchKernelGGL(polyak_kernel, dim3(stream_grid(n, OPT_THREADS)), dim3(OPT_THREADS), 0, s, src, dst, n, tau, 1.0f - tau);
    LAUNCH_CHECK("ac_polyak");
    return MORL_OK;
}

static HeadArgs head_args(morl_ac_ctx* c, Tape& tp, int rows, const float* eps, const morl_ac_state* st,
                          const morl_ac_cfg* cfg, float* action, float* logp, bool save, Tape* qin = nullptr) {
    HeadArgs a{};
    a.head = tp.out;
    a.head_gstride = (long long)c->cap * c->pol.ld[c->pol.L];
    a.ldh = c->pol.ld[c->pol.L];
    a.eps = eps;
    a.scale = st->action_scale; a.bias = st->action_bias;
    a.action = action; a.logp = logp;
    a.save_y = save ? c->save_y : nullptr;
    a.save_std = save ? c->save_std : nullptr;
    if (qin) {                // also drop the action straight into the critics' input rows
        a.xdst = qin->x; a.x_gstride = (long long)qin->cap * c->q.ld[0]; a.xld = c->q.ld[0]; a.xcol0 = c->d.obs_dim;
    }
    a.rows = rows; a.Ad = c->d.act_dim; a.G = c->PG; a.algo = c->d.algo;
    a.policy_noise = cfg ? cfg->policy_noise : 0.f;
    a.noise_clip = cfg ? cfg->noise_clip : 0.f;
    return a;
}
static int head_forward(morl_ac_ctx* c, Tape& tp, int rows, const float* eps, const morl_ac_state* st,
                        const morl_ac_cfg* cfg, float* action, float* logp, bool save, hipStream_t s,
                        Tape* qin = nullptr) {
    const HeadArgs a = head_args(c, tp, rows, eps, st, cfg, action, logp, save, qin);
    hipLaunchKernelGGL(ac_head_fwd_kernel, dim3((c->PG * rows + 255) / 256), dim3(256), 0, s, a);
    LAUNCH_CHECK("ac_head_fwd");
    return MORL_OK;
}

static int check_state(const morl_ac_ctx* c, const morl_ac_state* st, int rows) {
    if (!c || !st) return fail(MORL_ERR_ARG, "ctx / state is NULL");
    if (rows < 1 || rows > c->cap) return fail(MORL_ERR_STATE, "rows %d outside 1..max_rows %d", rows, c->cap);
    if (!st->pol || !st->q || !st->action_scale || !st->action_bias) return fail(MORL_ERR_ARG, "state has NULL network pointers");
    return MORL_OK;
}

// MOSAC with discrete actions (mosac_discrete_action.py:440-503): no action input to the critics, exact expectation over
// the actions instead of sampling, one actor step (and one alpha step) per call.
static int sacd_update(morl_ac_ctx* c, const morl_ac_state* st, const morl_ac_batch* bt, const morl_ac_cfg* cfg,
                       const morl_ac_out* out, int PG, hipStream_t s) {
    const morl_ac_desc& d = c->d;
    const int rows = bt->rows, D = d.obs_dim, A = d.act_dim, R = d.reward_dim;
    const Mlp &Q = c->q, &P = c->pol;
    const bool autotune = cfg->autotune != 0;
    const DropSpec nodrop;
    int rc;
    {
        auto fill = [&](ConcatArgs& a, float* dst, int ld, const float* src) {
            a.src[0] = src; a.width[0] = D; a.gstride[0] = (long long)rows * D; a.rstride[0] = D; a.n_src = 1;
            a.dst = dst; a.ld = ld; a.dst_gstride = (long long)c->cap * ld; a.rows = rows; a.G = PG;
        };
        ConcatMulti m{};
        m.n = 4;
        fill(m.c[0], c->tp_a.x, P.ld[0], bt->next_obs);
        fill(m.c[1], c->tq_a.x, Q.ld[0], bt->next_obs);
        fill(m.c[2], c->tq_b.x, Q.ld[0], bt->obs);
        fill(m.c[3], c->tp_b.x, P.ld[0], bt->obs);
        hipLaunchKernelGGL(ac_concat_multi_kernel, dim3(stream_grid((long long)PG * rows * Q.ld[0], 256, 512), 4), dim3(256), 0, s, m);
        LAUNCH_CHECK("sacd_inputs");
    }
    // K-major shadow copies for the forward GEMMs (single learners / small populations; see morl_ac_update)
    int widest = 0;
    for (int l = 1; l < Q.L; ++l) widest = std::max(widest, Q.dims[l]);
    constexpr bool shadow_env = true;
    // (a population whose every pass runs on the 32 / 64-row chain needs none: N-major weight stream, see chain_nmajor)
    const bool nmajor = chain_nmajor(Q, rows, c->QG) && chain_nmajor(P, rows, PG);
    const bool use_wt = shadow_env && !nmajor &&
                        (use_wave_tiles((long long)((rows + 127) / 128) * ((widest + 127) / 128) * c->QG * 2) ||
                         (chain_shape_ok(Q) && chain_shape_ok(P)));
    const float* wq = use_wt ? c->wt_q : nullptr;
    const float* wqt = use_wt ? c->wt_qt : nullptr;
    const float* wp = use_wt ? c->wt_pol : nullptr;
    if (use_wt) {
        TransposeMulti tm{};
        tm.n = 3;
        tm.src[0] = st->q; tm.dst[0] = c->wt_q; tm.lay[0] = layout_of(Q); tm.nets[0] = c->QG;
        tm.src[1] = st->q_target; tm.dst[1] = c->wt_qt; tm.lay[1] = layout_of(Q); tm.nets[1] = c->QG;
        tm.src[2] = st->pol; tm.dst[2] = c->wt_pol; tm.lay[2] = layout_of(P); tm.nets[2] = PG;
        if ((rc = launch_transposes(tm, s))) return rc;
    }
    // pi(s') and pi(s): the actor is stepped only at the end of the update, so both passes share their launches
    if ((rc = mlp_forward(P, st->pol, P.P, c->tp_a, rows, 1, nodrop, s, st->pol, &c->tp_b, &nodrop, wp, wp))) return rc;
    // target critics at s' and online critics at s: independent passes, one launch per layer
    if ((rc = mlp_forward(Q, st->q_target, Q.P, c->tq_a, rows, 2, nodrop, s, st->q, &c->tq_b, &nodrop, wqt, wq))) return rc;
    const long long q_gs = (long long)c->cap * Q.ld[Q.L], p_gs = (long long)c->cap * P.ld[P.L];
    {
        SacdCriticArgs a{};
        a.logits_next = c->tp_a.out; a.p_gstride = p_gs; a.ldp = P.ld[P.L];
        a.tq = c->tq_a.out; a.q = c->tq_b.out; a.dq = c->tq_b.g[Q.L - 1]; a.gstride = q_gs; a.ldo = Q.ld[Q.L];
        a.actions = bt->actions; a.rewards = bt->rewards; a.dones = bt->dones; a.w = bt->w;
        a.log_alpha = autotune ? st->log_alpha : nullptr; a.alpha_const = cfg->alpha;
        a.target_out = out->target_q; a.loss_out = out->critic_loss; a.q_losses = out->q_losses;
        a.rows = rows; a.A = A; a.R = R; a.gamma = cfg->gamma;
        hipLaunchKernelGGL(sacd_critic_kernel, dim3(PG), dim3(256), 0, s, a);
        LAUNCH_CHECK("sacd_critic");
    }
    if ((rc = mlp_backward(Q, st->q, Q.P, c->tq_b, rows, 2, false, c->gq, false, s))) return rc;
    if (out->q_grads)
        HIP_TRY(hipMemcpyAsync(out->q_grads, c->gq, (size_t)c->QG * Q.P * sizeof(float), hipMemcpyDeviceToDevice, s));
    if ((rc = adam(st->q, c->gq, st->q_exp_avg, st->q_exp_avg_sq, 2ll * Q.P, PG, cfg->q_lr, st->q_steps,
                   st->q_steps ? 1 : cfg->q_step, cfg, s, use_wt ? c->wt_q : nullptr, &Q))) return rc;
    // actor (+ alpha) through the UPDATED critics (the actor's logits at s are already in tp_b)
    if ((rc = mlp_forward(Q, st->q, Q.P, c->tq_b, rows, 2, nodrop, s, nullptr, nullptr, nullptr, wq))) return rc;
    {
        SacdActorArgs a{};
        a.logits = c->tp_b.out; a.dlogits = c->tp_b.g[P.L - 1]; a.p_gstride = p_gs; a.ldp = P.ld[P.L];
        a.q = c->tq_b.out; a.gstride = q_gs; a.ldo = Q.ld[Q.L];
        a.w = bt->w;
        a.log_alpha = st->log_alpha; a.la_m = st->log_alpha_exp_avg; a.la_v = st->log_alpha_exp_avg_sq;
        a.alpha_const = cfg->alpha; a.autotune = autotune ? 1 : 0; a.target_entropy = cfg->target_entropy;
        a.steps = st->pol_steps; a.step_add = st->pol_steps ? 1 : cfg->policy_step;
        a.alpha_lr = cfg->alpha_lr; a.b1 = cfg->beta1; a.b2 = cfg->beta2; a.eps = (float)cfg->eps;
        a.loss_out = out->policy_loss; a.alpha_loss_out = out->alpha_loss;
        a.rows = rows; a.A = A; a.R = R;
        hipLaunchKernelGGL(sacd_actor_kernel, dim3(PG), dim3(256), 0, s, a);
        LAUNCH_CHECK("sacd_actor");
    }
    if ((rc = mlp_backward(P, st->pol, P.P, c->tp_b, rows, 1, false, c->gp, false, s))) return rc;
    if (out->pol_grads)
        HIP_TRY(hipMemcpyAsync(out->pol_grads, c->gp, (size_t)PG * P.P * sizeof(float), hipMemcpyDeviceToDevice, s));
    if ((rc = adam(st->pol, c->gp, st->pol_exp_avg, st->pol_exp_avg_sq, P.P, PG, cfg->policy_lr, st->pol_steps,
                   st->pol_steps ? 1 : cfg->policy_step, cfg, s))) return rc;
    {
        int32_t* qs = st->q_steps;
        int32_t* ps = st->pol_steps;
        if (cfg->do_target) {
            const long long n = (long long)c->QG * Q.P;
            hipLaunchKernelGGL(ac_polyak_advance_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const float*)st->q,
                               st->q_target, n, cfg->tau, 1.0f - cfg->tau, qs, ps, PG, 1);
            LAUNCH_CHECK("ac_polyak_advance");
        } else if (qs || ps) {
            hipLaunchKernelGGL(ac_step_advance_kernel, dim3((PG + 63) / 64), dim3(64), 0, s, qs, ps, PG, 1);
            LAUNCH_CHECK("ac_step_advance");
        }
    }
    if (out->alpha) {
        hipLaunchKernelGGL(ac_alpha_prepare_kernel, dim3((PG + 63) / 64), dim3(64), 0, s, (const float*)st->log_alpha, cfg->alpha,
                           autotune ? 1 : 0, out->alpha, PG);
        LAUNCH_CHECK("ac_alpha_export");
    }
    return MORL_OK;
}

extern "C" int morl_ac_update(morl_ac_ctx* c, const morl_ac_state* st, const morl_ac_batch* bt, const morl_ac_cfg* cfg,
                              const morl_ac_out* out_in, void* stream) {
    if (!bt || !cfg) return fail(MORL_ERR_ARG, "batch / cfg is NULL");
    int rc = check_state(c, st, bt->rows);
    if (rc) return rc;
    static const morl_ac_out no_out{};
    const morl_ac_out* out = out_in ? out_in : &no_out;
    const morl_ac_desc& d = c->d;
    const int algo = d.algo, rows = bt->rows, D = d.obs_dim, Ad = d.act_dim, R = d.reward_dim, nq = d.num_q;
    hipStream_t s = (hipStream_t)stream;
    if (!st->q_target || !st->q_exp_avg || !st->q_exp_avg_sq || !st->pol_exp_avg || !st->pol_exp_avg_sq)
        return fail(MORL_ERR_ARG, "state has NULL optimiser / target pointers");
    if (algo == MORL_AC_TD3 && !st->pol_target) return fail(MORL_ERR_ARG, "TD3 needs pol_target");
    if (!bt->obs || !bt->actions || !bt->rewards || !bt->next_obs || !bt->dones || !bt->w ||
        (!bt->eps_next && algo != MORL_AC_SACD))
        return fail(MORL_ERR_ARG, "batch has NULL arrays");
    const bool autotune = (algo == MORL_AC_MOSAC || algo == MORL_AC_SACD) && cfg->autotune;
    if (autotune && (!st->log_alpha || !st->log_alpha_exp_avg || !st->log_alpha_exp_avg_sq))
        return fail(MORL_ERR_ARG, "autotune needs log_alpha and its Adam state");
    const int iters = (algo == MORL_AC_MOSAC) ? std::max(1, cfg->policy_iters) : 1;
    if (cfg->do_policy && algo != MORL_AC_TD3 && algo != MORL_AC_SACD && !bt->eps_pi) return fail(MORL_ERR_ARG, "eps_pi is NULL");
    if (cfg->do_policy && autotune && algo == MORL_AC_MOSAC && !bt->eps_alpha) return fail(MORL_ERR_ARG, "eps_alpha is NULL");
    if (cfg->n_per < 0 || cfg->n_per > rows) return fail(MORL_ERR_ARG, "n_per %d outside 0..rows", cfg->n_per);
    if (cfg->grad_hook) {
        if (autotune || algo == MORL_AC_SACD)
            return fail(MORL_ERR_ARG, "grad_hook: a learnt entropy coefficient / discrete SAC is not data-parallel here");
        if (!out->q_grads || (cfg->do_policy && !out->pol_grads))
            return fail(MORL_ERR_ARG, "grad_hook needs out->q_grads (and out->pol_grads with do_policy)");
    }
    const Mlp &Q = c->q, &P = c->pol;
    // learners advanced by this call (a leading sub-range of the context's capacity)
    if (bt->active < 0 || bt->active > d.population) return fail(MORL_ERR_STATE, "active %d outside 0..population %d", bt->active, d.population);
    const int PG = bt->active > 0 ? bt->active : d.population, QG = PG * nq;
    c->PG = PG; c->QG = QG;
    c->tq_a.G = c->tq_b.G = QG;
    c->tp_a.G = c->tp_b.G = PG;
    int64_t mask_net = 0;                                    // explicit dropout masks: bytes per net and per phase
    for (int l = 0; l < Q.L - 1; ++l) mask_net += (int64_t)rows * Q.dims[l + 1];
    const int64_t mask_phase = (int64_t)QG * mask_net;
    const float* w_rows = c->w_input ? bt->w : nullptr;      // weight vector as a network input
    const int wR = c->w_input ? R : 0;
    auto dropspec = [&](int phase) {
        DropSpec ds;
        ds.active = Q.drop > 0.f;
        ds.ext = bt->drop_masks ? bt->drop_masks + phase * mask_phase : nullptr;
        ds.ext_net_bytes = mask_net;
        ds.seed = cfg->dropout_seed * 4 + (unsigned long long)phase;
        return ds;
    };
    const DropSpec nodrop;
    const long long q_ldo = Q.ld[Q.L], q_gs = (long long)c->cap * q_ldo;

    const float* la = autotune ? st->log_alpha : nullptr;    // entropy coefficient source of every kernel below
    if (algo == MORL_AC_SACD) return sacd_update(c, st, bt, cfg, out, PG, s);

    // ---- every network input of the update in one launch: policy at s' / s, critics at (s', .) / (s, a) -----------------------
    // a ~ pi(s) of the actor phase's first iteration is sampled next to a' ~ pi(s') (one head launch instead of two): it then needs
    // critic input rows of its own, the critic phase is not done with (s, a, w)
    static const bool heads_env = [] { const char* e = getenv("MORL_AC_HEADS_PAIRED"); return e ? atoi(e) != 0 : true; }();   // (A/B)
    const bool heads_early = heads_env && cfg->do_policy != 0;
    c->tq_b.x = c->xq_b;
    ConcatMulti m{};
    {
        auto fill = [&](ConcatArgs& a, float* dst, int ld, const float* s0, int w0, const float* s1, int w1, const float* s2, int w2) {
            const float* src[3] = {s0, s1, s2};
            const int wd[3] = {w0, w1, w2};
            int n = 0;
            for (int k = 0; k < 3; ++k)
                if (wd[k] > 0) {
                    a.src[n] = src[k]; a.width[n] = wd[k];
                    a.gstride[n] = (long long)rows * wd[k]; a.rstride[n] = wd[k];
                    ++n;
                }
            a.n_src = n;
            a.dst = dst; a.ld = ld; a.dst_gstride = (long long)c->cap * ld; a.rows = rows; a.G = PG;
        };
        m.n = 4;
        fill(m.c[0], c->tp_a.x, P.ld[0], bt->next_obs, D, w_rows, wR, nullptr, 0);
        fill(m.c[1], c->tq_a.x, Q.ld[0], bt->next_obs, D, nullptr, Ad, w_rows, wR);      // a' is written by the head kernel
        fill(m.c[2], c->tq_b.x, Q.ld[0], bt->obs, D, bt->actions, Ad, w_rows, wR);
        fill(m.c[3], c->tp_b.x, P.ld[0], bt->obs, D, w_rows, wR, nullptr, 0);
        if (heads_early) {       // the actor phase's critic input rows (s, pi(s), w): pi(s) is written by the paired head launch
            m.n = 5;
            fill(m.c[4], c->xq_pi, Q.ld[0], bt->obs, D, nullptr, Ad, w_rows, wR);
        }
    }
    const int concat_bx = stream_grid((long long)PG * rows * Q.ld[0], 256, 512);

    // ---- K-major shadow copies of every parameter set this update reads in a forward pass (one launch) -----------------------
    constexpr bool shadow_env = true;
    // only where the wave-tile engine runs the layers (single learners, small populations): the LDS-tiled engine of a large
    // population stages 128 x 32 chunks either way and measured slower with the K-major weights (1.70 vs 1.27 ms at 64 learners)
    int widest = 0;
    for (int l = 1; l < Q.L; ++l) widest = std::max(widest, Q.dims[l]);
    // ... and for the small populations whose networks the 16-row chain takes (it reads K-major weights); a population whose
    // every pass runs on the 32 / 64-row chain needs none (N-major weight stream, see chain_nmajor)
    const bool nmajor = chain_nmajor(Q, rows, QG) && chain_nmajor(P, rows, c->PG);
    const bool use_wt = shadow_env && !nmajor &&
                        (use_wave_tiles((long long)((rows + 127) / 128) * ((widest + 127) / 128) * QG * 2) ||
                         (chain_shape_ok(Q) && chain_shape_ok(P)));
#define WT(p) (use_wt ? (p) : nullptr)
    if (use_wt) {
        TransposeMulti tm{};
        const bool td3 = algo == MORL_AC_TD3;
        const float* srcs[4] = {st->q, st->q_target, st->pol, td3 ? st->pol_target : nullptr};
        float* dsts[4] = {c->wt_q, c->wt_qt, c->wt_pol, c->wt_polt};
        for (int k = 0; k < 4; ++k) {
            if (!srcs[k]) continue;
            const bool is_q = k < 2;
            tm.src[tm.n] = srcs[k]; tm.dst[tm.n] = dsts[k];
            tm.lay[tm.n] = layout_of(is_q ? Q : P);
            tm.nets[tm.n] = is_q ? QG : PG;
            ++tm.n;
        }
        // a single learner: the shadow copies are scattered by extra workgroups of the input-assembly launch
        long long longest = 0;
        for (int k = 0; k < tm.n; ++k) longest = std::max(longest, tm.lay[k].P * tm.nets[k]);
        static const long long scatter_max = [] { const char* e = getenv("MORL_AC_SCATTER_MAX"); return e ? atoll(e) : (1ll << 20); }();
        if (tm.n >= 1 && longest >= 1 && longest <= scatter_max) {
            const int tx = stream_grid(longest, 256, 1024);
            hipLaunchKernelGGL(ac_inputs_shadows_kernel, dim3(concat_bx * m.n + tx * tm.n), dim3(256), 0, s, m, tm, concat_bx, tx);
            LAUNCH_CHECK("ac_inputs_shadows");
        } else {
            hipLaunchKernelGGL(ac_concat_multi_kernel, dim3(concat_bx, m.n), dim3(256), 0, s, m);
            LAUNCH_CHECK("ac_inputs");
            if ((rc = launch_transposes(tm, s))) return rc;
        }
    } else {
        hipLaunchKernelGGL(ac_concat_multi_kernel, dim3(concat_bx, m.n), dim3(256), 0, s, m);
        LAUNCH_CHECK("ac_inputs");
    }

    // ---- critic phase: a' ~ pi(s'), target critics at (s', a'), critics at (s, a), TD loss, backward, Adam --------------
    // the actor phase's first forward, pi(s), uses the not-yet-updated actor too: it shares the launches of pi(s')
    const bool pi_s_early = cfg->do_policy != 0;
    if ((rc = mlp_forward(P, algo == MORL_AC_TD3 ? st->pol_target : st->pol, P.P, c->tp_a, rows, 1, nodrop, s,
                          pi_s_early ? st->pol : nullptr, pi_s_early ? &c->tp_b : nullptr, &nodrop,
                          algo == MORL_AC_TD3 ? WT(c->wt_polt) : WT(c->wt_pol), pi_s_early ? WT(c->wt_pol) : nullptr)))
        return rc;
    if (heads_early) {
        const HeadArgs h_next = head_args(c, c->tp_a, rows, bt->eps_next, st, cfg, c->act, c->logp_next, false, &c->tq_a);
        Tape xpi = c->tq_b;
        xpi.x = c->xq_pi;
        const HeadArgs h_pi = head_args(c, c->tp_b, rows, (algo == MORL_AC_TD3) ? nullptr : bt->eps_pi, st, cfg, c->act_pi,
                                        c->logp_pi, true, &xpi);
        hipLaunchKernelGGL(ac_head_fwd_pair_kernel, dim3((c->PG * rows + 255) / 256, 2), dim3(256), 0, s, h_next, h_pi);
        LAUNCH_CHECK("ac_head_fwd_pair");
    } else if ((rc = head_forward(c, c->tp_a, rows, bt->eps_next, st, cfg, c->act, c->logp_next, false, s, &c->tq_a))) return rc;
    {
        // target critics at (s', a') and online critics at (s, a): independent passes, one launch per layer
        const DropSpec d0 = dropspec(0), d1 = dropspec(1);
        if ((rc = mlp_forward(Q, st->q_target, Q.P, c->tq_a, rows, nq, d0, s, st->q, &c->tq_b, &d1, WT(c->wt_qt), WT(c->wt_q)))) return rc;
    }
    // the optimiser step (+ the K-major shadow copy, + the Polyak average of the target critics, which nothing reads before the
    // next update) rides in the weight-gradient launch when nobody asks for the gradients themselves
    static const bool adam_in_dw = [] { const char* e = getenv("MORL_AC_ADAM_IN_DW"); return e ? atoi(e) != 0 : true; }();   // (A/B)
    static const long long scatter_max = [] { const char* e = getenv("MORL_AC_SCATTER_MAX"); return e ? atoll(e) : (1ll << 20); }();
    const bool may_fuse = adam_in_dw && !cfg->grad_hook;
    const bool offer_q = may_fuse && !out->q_grads && (!use_wt || (long long)QG * Q.P <= scatter_max);
    const bool offer_p = may_fuse && !out->pol_grads && (!use_wt || (long long)PG * P.P <= scatter_max);
    {
        CriticArgs a{};
        if (offer_q) a.corr = AdamCorrOut{c->adam_corr, st->q_steps, st->q_steps ? 1 : cfg->q_step, cfg->q_lr, cfg->beta1, cfg->beta2};
        a.tq = c->tq_a.out; a.q = c->tq_b.out; a.dq = c->tq_b.g[Q.L - 1];
        a.gstride = q_gs; a.ldo = (int)q_ldo;
        a.logp_next = c->logp_next; a.rewards = bt->rewards; a.dones = bt->dones;
        a.w = bt->w; a.w_per_row = c->w_input ? 1 : 0;
        a.log_alpha = la; a.alpha_const = cfg->alpha;
        a.target_out = out->target_q; a.loss_out = out->critic_loss; a.q_losses = out->q_losses;
        a.priority = (algo == MORL_AC_TD3 && cfg->n_per > 0) ? out->priority : nullptr;
        a.n_per = cfg->n_per;
        a.rows = rows; a.R = R; a.nq = nq; a.algo = algo; a.gamma = cfg->gamma;
        hipLaunchKernelGGL(ac_critic_kernel, dim3(c->PG), dim3(256), 0, s, a);
        LAUNCH_CHECK("ac_critic");
    }
    bool q_fused = false, q_target_fused = false, steps_advanced = false;
    {
        AdamFuse f{};
        f.corr = c->adam_corr;
        if (!cfg->do_policy) f.adv_q = st->q_steps;          // nothing behind this launch reads the counter
        f.params = st->q; f.exp_avg = st->q_exp_avg; f.exp_avg_sq = st->q_exp_avg_sq;
        f.wt = WT(c->wt_q);
        f.target = cfg->do_target ? st->q_target : nullptr;
        f.steps = st->q_steps; f.step_add = st->q_steps ? 1 : cfg->q_step; f.nets_per_learner = nq;
        f.lr = cfg->q_lr; f.b1 = cfg->beta1; f.b2 = cfg->beta2; f.eps = (float)cfg->eps; f.tau = cfg->tau;
        if ((rc = mlp_backward(Q, st->q, Q.P, c->tq_b, rows, nq, Q.drop > 0.f, c->gq, false, s, -1, WT(c->wt_q),
                               offer_q ? &f : nullptr, &q_fused))) return rc;
        q_target_fused = q_fused && cfg->do_target;
        steps_advanced = q_fused && !cfg->do_policy;
    }
    if (out->q_grads)
        HIP_TRY(hipMemcpyAsync(out->q_grads, c->gq, (size_t)c->QG * Q.P * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (cfg->grad_hook) {
        // data-parallel job: the caller's buffer is reduced over the processes, then becomes the gradient of the step
        if (cfg->grad_hook(cfg->grad_hook_user, 0, out->q_grads, (int64_t)c->QG * Q.P, stream))
            return fail(MORL_ERR_STATE, "grad_hook failed on the critic gradients");
        HIP_TRY(hipMemcpyAsync(c->gq, out->q_grads, (size_t)c->QG * Q.P * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (!q_fused && (rc = adam(st->q, c->gq, st->q_exp_avg, st->q_exp_avg_sq, (long long)nq * Q.P, PG, cfg->q_lr, st->q_steps,
                               st->q_steps ? 1 : cfg->q_step, cfg, s, WT(c->wt_q), &Q))) return rc;

    // ---- actor phase ---------------------------------------------------------------------------------------------------
    if (cfg->do_policy) {
        for (int it = 0; it < iters; ++it) {
            const float* eps_pi = (algo == MORL_AC_TD3) ? nullptr : bt->eps_pi + (long long)it * c->PG * rows * Ad;
            // the critics' input rows (obs | . | w) are already in tq_b; the head kernel overwrites the action columns with pi(s).
            // The first iteration's pre-activations were computed next to pi(s') above.  From the second iteration on with a
            // learnt alpha, the actor's trunk and head pre-activations at s are already in tp_b: the alpha re-sample below ran the UPDATED actor on the same rows and nothing has changed it since --
            // only the noise differs, which enters in head_forward
            const bool trunk_current = (it == 0 && pi_s_early) || (it > 0 && autotune && algo == MORL_AC_MOSAC);
            if (!trunk_current && (rc = mlp_forward(P, st->pol, P.P, c->tp_b, rows, 1, nodrop, s, nullptr, nullptr, nullptr, WT(c->wt_pol))))
                return rc;
            if (it == 0 && heads_early) c->tq_b.x = c->xq_pi;     // (restored at the start of the next update)
            else if ((rc = head_forward(c, c->tp_b, rows, eps_pi, st, cfg, c->act, c->logp_pi, true, s, &c->tq_b))) return rc;
            if ((rc = mlp_forward(Q, st->q, Q.P, c->tq_b, rows, nq, dropspec(2), s, nullptr, nullptr, nullptr, WT(c->wt_q)))) return rc;
            {
                ActorLossArgs a{};
                if (offer_p)
                    a.corr = AdamCorrOut{c->adam_corr + 2 * c->PG, st->pol_steps, (st->pol_steps ? 1 : cfg->policy_step) + it,
                                         cfg->policy_lr, cfg->beta1, cfg->beta2};
                a.q = c->tq_b.out; a.dq = c->tq_b.g[Q.L - 1];
                a.gstride = q_gs; a.ldo = (int)q_ldo;
                a.logp = c->logp_pi; a.w = bt->w; a.w_per_row = c->w_input ? 1 : 0;
                a.log_alpha = la; a.alpha_const = cfg->alpha;
                a.loss_out = out->policy_loss;
                a.rows = rows; a.R = R; a.nq = nq; a.algo = algo;
                hipLaunchKernelGGL(ac_actor_loss_kernel, dim3(c->PG), dim3(256), 0, s, a);
                LAUNCH_CHECK("ac_actor_loss");
            }
            if ((rc = mlp_backward(Q, st->q, Q.P, c->tq_b, rows, nq, Q.drop > 0.f, nullptr, true, s, -1, WT(c->wt_q)))) return rc;
            HeadBwdArgs hb{};
            {
                HeadBwdArgs a{};
                a.dx_q = c->tq_b.dx; a.dxq_gstride = (long long)c->cap * Q.ld[0];
                a.nq = nq; a.ld_qin = Q.ld[0]; a.col0 = D;
                a.head = c->tp_b.out; a.head_gstride = (long long)c->cap * P.ld[P.L]; a.ldh = P.ld[P.L];
                a.eps = eps_pi; a.save_y = c->save_y; a.save_std = c->save_std; a.logp = c->logp_pi;
                a.scale = st->action_scale; a.log_alpha = la; a.alpha_const = cfg->alpha;
                a.dhead = c->tp_b.g[P.L - 1];
                a.rows = rows; a.Ad = Ad; a.G = c->PG; a.algo = algo;
                hb = a;
            }
            bool p_fused = false;
            {
                AdamFuse f{};
                f.params = st->pol; f.exp_avg = st->pol_exp_avg; f.exp_avg_sq = st->pol_exp_avg_sq;
                f.wt = WT(c->wt_pol);
                f.target = (algo == MORL_AC_TD3) ? st->pol_target : nullptr;
                f.steps = st->pol_steps; f.step_add = (st->pol_steps ? 1 : cfg->policy_step) + it; f.nets_per_learner = 1;
                f.lr = cfg->policy_lr; f.b1 = cfg->beta1; f.b2 = cfg->beta2; f.eps = (float)cfg->eps; f.tau = cfg->tau;
                f.corr = c->adam_corr + 2 * c->PG;
                const bool last_reader = (it == iters - 1) && !autotune;       // (with a learnt alpha: its step kernel below)
                if (last_reader) { f.adv_q = st->q_steps; f.adv_p = st->pol_steps; f.adv_p_by = iters; }
                if ((rc = mlp_backward(P, st->pol, P.P, c->tp_b, rows, 1, false, c->gp, false, s, -1, WT(c->wt_pol),
                                       offer_p ? &f : nullptr, &p_fused, &hb))) return rc;
                if (p_fused && last_reader) steps_advanced = true;
            }
            if (out->pol_grads)
                HIP_TRY(hipMemcpyAsync(out->pol_grads, c->gp, (size_t)c->PG * P.P * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (cfg->grad_hook) {
                if (cfg->grad_hook(cfg->grad_hook_user, 1, out->pol_grads, (int64_t)c->PG * P.P, stream))
                    return fail(MORL_ERR_STATE, "grad_hook failed on the actor gradients");
                HIP_TRY(hipMemcpyAsync(c->gp, out->pol_grads, (size_t)c->PG * P.P * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
            if (!p_fused && (rc = adam(st->pol, c->gp, st->pol_exp_avg, st->pol_exp_avg_sq, P.P, PG, cfg->policy_lr, st->pol_steps,
                                       (st->pol_steps ? 1 : cfg->policy_step) + it, cfg, s, WT(c->wt_pol), &P))) return rc;
            if (autotune) {
                // log-prob of a fresh sample under the UPDATED actor (mosac_continuous_action.py:467-468)
                const float* eps_al = bt->eps_alpha + (long long)it * c->PG * rows * Ad;
                if ((rc = mlp_forward(P, st->pol, P.P, c->tp_b, rows, 1, nodrop, s, nullptr, nullptr, nullptr, WT(c->wt_pol)))) return rc;
                if ((rc = head_forward(c, c->tp_b, rows, eps_al, st, cfg, c->act, c->logp_pi, false, s))) return rc;
                hipLaunchKernelGGL(ac_alpha_step_kernel, dim3(c->PG), dim3(256), 0, s, st->log_alpha, st->log_alpha_exp_avg,
                                   st->log_alpha_exp_avg_sq, (const float*)c->logp_pi, rows, cfg->target_entropy,
                                   (const int*)st->pol_steps, (st->pol_steps ? 1 : cfg->policy_step) + it, cfg->alpha_lr,
                                   cfg->beta1, cfg->beta2, (float)cfg->eps, out->alpha_loss,
                                   it == iters - 1 ? st->q_steps : nullptr, it == iters - 1 ? st->pol_steps : nullptr, iters);
                LAUNCH_CHECK("ac_alpha_step");
                if (it == iters - 1) steps_advanced = true;
            }
            if (algo == MORL_AC_TD3 && !p_fused)
                if ((rc = polyak(st->pol, st->pol_target, (long long)c->PG * P.P, cfg->tau, s))) return rc;
        }
    }
    c->tq_b.x = c->xq_b;
    {
        int32_t* qs = steps_advanced ? nullptr : st->q_steps;
        int32_t* ps = (cfg->do_policy && !steps_advanced) ? st->pol_steps : nullptr;
        if (cfg->do_target && !q_target_fused) {
            const long long n = (long long)c->QG * Q.P;
            hipLaunchKernelGGL(ac_polyak_advance_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const float*)st->q,
                               st->q_target, n, cfg->tau, 1.0f - cfg->tau, qs, ps, PG, iters);
            LAUNCH_CHECK("ac_polyak_advance");
        } else if (qs || ps) {
            hipLaunchKernelGGL(ac_step_advance_kernel, dim3((PG + 63) / 64), dim3(64), 0, s, qs, ps, PG, iters);
            LAUNCH_CHECK("ac_step_advance");
        }
    }
    if (out->alpha) {
        hipLaunchKernelGGL(ac_alpha_prepare_kernel, dim3((PG + 63) / 64), dim3(64), 0, s, (const float*)st->log_alpha, cfg->alpha,
                           autotune ? 1 : 0, out->alpha, PG);
        LAUNCH_CHECK("ac_alpha_export");
    }
    return MORL_OK;
}

#undef WT
extern "C" int morl_ac_policy_forward(morl_ac_ctx* c, const morl_ac_state* st, const float* obs, const float* w, int rows,
                                      int mode, const float* eps, int use_target, const morl_ac_cfg* cfg,
                                      float* actions_out, float* logp_out, void* stream) {
    int rc = check_state(c, st, rows);
    if (rc) return rc;
    if (!obs || !actions_out) return fail(MORL_ERR_ARG, "obs / actions_out is NULL");
    if (c->w_input && !w) return fail(MORL_ERR_ARG, "this policy is weight-conditioned: w is NULL");
    if (mode == 1 && !eps) return fail(MORL_ERR_ARG, "mode 1 needs eps");
    if (use_target && !st->pol_target) return fail(MORL_ERR_ARG, "pol_target is NULL");
    hipStream_t s = (hipStream_t)stream;
    const Mlp& P = c->pol;
    c->PG = c->d.population; c->QG = c->PG * c->d.num_q;
    c->tp_a.G = c->PG;
    if ((rc = concat(c, c->tp_a.x, P.ld[0], c->PG, rows, obs, c->d.obs_dim, c->w_input ? w : nullptr,
                     c->w_input ? c->d.reward_dim : 0, nullptr, 0, s))) return rc;
    if ((rc = mlp_forward(P, use_target ? st->pol_target : st->pol, P.P, c->tp_a, rows, 1, DropSpec(), s))) return rc;
    if (c->d.algo == MORL_AC_SACD) {   // discrete actor: hand back the logits [pop][rows][A]; sampling is the caller's
        const int A = c->d.act_dim;
        ConcatArgs a{};
        a.src[0] = c->tp_a.out; a.width[0] = A; a.gstride[0] = (long long)c->cap * P.ld[P.L]; a.rstride[0] = P.ld[P.L];
        a.n_src = 1;
        a.dst = actions_out; a.ld = A; a.dst_gstride = (long long)rows * A; a.rows = rows; a.G = c->PG;
        hipLaunchKernelGGL(ac_concat_kernel, dim3(stream_grid((long long)c->PG * rows * A, 256)), dim3(256), 0, s, a);
        LAUNCH_CHECK("sacd_logits");
        return MORL_OK;
    }
    return head_forward(c, c->tp_a, rows, mode == 1 ? eps : nullptr, st, cfg, actions_out, logp_out, false, s);
}

extern "C" int morl_ac_q_forward(morl_ac_ctx* c, const morl_ac_state* st, const float* obs, const float* actions,
                                 const float* w, int rows, int use_target, float* q_out, void* stream) {
    int rc = check_state(c, st, rows);
    if (rc) return rc;
    const bool sacd = c && c->d.algo == MORL_AC_SACD;
    if (!obs || (!actions && !sacd) || !q_out) return fail(MORL_ERR_ARG, "obs / actions / q_out is NULL");
    if (c->w_input && !w) return fail(MORL_ERR_ARG, "this critic is weight-conditioned: w is NULL");
    if (use_target && !st->q_target) return fail(MORL_ERR_ARG, "q_target is NULL");
    hipStream_t s = (hipStream_t)stream;
    const Mlp& Q = c->q;
    const int R = Q.dims[Q.L];                   // values per row: R, or A * R for the discrete critics
    c->PG = c->d.population; c->QG = c->PG * c->d.num_q;
    c->tq_a.G = c->QG;
    if ((rc = concat(c, c->tq_a.x, Q.ld[0], c->PG, rows, obs, c->d.obs_dim, sacd ? nullptr : actions, sacd ? 0 : c->d.act_dim,
                     c->w_input ? w : nullptr, c->w_input ? c->d.reward_dim : 0, s))) return rc;
    if ((rc = mlp_forward(Q, use_target ? st->q_target : st->q, Q.P, c->tq_a, rows, c->d.num_q, DropSpec(), s))) return rc;
    // compact [QG][cap][ld] -> [QG][rows][R]
    ConcatArgs a{};
    a.src[0] = c->tq_a.out; a.width[0] = R; a.gstride[0] = (long long)c->cap * Q.ld[Q.L]; a.rstride[0] = Q.ld[Q.L];
    a.n_src = 1;
    a.dst = q_out; a.ld = R; a.dst_gstride = (long long)rows * R; a.rows = rows; a.G = c->QG;
    hipLaunchKernelGGL(ac_concat_kernel, dim3(stream_grid((long long)c->QG * rows * R, 256)), dim3(256), 0, s, a);
    LAUNCH_CHECK("ac_q_compact");
    return MORL_OK;
}

// =====================================================================================================================
// GPI-PD / GPI-LS with discrete actions (include/morl_hip.h, "GPI-PD / GPI-LS with discrete actions")
// =====================================================================================================================
#include "gpi_kernels.h"

namespace {

struct GpiTape {
    Tape t;                    // trunk `net` activations (t.x = sf * wf)
    float* sf = nullptr;       // [nn][cap][ldH]
    float* wf = nullptr;
    float* dwf = nullptr;
};

__global__ __launch_bounds__(256) void gpi_expand_kernel(const float* __restrict__ obs, int D,
                                                         const float* __restrict__ support, int R, int K, int rows,
                                                         float* __restrict__ obs_rep, float* __restrict__ w_rep) {
    const long long total = (long long)rows * K * (D + R);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % (D + R));
        const long long rk = e / (D + R);
        const int k = (int)(rk % K);
        const long long row = rk / K;
        if (c < D) obs_rep[rk * D + c] = obs[row * D + c];
        else w_rep[rk * R + (c - D)] = support[(long long)k * R + (c - D)];
    }
}

}  // namespace

struct morl_gpi_ctx {
    morl_gpi_desc d{};
    Mlp net;
    int H0 = 0, ldH = 0, nn = 0, cap = 0, cap_env = 0;
    int64_t offWw = 0, offBw = 0, offWs = 0, offBs = 0, offNet = 0, P = 0;
    GpiTape tt, te, tq;        // target nets on rows, target nets on rows * K, online nets on rows (kept for backward)
    float* grads = nullptr;    // [nn][P]
    float* target = nullptr;   // [cap][R]
    float* target_env = nullptr;
    float* obs_rep = nullptr;  // [cap_env][D]
    float* w_rep = nullptr;    // [cap_env][R]
    float* qmin = nullptr;     // [ldq]
    float* wt_q = nullptr;     // [nn][P] K-major shadow copies of the trunk matrices (ac_kernels.h: MlpLayout), online ...
    float* wt_qt = nullptr;    // ... and target ensemble
    std::vector<void*> allocs;
};

static int gpi_fill(const morl_gpi_desc* d, morl_gpi_ctx& c) {
    if (!d) return fail(MORL_ERR_ARG, "desc is NULL");
    if (d->n_hidden < 2 || d->n_hidden > MORL_MAX_LAYERS) return fail(MORL_ERR_ARG, "n_hidden %d not in 2..%d", d->n_hidden, MORL_MAX_LAYERS);
    if (d->obs_dim < 1 || d->n_actions < 1 || d->reward_dim < 1 || d->reward_dim > MORL_MAX_OBJ)
        return fail(MORL_ERR_ARG, "bad dims D=%d A=%d R=%d", d->obs_dim, d->n_actions, d->reward_dim);
    if (d->num_nets < 1 || d->num_nets > 4) return fail(MORL_ERR_ARG, "num_nets %d not in 1..4", d->num_nets);
    if (d->drop_rate < 0.f || d->drop_rate >= 1.f) return fail(MORL_ERR_ARG, "drop rate %g", (double)d->drop_rate);
    for (int l = 0; l < d->n_hidden; ++l)
        if (d->hidden[l] < 1 || d->hidden[l] > 64 * POST_MAXJ) return fail(MORL_ERR_ARG, "hidden[%d] = %d", l, d->hidden[l]);
    c.d = *d;
    c.nn = d->num_nets;
    c.H0 = d->hidden[0];
    c.ldH = round_up(c.H0, 4);
    c.net = Mlp();
    c.net.L = d->n_hidden;                       // (n_hidden - 1) hidden layers + output
    for (int l = 0; l < d->n_hidden; ++l) c.net.dims[l] = d->hidden[l];
    c.net.dims[d->n_hidden] = d->n_actions * d->reward_dim;
    c.net.ln = d->layer_norm != 0;
    c.net.drop = d->drop_rate;
    c.net.finish();
    int64_t o = 0;
    c.offWw = o; o += (int64_t)c.H0 * d->reward_dim;
    c.offBw = o; o += c.H0;
    c.offWs = o; o += (int64_t)c.H0 * d->obs_dim;
    c.offBs = o; o += c.H0;
    c.offNet = o;
    c.P = o + c.net.P;
    return MORL_OK;
}

extern "C" int64_t morl_gpi_param_count(const morl_gpi_desc* d) {
    morl_gpi_ctx c;
    return gpi_fill(d, c) ? -1 : c.P;
}

extern "C" int64_t morl_gpi_mask_bytes(const morl_gpi_desc* d, int rows) {
    morl_gpi_ctx c;
    if (gpi_fill(d, c)) return -1;
    int64_t per_net = 0;
    for (int l = 0; l < c.net.L - 1; ++l) per_net += (int64_t)rows * c.net.dims[l + 1];
    return (int64_t)c.nn * per_net;
}

extern "C" int morl_gpi_destroy(morl_gpi_ctx* c) {
    if (!c) return MORL_OK;
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    return MORL_OK;
}

static int gpi_alloc_tape(morl_gpi_ctx* c, GpiTape& g, int cap, bool post) {
    int rc;
    if ((rc = alloc_tape(c->allocs, c->net, g.t, c->nn, c->nn, cap, post))) return rc;
    const size_t n = (size_t)c->nn * cap * c->ldH;
    if ((rc = alloc_f(c->allocs, &g.sf, n)) || (rc = alloc_f(c->allocs, &g.wf, n)) || (rc = alloc_f(c->allocs, &g.dwf, n))) return rc;
    return MORL_OK;
}

extern "C" int morl_gpi_create(morl_gpi_ctx** out, const morl_gpi_desc* d) {
    if (!out) return fail(MORL_ERR_ARG, "out is NULL");
    *out = nullptr;
    morl_gpi_ctx* c = new (std::nothrow) morl_gpi_ctx();
    if (!c) return fail(MORL_ERR_ALLOC, "out of host memory");
    int rc = gpi_fill(d, *c);
    if (rc) { delete c; return rc; }
    if (d->max_rows < 1 || d->max_support < 1) { delete c; return fail(MORL_ERR_ARG, "max_rows %d / max_support %d", d->max_rows, d->max_support); }
    c->cap = d->max_rows;
    c->cap_env = d->max_rows * d->max_support;
    const bool post = c->net.ln || c->net.drop > 0.f;
    const int ldq = c->net.ld[c->net.L];
    if ((rc = gpi_alloc_tape(c, c->tt, c->cap, post)) || (rc = gpi_alloc_tape(c, c->te, c->cap_env, post)) ||
        (rc = gpi_alloc_tape(c, c->tq, c->cap, post)) || (rc = alloc_f(c->allocs, &c->grads, (size_t)c->nn * c->P)) ||
        (rc = alloc_f(c->allocs, &c->wt_q, (size_t)c->nn * c->P)) || (rc = alloc_f(c->allocs, &c->wt_qt, (size_t)c->nn * c->P)) ||
        (rc = alloc_f(c->allocs, &c->target, (size_t)c->cap * d->reward_dim)) ||
        (rc = alloc_f(c->allocs, &c->target_env, (size_t)c->cap * d->reward_dim)) ||
        (rc = alloc_f(c->allocs, &c->obs_rep, (size_t)c->cap_env * d->obs_dim)) ||
        (rc = alloc_f(c->allocs, &c->w_rep, (size_t)c->cap_env * d->reward_dim)) || (rc = alloc_f(c->allocs, &c->qmin, ldq))) {
        morl_gpi_destroy(c);
        return rc;
    }
    if (hipDeviceSynchronize() != hipSuccess) { morl_gpi_destroy(c); return fail(MORL_ERR_HIP, "workspace init failed"); }
    *out = c;
    return MORL_OK;
}

// QNet.forward of n_nets nets (params + g * P) on `rows` inputs shared by the nets
// params2 / g2 / obs2 / ds2: a second, independent pass (same number of nets, rows and weight rows -- the target ensemble at
// s' next to the online ensemble at s) whose GEMMs share the launches of the first (GemmBatched::split)
static int gpi_forward(morl_gpi_ctx* c, const float* params, int n_nets, GpiTape& g, const float* obs, const float* w,
                       int w_rstride, int rows, const DropSpec& ds, hipStream_t s, const float* params2 = nullptr,
                       GpiTape* g2 = nullptr, const float* obs2 = nullptr, const DropSpec* ds2 = nullptr,
                       const float* wt = nullptr, const float* wt2 = nullptr) {
    // wt / wt2: K-major shadow copies of params / params2 for the trunk layers (morl_gpi_update fills them)
    const morl_gpi_desc& d = c->d;
    g.t.G = n_nets;
    if (g2) g2->t.G = n_nets;
    {   // sf = relu(obs @ Ws^T + bs): every net reads the same obs rows
        GemmBatched b{};
        GemmProblem& p = b.p;
        p.A = obs; p.lda = d.obs_dim; b.sA = 0; b.a_div = 1;
        p.B = params + c->offWs; p.ldb = d.obs_dim; b.sB = c->P;
        p.bias = params + c->offBs; b.sBias = c->P;
        p.C = g.sf; p.ldc = c->ldH; b.sC = (long long)g.t.cap * c->ldH;
        p.M = rows; p.N = c->H0; p.K = d.obs_dim;
        if (g2) {
            if (g2->t.cap != g.t.cap) return fail(MORL_ERR_STATE, "paired passes need equally shaped tapes");
            b.split = n_nets;
            b.A2 = obs2; b.B2 = params2 + c->offWs; b.bias2 = params2 + c->offBs; b.C2 = g2->sf;
        }
        int rc = launch_bgemm<true, true, EPI_BIAS_RELU>(b, g2 ? 2 * n_nets : n_nets, s, "gpi_gemm_sf");
        if (rc) return rc;
    }
    EmbedArgs emb[2];
    for (int pass = 0; pass < (g2 ? 2 : 1); ++pass) {
        GpiTape& gg = pass ? *g2 : g;
        const float* pp = pass ? params2 : params;
        EmbedArgs a{};
        a.sf = gg.sf; a.wf = gg.wf; a.x = gg.t.x;
        a.w = w; a.w_rstride = w_rstride;
        a.params = pp; a.pstride = c->P; a.offWw = c->offWw; a.offBw = c->offBw;
        a.gstride = (long long)gg.t.cap * c->ldH;
        a.H = c->H0; a.ld = c->ldH; a.R = d.reward_dim; a.rows = rows; a.G = n_nets;
        emb[pass] = a;
    }
    if (g2) hipLaunchKernelGGL(gpi_embed_fwd_pair_kernel, dim3(stream_grid((long long)n_nets * rows * c->H0, 256), 2), dim3(256), 0, s,
                               emb[0], emb[1]);
    else hipLaunchKernelGGL(gpi_embed_fwd_kernel, dim3(stream_grid((long long)n_nets * rows * c->H0, 256)), dim3(256), 0, s, emb[0]);
    LAUNCH_CHECK("gpi_embed_fwd");
    if (g2) return mlp_forward(c->net, params + c->offNet, c->P, g.t, rows, 1, ds, s, params2 + c->offNet, &g2->t, ds2,
                               wt ? wt + c->offNet : nullptr, wt2 ? wt2 + c->offNet : nullptr);
    return mlp_forward(c->net, params + c->offNet, c->P, g.t, rows, 1, ds, s, nullptr, nullptr, nullptr,
                       wt ? wt + c->offNet : nullptr);
}

static int gpi_backward(morl_gpi_ctx* c, const float* params, GpiTape& g, const float* obs, const float* w, int w_rstride,
                        int rows, bool dropped, float* grads, hipStream_t s) {
    const morl_gpi_desc& d = c->d;
    int rc = mlp_backward(c->net, params + c->offNet, c->P, g.t, rows, 1, dropped, grads + c->offNet, true, s, c->P);
    if (rc) return rc;
    {
        EmbedBwdArgs a{};
        a.dx = g.t.dx; a.dwf = g.dwf; a.sf = g.sf; a.wf = g.wf;
        a.gstride = (long long)g.t.cap * c->ldH;
        a.H = c->H0; a.ld = c->ldH; a.rows = rows; a.G = c->nn;
        hipLaunchKernelGGL(gpi_embed_bwd_kernel, dim3(stream_grid((long long)c->nn * rows * c->H0, 256)), dim3(256), 0, s, a);
        LAUNCH_CHECK("gpi_embed_bwd");
    }
    {   // dWs = d(z_s)^T @ obs, dbs = column sums
        GemmGroupBatched grp{};
        grp.n = 1;
        grp.sC = c->P;
        GemmProblem& p = grp.p[0];
        p.A = g.t.dx; p.lda = c->ldH; grp.sA[0] = (long long)g.t.cap * c->ldH;
        p.B = obs; p.ldb = d.obs_dim; grp.sB[0] = 0; grp.b_div[0] = 1;
        p.C = grads + c->offWs; p.ldc = d.obs_dim;
        p.colsum = grads + c->offBs;
        p.M = c->H0; p.N = d.obs_dim; p.K = rows;
        p.k_per_split = round_up(rows, GEMM_BK);
        p.a_vec = vec_ok(p.A, p.lda) && (grp.sA[0] % 4 == 0);
        p.b_vec = vec_ok(p.B, p.ldb);
        p.tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM;
        p.tiles_n = (p.N + GEMM_BN - 1) / GEMM_BN;
        int tiles = p.tiles_m * p.tiles_n;
        grp.tile_start[0] = 0;
        if (use_wave_tiles((long long)tiles * c->nn)) {
            p.tiles_m = (p.M + 31) / 32;
            p.tiles_n = (p.N + 31) / 32;
            tiles = p.tiles_m * p.tiles_n;
            grp.tile_start[1] = tiles;
            if (rows > 32)
                hipLaunchKernelGGL(gemm_wave4_grouped_tn_batched_kernel, dim3(tiles, 1, c->nn), dim3(256), 0, s, grp);
            else
                hipLaunchKernelGGL(gemm_wave_grouped_tn_batched_kernel, dim3((tiles + 3) / 4, 1, c->nn), dim3(256), 0, s, grp);
        } else {
            grp.tile_start[1] = tiles;
            hipLaunchKernelGGL(gemm_grouped_tn_batched_kernel, dim3(tiles, 1, c->nn), dim3(GEMM_THREADS), 0, s, grp);
        }
        LAUNCH_CHECK("gpi_gemm_dws");
    }
    {
        EmbedGradArgs a{};
        a.dwf = g.dwf; a.w = w; a.w_rstride = w_rstride;
        a.grads = grads; a.pstride = c->P; a.offWw = c->offWw; a.offBw = c->offBw;
        a.gstride = (long long)g.t.cap * c->ldH;
        a.H = c->H0; a.ld = c->ldH; a.R = d.reward_dim; a.rows = rows; a.G = c->nn;
        hipLaunchKernelGGL(gpi_embed_grad_kernel, dim3((c->H0 + 63) / 64, c->nn), dim3(1024), 0, s, a);
        LAUNCH_CHECK("gpi_embed_grad");
    }
    return MORL_OK;
}

static int gpi_envelope_inputs(morl_gpi_ctx* c, const float* obs, const float* support, int K, int rows, hipStream_t s) {
    const int D = c->d.obs_dim, R = c->d.reward_dim;
    hipLaunchKernelGGL(gpi_expand_kernel, dim3(stream_grid((long long)rows * K * (D + R), 256)), dim3(256), 0, s, obs, D, support,
                       R, K, rows, c->obs_rep, c->w_rep);
    LAUNCH_CHECK("gpi_expand");
    return MORL_OK;
}

extern "C" int morl_gpi_update(morl_gpi_ctx* c, float* q, const float* q_target, float* exp_avg, float* exp_avg_sq,
                               const float* obs, const int32_t* actions, const float* rewards, const float* next_obs,
                               const float* dones, const float* w, int rows, const float* sampled_w, int K,
                               const uint8_t* drop_masks, const morl_gpi_cfg* cfg, const morl_gpi_out* out_in, void* stream) {
    if (!c || !cfg) return fail(MORL_ERR_ARG, "ctx / cfg is NULL");
    if (!q || !q_target || !exp_avg || !exp_avg_sq || !obs || !actions || !rewards || !next_obs || !dones || !w)
        return fail(MORL_ERR_ARG, "NULL argument");
    if (rows < 1 || rows > c->cap) return fail(MORL_ERR_STATE, "rows %d outside 1..max_rows %d", rows, c->cap);
    const bool env = cfg->gpi_pd != 0;
    if (env && (!sampled_w || K < 1 || K > c->d.max_support)) return fail(MORL_ERR_STATE, "K %d outside 1..max_support %d", K, c->d.max_support);
    if (cfg->n_per < 0 || cfg->n_per > rows) return fail(MORL_ERR_ARG, "n_per %d outside 0..rows", cfg->n_per);
    static const morl_gpi_out no_out{};
    const morl_gpi_out* out = out_in ? out_in : &no_out;
    hipStream_t s = (hipStream_t)stream;
    const morl_gpi_desc& d = c->d;
    const int R = d.reward_dim, A = d.n_actions, L = c->net.L;
    const int ldq = c->net.ld[L];
    const int64_t mb_rows = morl_gpi_mask_bytes(&d, rows), mb_env = env ? morl_gpi_mask_bytes(&d, rows * K) : 0;
    auto dropspec = [&](int phase) {
        DropSpec ds;
        ds.active = c->net.drop > 0.f;
        const int64_t off = phase == 0 ? 0 : (phase == 1 ? mb_rows : mb_rows + mb_env);
        const int64_t bytes = phase == 1 ? mb_env : mb_rows;
        ds.ext = drop_masks ? drop_masks + off : nullptr;
        ds.ext_net_bytes = bytes / c->nn;
        ds.seed = cfg->dropout_seed * 4 + (unsigned long long)phase;
        return ds;
    };
    int rc;
    // K-major shadow copies of the trunk matrices of both ensembles for this update's forward passes (one launch) -- in the
    // wave-tile regime only (see morl_ac_update): a very large batch runs the LDS-tiled engine on the original layout
    int widest = 0;
    for (int l = 1; l < c->net.L; ++l) widest = std::max(widest, c->net.dims[l]);
    const long long big_rows = env ? (long long)rows * K : rows;
    const bool use_wt = use_wave_tiles(((big_rows + 127) / 128) * ((widest + 127) / 128) * c->nn);
    const float* wq = use_wt ? c->wt_q : nullptr;
    const float* wqt = use_wt ? c->wt_qt : nullptr;
    if (use_wt) {
        TransposeMulti tm{};
        MlpLayout lay = layout_of(c->net);
        for (int l = 0; l < lay.L; ++l) lay.offW[l] += c->offNet;
        lay.P = c->P;
        tm.n = 2;
        tm.src[0] = q; tm.dst[0] = c->wt_q; tm.src[1] = q_target; tm.dst[1] = c->wt_qt;
        tm.lay[0] = tm.lay[1] = lay;
        tm.nets[0] = tm.nets[1] = c->nn;
        if ((rc = launch_transposes(tm, s))) return rc;
    }
    {
        // the target ensemble at s' and the online ensemble at s do not depend on each other: one launch per layer for both
        const DropSpec d0 = dropspec(0), d2 = dropspec(2);
        if ((rc = gpi_forward(c, q_target, c->nn, c->tt, next_obs, w, R, rows, d0, s, q, &c->tq, obs, &d2, wqt, wq)))
            return rc;
    }
    if (env) {
        if ((rc = gpi_envelope_inputs(c, next_obs, sampled_w, K, rows, s))) return rc;
        if ((rc = gpi_forward(c, q_target, c->nn, c->te, c->obs_rep, c->w_rep, R, rows * K, dropspec(1), s, nullptr, nullptr,
                              nullptr, nullptr, wqt)))
            return rc;
    }
    {
        GpiTargetArgs a{};
        a.qt = c->tt.t.out; a.gstride = (long long)c->tt.t.cap * ldq;
        a.qt_env = env ? c->te.t.out : nullptr; a.gstride_env = (long long)c->te.t.cap * ldq;
        a.ldq = ldq;
        a.w = w; a.w_rstride = R;
        a.rewards = rewards; a.dones = dones;
        a.target = c->target; a.target_env = env ? c->target_env : nullptr;
        a.rows = rows; a.A = A; a.R = R; a.nn = c->nn; a.K = K; a.gamma = cfg->gamma;
        hipLaunchKernelGGL(gpi_target_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, a);
        LAUNCH_CHECK("gpi_target");
    }
    {
        GpiLossArgs a{};
        a.q = c->tq.t.out; a.dq = c->tq.t.g[L - 1]; a.gstride = (long long)c->tq.t.cap * ldq; a.ldq = ldq;
        a.actions = actions; a.target = c->target; a.target_env = env ? c->target_env : nullptr; a.w = w;
        a.loss_out = out->critic_loss; a.td_prio = out->td_error; a.gtd_prio = out->gtd_error;
        a.rows = rows; a.A = A; a.R = R; a.nn = c->nn; a.n_per = cfg->n_per; a.delta = cfg->min_priority;
        hipLaunchKernelGGL(gpi_loss_kernel, dim3(1), dim3(256), 0, s, a);
        LAUNCH_CHECK("gpi_loss");
    }
    if ((rc = gpi_backward(c, q, c->tq, obs, w, R, rows, c->net.drop > 0.f, c->grads, s))) return rc;
    if (cfg->max_grad_norm >= 0.f) {
        hipLaunchKernelGGL(gpi_clip_kernel, dim3(c->nn), dim3(1024), 0, s, c->grads, (long long)c->P, cfg->max_grad_norm, out->grad_norm);
        LAUNCH_CHECK("gpi_clip");
    }
    const size_t rb = (size_t)rows * R * sizeof(float);
    if (out->target_q) HIP_TRY(hipMemcpyAsync(out->target_q, c->target, rb, hipMemcpyDeviceToDevice, s));
    if (out->target_q_envelope && env) HIP_TRY(hipMemcpyAsync(out->target_q_envelope, c->target_env, rb, hipMemcpyDeviceToDevice, s));
    if (out->grads) HIP_TRY(hipMemcpyAsync(out->grads, c->grads, (size_t)c->nn * c->P * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (cfg->apply_step) {
        morl_ac_cfg ac{};
        ac.beta1 = cfg->beta1; ac.beta2 = cfg->beta2; ac.eps = cfg->eps;
        if ((rc = adam(q, c->grads, exp_avg, exp_avg_sq, (long long)c->nn * c->P, 1, cfg->lr, nullptr, cfg->adam_step, &ac, s))) return rc;
    }
    return MORL_OK;
}

extern "C" int morl_gpi_q_forward(morl_gpi_ctx* c, const float* params, int n_nets, const float* obs, const float* w,
                                  int w_per_row, int rows, float* q_out, void* stream) {
    if (!c || !params || !obs || !w || !q_out) return fail(MORL_ERR_ARG, "NULL argument");
    if (n_nets < 1 || n_nets > c->nn) return fail(MORL_ERR_ARG, "n_nets %d outside 1..%d", n_nets, c->nn);
    if (rows < 1 || rows > c->cap_env) return fail(MORL_ERR_STATE, "rows %d outside 1..%d", rows, c->cap_env);
    hipStream_t s = (hipStream_t)stream;
    const int R = c->d.reward_dim, AR = c->d.n_actions * R, ldq = c->net.ld[c->net.L];
    int rc = gpi_forward(c, params, n_nets, c->te, obs, w, w_per_row ? R : 0, rows, DropSpec(), s);
    if (rc) return rc;
    ConcatArgs a{};
    a.src[0] = c->te.t.out; a.width[0] = AR; a.gstride[0] = (long long)c->te.t.cap * ldq; a.rstride[0] = ldq;
    a.n_src = 1;
    a.dst = q_out; a.ld = AR; a.dst_gstride = (long long)rows * AR; a.rows = rows; a.G = n_nets;
    hipLaunchKernelGGL(ac_concat_kernel, dim3(stream_grid((long long)n_nets * rows * AR, 256)), dim3(256), 0, s, a);
    LAUNCH_CHECK("gpi_q_compact");
    return MORL_OK;
}

extern "C" int morl_gpi_action(morl_gpi_ctx* c, const float* q, const float* obs, const float* support, int M, const float* w,
                               int32_t* action_out, int32_t* policy_out, void* stream) {
    if (!c || !q || !obs || !w || !action_out) return fail(MORL_ERR_ARG, "NULL argument");
    if (M < 0 || M > c->cap_env || (M > 0 && !support)) return fail(MORL_ERR_STATE, "support size %d outside 0..%d", M, c->cap_env);
    hipStream_t s = (hipStream_t)stream;
    const int R = c->d.reward_dim, A = c->d.n_actions, D = c->d.obs_dim, ldq = c->net.ld[c->net.L];
    int rc;
    if (M > 0) {
        // rows = the M support weights, same observation: expand with "K = M, one row"
        if ((rc = gpi_envelope_inputs(c, obs, support, M, 1, s))) return rc;
        if ((rc = gpi_forward(c, q, 1, c->te, c->obs_rep, c->w_rep, R, M, DropSpec(), s))) return rc;
        hipLaunchKernelGGL(gpi_action_kernel, dim3(1), dim3(64), 0, s, (const float*)c->te.t.out, ldq, M, A, R, w, action_out, policy_out);
    } else {
        (void)D;
        if ((rc = gpi_forward(c, q, c->nn, c->te, obs, w, 0, 1, DropSpec(), s))) return rc;
        hipLaunchKernelGGL(gpi_min_nets_kernel, dim3((A * R + 255) / 256), dim3(256), 0, s, (const float*)c->te.t.out,
                           (long long)c->te.t.cap * ldq, c->nn, A * R, c->qmin);
        LAUNCH_CHECK("gpi_min_nets");
        hipLaunchKernelGGL(gpi_action_kernel, dim3(1), dim3(64), 0, s, (const float*)c->qmin, ldq, 1, A, R, w, action_out, policy_out);
    }
    LAUNCH_CHECK("gpi_action");
    return MORL_OK;
}

extern "C" int morl_gpi_actions(morl_gpi_ctx* c, const float* q, const float* obs, int n, const float* support, int M,
                                const float* w, int32_t* actions_out, void* stream) {
    if (!c || !q || !obs || !support || !w || !actions_out) return fail(MORL_ERR_ARG, "NULL argument");
    if (n < 1 || M < 1 || (long long)n * M > c->cap_env)
        return fail(MORL_ERR_STATE, "n * M = %lld exceeds max_rows * max_support = %d", (long long)n * M, c->cap_env);
    hipStream_t s = (hipStream_t)stream;
    const int R = c->d.reward_dim, A = c->d.n_actions, ldq = c->net.ld[c->net.L];
    int rc;
    if ((rc = gpi_envelope_inputs(c, obs, support, M, n, s))) return rc;
    if ((rc = gpi_forward(c, q, 1, c->te, c->obs_rep, c->w_rep, R, n * M, DropSpec(), s))) return rc;
    hipLaunchKernelGGL(gpi_actions_kernel, dim3((n + 3) / 4), dim3(256), 0, s, (const float*)c->te.t.out, ldq, n, M, A, R, w,
                       0, actions_out);
    LAUNCH_CHECK("gpi_actions");
    return MORL_OK;
}

// evaluation in lock-step over many (observation, weight) pairs: row i acts under its own w_rows[i]
extern "C" int morl_gpi_actions_rows(morl_gpi_ctx* c, const float* q, const float* obs, const float* w_rows, int n,
                                     const float* support, int M, int32_t* actions_out, void* stream) {
    if (!c || !q || !obs || !w_rows || !actions_out) return fail(MORL_ERR_ARG, "NULL argument");
    if (M < 0 || (M > 0 && !support)) return fail(MORL_ERR_ARG, "support size %d without a support set", M);
    if (n < 1 || (long long)n * std::max(M, 1) > c->cap_env)
        return fail(MORL_ERR_STATE, "n * max(M, 1) = %lld exceeds max_rows * max_support = %d", (long long)n * std::max(M, 1),
                    c->cap_env);
    hipStream_t s = (hipStream_t)stream;
    const int R = c->d.reward_dim, A = c->d.n_actions, ldq = c->net.ld[c->net.L];
    int rc;
    if (M > 0) {
        // gpi_action (gpi_pd.py:564-582) per row: net 0 on (obs_i, support_k), scalarised with the row's weight
        if ((rc = gpi_envelope_inputs(c, obs, support, M, n, s))) return rc;
        if ((rc = gpi_forward(c, q, 1, c->te, c->obs_rep, c->w_rep, R, n * M, DropSpec(), s))) return rc;
    } else {
        // max_action (gpi_pd.py:608-617) per row: element-wise min over the ensemble at (obs_i, w_i), in place in net 0's rows
        if ((rc = gpi_forward(c, q, c->nn, c->te, obs, w_rows, R, n, DropSpec(), s))) return rc;
        if (c->nn > 1) {
            const int e = n * ldq;
            hipLaunchKernelGGL(gpi_min_nets_kernel, dim3((e + 255) / 256), dim3(256), 0, s, (const float*)c->te.t.out,
                               (long long)c->te.t.cap * ldq, c->nn, e, c->te.t.out);
            LAUNCH_CHECK("gpi_min_nets_rows");
        }
    }
    hipLaunchKernelGGL(gpi_actions_kernel, dim3((n + 3) / 4), dim3(256), 0, s, (const float*)c->te.t.out, ldq, n, std::max(M, 1),
                       A, R, w_rows, R, actions_out);
    LAUNCH_CHECK("gpi_actions_rows");
    return MORL_OK;
}

extern "C" int morl_gpi_priorities(morl_gpi_ctx* c, const float* q, const float* q_target, const float* obs,
                                   const int32_t* actions, const float* rewards, const float* next_obs, const float* dones,
                                   int rows, const float* w, const float* support, int M, int gpi_pd, float gamma,
                                   float* gtd_out, void* stream) {
    if (!c || !q || !q_target || !obs || !actions || !rewards || !next_obs || !dones || !w || !gtd_out)
        return fail(MORL_ERR_ARG, "NULL argument");
    if (rows < 1 || rows > c->cap) return fail(MORL_ERR_STATE, "rows %d outside 1..max_rows %d", rows, c->cap);
    if (gpi_pd && (!support || M < 1 || (long long)rows * M > c->cap_env))
        return fail(MORL_ERR_STATE, "rows * M = %lld exceeds max_rows * max_support = %d", (long long)rows * M, c->cap_env);
    hipStream_t s = (hipStream_t)stream;
    const int R = c->d.reward_dim, A = c->d.n_actions, ldq = c->net.ld[c->net.L];
    int rc;
    GpiTargetArgs a{};
    a.ldq = ldq; a.w = w; a.w_rstride = 0; a.rewards = rewards; a.dones = dones;
    a.rows = rows; a.A = A; a.R = R; a.gamma = gamma;
    if (gpi_pd) {
        if ((rc = gpi_envelope_inputs(c, next_obs, support, M, rows, s))) return rc;
        if ((rc = gpi_forward(c, q_target, c->nn, c->te, c->obs_rep, c->w_rep, R, rows * M, DropSpec(), s))) return rc;
        a.qt_env = c->te.t.out; a.gstride_env = (long long)c->te.t.cap * ldq;
        a.target_env = c->target_env; a.nn = c->nn; a.K = M;
        hipLaunchKernelGGL(gpi_target_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, a);
    } else {
        // gpi_pd.py:641-649: greedy action of ONLINE net 0 under w, value from TARGET net 0 -- as the envelope kernel with the
        // "ensemble" = {online 0} would pick the wrong values, run the two single nets and let a 1-net target pick from qt
        if ((rc = gpi_forward(c, q, 1, c->tt, next_obs, w, 0, rows, DropSpec(), s))) return rc;          // online: arg-max source
        if ((rc = gpi_forward(c, q_target, 1, c->te, next_obs, w, 0, rows, DropSpec(), s))) return rc;   // target: values
        hipLaunchKernelGGL(gpi_ddqn_target_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, (const float*)c->tt.t.out,
                           (const float*)c->te.t.out, ldq, w, rewards, dones, rows, A, R, gamma, c->target_env);
    }
    LAUNCH_CHECK("gpi_prio_target");
    if ((rc = gpi_forward(c, q, 1, c->tq, obs, w, 0, rows, DropSpec(), s))) return rc;
    hipLaunchKernelGGL(gpi_gtd_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, (const float*)c->tq.t.out, ldq, actions,
                       (const float*)c->target_env, w, 0, rows, R, gtd_out);
    LAUNCH_CHECK("gpi_gtd");
    return MORL_OK;
}

// =====================================================================================================================
// Probabilistic dynamics ensemble (include/morl_hip.h, "Probabilistic dynamics ensemble of the Dyna part of GPI-PD")
// =====================================================================================================================
#include "ens_kernels.h"

struct morl_ens_ctx {
    morl_ens_desc d{};
    Mlp net;                   // dims [in, hidden..., 2 * out]; members are the batch axis of every launch
    int E = 0, cap = 0;
    Tape t;
    float* grads = nullptr;    // [E][Pm]
    double* part = nullptr;    // [n_blocks][1 + 2 * out]
    int max_blocks = 0;
    std::vector<void*> allocs;
};

static int ens_fill(const morl_ens_desc* d, morl_ens_ctx& c) {
    if (!d) return fail(MORL_ERR_ARG, "desc is NULL");
    if (d->n_hidden < 1 || d->n_hidden > MORL_MAX_LAYERS - 1) return fail(MORL_ERR_ARG, "n_hidden %d out of range", d->n_hidden);
    if (d->input_dim < 1 || d->output_dim < 1 || d->output_dim > ENS_MAX_OUT) return fail(MORL_ERR_ARG, "bad dims in=%d out=%d", d->input_dim, d->output_dim);
    if (d->ensemble_size < 1 || d->ensemble_size > 64) return fail(MORL_ERR_ARG, "ensemble_size %d", d->ensemble_size);
    c.d = *d;
    c.E = d->ensemble_size;
    c.net = Mlp();
    c.net.L = d->n_hidden + 1;
    c.net.dims[0] = d->input_dim;
    for (int l = 0; l < d->n_hidden; ++l) {
        if (d->hidden[l] < 1) return fail(MORL_ERR_ARG, "hidden[%d] = %d", l, d->hidden[l]);
        c.net.dims[l + 1] = d->hidden[l];
    }
    c.net.dims[c.net.L] = 2 * d->output_dim;
    c.net.finish();
    return MORL_OK;
}

extern "C" int64_t morl_ens_param_count(const morl_ens_desc* d) {
    morl_ens_ctx c;
    return ens_fill(d, c) ? -1 : c.net.P;
}

extern "C" int morl_ens_destroy(morl_ens_ctx* c) {
    if (!c) return MORL_OK;
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    return MORL_OK;
}

extern "C" int morl_ens_create(morl_ens_ctx** out, const morl_ens_desc* d) {
    if (!out) return fail(MORL_ERR_ARG, "out is NULL");
    *out = nullptr;
    morl_ens_ctx* c = new (std::nothrow) morl_ens_ctx();
    if (!c) return fail(MORL_ERR_ALLOC, "out of host memory");
    int rc = ens_fill(d, *c);
    if (rc) { delete c; return rc; }
    if (d->max_rows < 1) { delete c; return fail(MORL_ERR_ARG, "max_rows %d", d->max_rows); }
    c->cap = d->max_rows;
    c->max_blocks = (int)(((long long)c->E * c->cap + 3) / 4);
    float* part = nullptr;
    if ((rc = alloc_tape(c->allocs, c->net, c->t, c->E, c->E, c->cap, false)) ||
        (rc = alloc_f(c->allocs, &c->grads, (size_t)c->E * c->net.P)) ||
        (rc = alloc_f(c->allocs, &part, (size_t)c->max_blocks * (1 + 2 * d->output_dim) * 2))) {
        morl_ens_destroy(c);
        return rc;
    }
    c->part = reinterpret_cast<double*>(part);
    if (hipDeviceSynchronize() != hipSuccess) { morl_ens_destroy(c); return fail(MORL_ERR_HIP, "workspace init failed"); }
    *out = c;
    return MORL_OK;
}

static int ens_forward_core(morl_ens_ctx* c, const float* params, const float* mu, const float* sigma, const float* x,
                            int x_per_member, int rows, hipStream_t s) {
    if (rows < 1 || rows > c->cap) return fail(MORL_ERR_STATE, "rows %d outside 1..max_rows %d", rows, c->cap);
    if ((mu == nullptr) != (sigma == nullptr)) return fail(MORL_ERR_ARG, "mu and sigma go together");
    c->t.G = c->E;
    EnsNormArgs a{};
    a.x = x; a.x_gstride = x_per_member ? (long long)rows * c->d.input_dim : 0;
    a.mu = mu; a.sigma = sigma;
    a.dst = c->t.x; a.dst_gstride = (long long)c->cap * c->net.ld[0];
    a.in_dim = c->d.input_dim; a.ld = c->net.ld[0]; a.rows = rows; a.E = c->E;
    hipLaunchKernelGGL(ens_norm_kernel, dim3(stream_grid((long long)c->E * rows * c->net.ld[0], 256)), dim3(256), 0, s, a);
    LAUNCH_CHECK("ens_norm");
    return mlp_forward(c->net, params, c->net.P, c->t, rows, 1, DropSpec(), s);
}

extern "C" int morl_ens_train_step(morl_ens_ctx* c, float* params, float* exp_avg, float* exp_avg_sq, float* logvar_bounds,
                                   float* lv_m, float* lv_v, const float* mu, const float* sigma, const float* x,
                                   const float* y, int rows, const morl_ens_cfg* cfg, float* loss_out, void* stream) {
    if (!c || !params || !exp_avg || !exp_avg_sq || !logvar_bounds || !lv_m || !lv_v || !x || !y || !cfg)
        return fail(MORL_ERR_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = ens_forward_core(c, params, mu, sigma, x, 1, rows, s);
    if (rc) return rc;
    const Mlp& m = c->net;
    const int O = c->d.output_dim;
    const int n_blocks = (int)(((long long)c->E * rows + 3) / 4);
    const float inv_count = 1.0f / ((float)c->E * (float)rows * (float)O);
    {
        EnsNllArgs a{};
        a.head = c->t.out; a.dhead = c->t.g[m.L - 1]; a.gstride = (long long)c->cap * m.ld[m.L]; a.ld = m.ld[m.L];
        a.y = y; a.bounds = logvar_bounds; a.part = c->part;
        a.rows = rows; a.O = O; a.E = c->E; a.inv_count = inv_count;
        hipLaunchKernelGGL(ens_nll_kernel, dim3(n_blocks), dim3(256), 0, s, a);
        LAUNCH_CHECK("ens_nll");
    }
    if ((rc = mlp_backward(m, params, m.P, c->t, rows, 1, false, c->grads, false, s))) return rc;
    const double b1 = cfg->beta1, b2 = cfg->beta2;
    const int t = std::max(1, cfg->adam_step);
    const float neg_step = (float)(-(cfg->lr / (1.0 - std::pow(b1, (double)t))));
    const float bc2_sqrt = (float)std::sqrt(1.0 - std::pow(b2, (double)t));
    {
        EnsAdamArgs a{};
        a.params = params; a.grads = c->grads; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq;
        a.Pm = m.P; a.n_layers = m.L; a.total = (long long)c->E * m.P;
        for (int l = 0; l < m.L; ++l) {
            a.layer_end[l] = m.offB[l] + m.dims[l + 1];
            a.wd[l] = cfg->weight_decay[l];
        }
        a.neg_step_size = neg_step; a.bc2_sqrt = bc2_sqrt;
        a.one_minus_b1 = (float)(1.0 - b1); a.b2 = (float)b2; a.one_minus_b2 = (float)(1.0 - b2); a.eps = (float)cfg->eps;
        hipLaunchKernelGGL(ens_adam_kernel, dim3(stream_grid(a.total, 256)), dim3(256), 0, s, a);
        LAUNCH_CHECK("ens_adam");
    }
    hipLaunchKernelGGL(ens_bounds_step_kernel, dim3(1), dim3(256), 0, s, (const double*)c->part, n_blocks, O, inv_count,
                       logvar_bounds, lv_m, lv_v, neg_step, bc2_sqrt, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2),
                       (float)cfg->eps, loss_out);
    LAUNCH_CHECK("ens_bounds_step");
    return MORL_OK;
}

extern "C" int morl_ens_forward(morl_ens_ctx* c, const float* params, const float* logvar_bounds, const float* mu,
                                const float* sigma, const float* x, int x_per_member, int rows, float* mean_out,
                                float* logvar_out, void* stream) {
    if (!c || !params || !x || !mean_out || (logvar_out && !logvar_bounds)) return fail(MORL_ERR_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = ens_forward_core(c, params, mu, sigma, x, x_per_member, rows, s);
    if (rc) return rc;
    EnsOutArgs a{};
    a.head = c->t.out; a.gstride = (long long)c->cap * c->net.ld[c->net.L]; a.ld = c->net.ld[c->net.L];
    a.bounds = logvar_bounds; a.mean = mean_out; a.logvar = logvar_out;
    a.rows = rows; a.O = c->d.output_dim; a.E = c->E;
    hipLaunchKernelGGL(ens_out_kernel, dim3(stream_grid((long long)c->E * rows * a.O, 256)), dim3(256), 0, s, a);
    LAUNCH_CHECK("ens_out");
    return MORL_OK;
}

extern "C" int morl_ens_mse(morl_ens_ctx* c, const float* params, const float* logvar_bounds, const float* mu,
                            const float* sigma, const float* x, const float* y, int rows, float* mse_out, void* stream) {
    (void)logvar_bounds;
    if (!c || !params || !x || !y || !mse_out) return fail(MORL_ERR_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = ens_forward_core(c, params, mu, sigma, x, 0, rows, s);
    if (rc) return rc;
    hipLaunchKernelGGL(ens_mse_kernel, dim3(c->E), dim3(256), 0, s, (const float*)c->t.out,
                       (long long)c->cap * c->net.ld[c->net.L], c->net.ld[c->net.L], y, rows, c->d.output_dim, mse_out);
    LAUNCH_CHECK("ens_mse");
    return MORL_OK;
}


// ---- the gradient_updates loop of the reference inside one library entry (include/morl_hip.h) ------------------------------
extern "C" int morl_ac_update_n(morl_ac_ctx* c, const morl_ac_state* st, int n, const morl_ac_batch* batches,
                                const morl_ac_cfg* cfgs, const morl_ac_out* outs, void* stream) {
    if (!c || !st || !batches || !cfgs) return fail(MORL_ERR_ARG, "NULL argument");
    if (n < 1) return fail(MORL_ERR_ARG, "n = %d updates", n);
    for (int k = 0; k < n; ++k) {
        const int rc = morl_ac_update(c, st, batches + k, cfgs + k, outs ? outs + k : nullptr, stream);
        if (rc) {
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", morl_last_error());
            return fail(rc, "update %d of %d: %s", k, n, msg);
        }
    }
    return MORL_OK;
}

// The same loop with prioritised replay for the TD3-style learner of GPI-PD with continuous actions
// (gpi_pd_continuous_action.py:373-417): see morl_gpi_update_n_per.
extern "C" int morl_ac_update_n_per(morl_ac_ctx* c, const morl_ac_state* st, int n, const morl_gpi_per* per, const morl_ac_batch* batches,
                                    const morl_ac_cfg* cfgs, const morl_ac_out* outs, void* stream) {
    if (!c || !st || !per || !batches || !cfgs || !outs) return fail(MORL_ERR_ARG, "NULL argument");
    if (n < 1) return fail(MORL_ERR_ARG, "n = %d updates", n);
    if (!per->tree || !per->running_max || !per->u01 || !per->records || !per->idx) return fail(MORL_ERR_ARG, "NULL field of morl_gpi_per");
    const int B = per->B;
    if (B < 1) return fail(MORL_ERR_ARG, "B = %d", B);
    // (refused HERE, before the first launch: morl_sumtree_update_clamped would only notice after iteration 0's gather and
    // optimiser step were enqueued, and leave the caller with a half-taken loop)
    if (B > morl_host::TREE_UPDATE_MAX) return fail(MORL_ERR_ARG, "prioritised loop: B=%d > %d entries of one tree-update launch (run one sample / update / "
                                  "update_priorities round per iteration instead)", B, morl_host::TREE_UPDATE_MAX);
    const int copies = per->doubled ? 2 : 1;
    for (int k = 0; k < n; ++k) {
        const morl_ac_batch& b = batches[k];
        if (b.rows != copies * B) return fail(MORL_ERR_ARG, "update %d: %d rows for %d sampled transitions x %d", k, b.rows, B, copies);
        if (b.active > 1) return fail(MORL_ERR_ARG, "update %d: one learner per prioritised loop", k);
        if (cfgs[k].n_per != B) return fail(MORL_ERR_ARG, "update %d: n_per %d != B %d", k, cfgs[k].n_per, B);
        if (!outs[k].priority) return fail(MORL_ERR_ARG, "update %d: the priority output is needed", k);
        int64_t* idx = per->idx + (size_t)k * B;
        int rc = MORL_OK;
        for (int cp = 0; cp < copies && !rc; ++cp)
            rc = morl_sample_gather(cp == 0 ? per->tree : nullptr, per->n_levels, cp == 0 ? per->u01 + (size_t)k * B : nullptr,
                                    cp == 0 ? nullptr : idx, per->records, per->record_floats, per->capacity, B, per->D, per->R,
                                    per->action_dim, (float*)b.obs + (size_t)cp * B * per->D, (float*)b.next_obs + (size_t)cp * B * per->D,
                                    (float*)b.rewards + (size_t)cp * B * per->R, (float*)b.dones + (size_t)cp * B,
                                    (float*)b.actions + (size_t)cp * B * per->action_dim, nullptr, cp == 0 ? idx : nullptr, nullptr,
                                    nullptr, 0, stream);
        if (!rc) rc = morl_ac_update(c, st, batches + k, cfgs + k, outs + k, stream);
        if (!rc) rc = morl_sumtree_update_clamped(per->tree, per->n_levels, idx, outs[k].priority, B, per->alpha, per->min_priority,
                                                  per->running_max, nullptr, stream);
        if (rc) {
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", morl_last_error());
            return fail(rc, "update %d of %d: %s", k, n, msg);
        }
    }
    return MORL_OK;
}

extern "C" int morl_gpi_update_n(morl_gpi_ctx* c, float* q, const float* q_target, float* exp_avg, float* exp_avg_sq, int n,
                                 const morl_gpi_batch* batches, const morl_gpi_cfg* cfgs, const morl_gpi_out* outs, void* stream) {
    if (!c || !batches || !cfgs) return fail(MORL_ERR_ARG, "NULL argument");
    if (n < 1) return fail(MORL_ERR_ARG, "n = %d updates", n);
    for (int k = 0; k < n; ++k) {
        const morl_gpi_batch& b = batches[k];
        const int rc = morl_gpi_update(c, q, q_target, exp_avg, exp_avg_sq, b.obs, b.actions, b.rewards, b.next_obs, b.dones, b.w,
                                       b.rows, b.sampled_w, b.K, b.drop_masks, cfgs + k, outs ? outs + k : nullptr, stream);
        if (rc) {
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", morl_last_error());
            return fail(rc, "update %d of %d: %s", k, n, msg);
        }
    }
    return MORL_OK;
}

// GPIPD.update's loop WITH prioritised replay (the reference's default): every iteration samples through the tree the iteration
// before it wrote.  Per iteration: descent + gather (morl_sample_gather) -> morl_gpi_update -> priorities = max(|td|, min) ** alpha
// -> tree (morl_sumtree_update_clamped), all enqueued here.
extern "C" int morl_gpi_update_n_per(morl_gpi_ctx* c, float* q, const float* q_target, float* exp_avg, float* exp_avg_sq, int n,
                                     const morl_gpi_per* per, const morl_gpi_batch* batches, const morl_gpi_cfg* cfgs,
                                     const morl_gpi_out* outs, void* stream) {
    if (!c || !per || !batches || !cfgs || !outs) return fail(MORL_ERR_ARG, "NULL argument");
    if (n < 1) return fail(MORL_ERR_ARG, "n = %d updates", n);
    if (!per->tree || !per->running_max || !per->u01 || !per->records || !per->idx) return fail(MORL_ERR_ARG, "NULL field of morl_gpi_per");
    const int B = per->B;
    if (B < 1) return fail(MORL_ERR_ARG, "B = %d", B);
    // (refused HERE, before the first launch: morl_sumtree_update_clamped would only notice after iteration 0's gather and
    // optimiser step were enqueued, and leave the caller with a half-taken loop)
    if (B > morl_host::TREE_UPDATE_MAX) return fail(MORL_ERR_ARG, "prioritised loop: B=%d > %d entries of one tree-update launch (run one sample / update / "
                                  "update_priorities round per iteration instead)", B, morl_host::TREE_UPDATE_MAX);
    const int copies = per->doubled ? 2 : 1;
    for (int k = 0; k < n; ++k) {
        const morl_gpi_batch& b = batches[k];
        char msg[400];
        if (b.rows != copies * B) return fail(MORL_ERR_ARG, "update %d: %d rows for %d sampled transitions x %d", k, b.rows, B, copies);
        if (cfgs[k].n_per != B) return fail(MORL_ERR_ARG, "update %d: n_per %d != B %d", k, cfgs[k].n_per, B);
        const float* raw = per->use_gtd ? outs[k].gtd_error : outs[k].td_error;
        if (!raw) return fail(MORL_ERR_ARG, "update %d: the %s output is needed for the priorities", k, per->use_gtd ? "gtd_error" : "td_error");
        int64_t* idx = per->idx + (size_t)k * B;
        int rc = MORL_OK;
        for (int cp = 0; cp < copies && !rc; ++cp)      // (the second copy re-reads the indices the first one wrote)
            rc = morl_sample_gather(cp == 0 ? per->tree : nullptr, per->n_levels, cp == 0 ? per->u01 + (size_t)k * B : nullptr,
                                    cp == 0 ? nullptr : idx, per->records, per->record_floats, per->capacity, B, per->D, per->R,
                                    per->action_dim, (float*)b.obs + (size_t)cp * B * per->D, (float*)b.next_obs + (size_t)cp * B * per->D,
                                    (float*)b.rewards + (size_t)cp * B * per->R, (float*)b.dones + (size_t)cp * B, nullptr,
                                    (int32_t*)b.actions + (size_t)cp * B * per->action_dim, cp == 0 ? idx : nullptr, nullptr, nullptr, 0, stream);
        if (!rc) rc = morl_gpi_update(c, q, q_target, exp_avg, exp_avg_sq, b.obs, b.actions, b.rewards, b.next_obs, b.dones, b.w,
                                      b.rows, b.sampled_w, b.K, b.drop_masks, cfgs + k, outs + k, stream);
        if (!rc) rc = morl_sumtree_update_clamped(per->tree, per->n_levels, idx, raw, B, per->alpha, per->min_priority,
                                                  per->running_max, nullptr, stream);
        if (rc) {
            snprintf(msg, sizeof(msg), "%s", morl_last_error());
            return fail(rc, "update %d of %d: %s", k, n, msg);
        }
    }
    return MORL_OK;
}
