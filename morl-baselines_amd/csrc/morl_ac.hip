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
    float* x = nullptr;                          // [xG][cap][ld0]
    float* h[MORL_MAX_LAYERS] = {};              // post-ReLU activations of the hidden layers
    float* zx[MORL_MAX_LAYERS] = {};             // Linear output -> xhat (LayerNorm / Dropout nets only)
    float* rstd[MORL_MAX_LAYERS] = {};
    uint8_t* mask[MORL_MAX_LAYERS] = {};
    float* out = nullptr;                        // [G][cap][ld[L]]
    float* g[MORL_MAX_LAYERS] = {};              // dLoss/dz of every linear layer
    float* dx = nullptr;                         // [G][cap][ld0]
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
    float* save_y = nullptr;                     // [PG][cap][Ad]
    float* save_std = nullptr;
    float* gq = nullptr;                         // [QG][Pq]
    float* gp = nullptr;                         // [PG][Pp]
    float* alpha_dev = nullptr;                  // [PG]
    std::vector<void*> allocs;
};

static int alloc_f(morl_ac_ctx* c, float** p, size_t n) {
    int rc = dmalloc((void**)p, std::max<size_t>(n, 4) * sizeof(float));
    if (rc) return rc;
    c->allocs.push_back(*p);
    return hipMemsetAsync(*p, 0, std::max<size_t>(n, 4) * sizeof(float), nullptr) == hipSuccess
               ? MORL_OK : fail(MORL_ERR_HIP, "hipMemsetAsync failed");
}

static int alloc_tape(morl_ac_ctx* c, const Mlp& m, Tape& t, int G, int xG, bool post) {
    int rc;
    const size_t cap = (size_t)c->cap;
    t.G = G; t.xG = xG;
    if ((rc = alloc_f(c, &t.x, (size_t)xG * cap * m.ld[0]))) return rc;
    if ((rc = alloc_f(c, &t.dx, (size_t)G * cap * m.ld[0]))) return rc;
    for (int l = 0; l < m.L; ++l) {
        const size_t n = (size_t)G * cap * m.ld[l + 1];
        if ((rc = alloc_f(c, &t.g[l], n))) return rc;
        if (l == m.L - 1) { if ((rc = alloc_f(c, &t.out, n))) return rc; }
        else {
            if ((rc = alloc_f(c, &t.h[l], n))) return rc;
            if (post) {
                if ((rc = alloc_f(c, &t.zx[l], n))) return rc;
                if ((rc = alloc_f(c, &t.rstd[l], (size_t)G * cap))) return rc;
                float* mk = nullptr;
                if ((rc = alloc_f(c, &mk, ((size_t)G * cap * m.dims[l + 1] + 3) / 4))) return rc;
                t.mask[l] = reinterpret_cast<uint8_t*>(mk);
            }
        }
    }
    return MORL_OK;
}

static int fill_nets(const morl_ac_desc* d, Mlp& q, Mlp& pol, int& heads, bool& w_input) {
    if (!d) return fail(MORL_ERR_ARG, "desc is NULL");
    if (d->algo < MORL_AC_CAPQL || d->algo > MORL_AC_TD3) return fail(MORL_ERR_ARG, "unknown algo %d", d->algo);
    if (d->n_hidden < 1 || d->n_hidden > MORL_MAX_LAYERS - 1) return fail(MORL_ERR_ARG, "n_hidden %d out of range", d->n_hidden);
    if (d->obs_dim < 1 || d->act_dim < 1 || d->reward_dim < 1 || d->reward_dim > MORL_MAX_OBJ)
        return fail(MORL_ERR_ARG, "bad dims D=%d Ad=%d R=%d", d->obs_dim, d->act_dim, d->reward_dim);
    if (d->num_q < 1 || d->num_q > 4) return fail(MORL_ERR_ARG, "num_q %d not in 1..4", d->num_q);
    if (d->algo == MORL_AC_MOSAC && d->num_q != 2) return fail(MORL_ERR_ARG, "MOSAC uses exactly two critics");
    if (d->q_drop_rate < 0.f || d->q_drop_rate >= 1.f) return fail(MORL_ERR_ARG, "drop rate %g", (double)d->q_drop_rate);
    w_input = d->algo != MORL_AC_MOSAC;
    heads = d->algo == MORL_AC_TD3 ? 1 : 2;
    q = Mlp();
    pol = Mlp();
    q.L = pol.L = d->n_hidden + 1;
    q.dims[0] = d->obs_dim + d->act_dim + (w_input ? d->reward_dim : 0);
    pol.dims[0] = d->obs_dim + (w_input ? d->reward_dim : 0);
    for (int l = 0; l < d->n_hidden; ++l) {
        if (d->hidden[l] < 1 || d->hidden[l] > 64 * POST_MAXJ) return fail(MORL_ERR_ARG, "hidden[%d] = %d", l, d->hidden[l]);
        q.dims[l + 1] = pol.dims[l + 1] = d->hidden[l];
    }
    q.dims[q.L] = d->reward_dim;
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
    if ((rc = alloc_tape(c, c->q, c->tq_a, c->QG, c->PG, post)) || (rc = alloc_tape(c, c->q, c->tq_b, c->QG, c->PG, post)) ||
        (rc = alloc_tape(c, c->pol, c->tp_a, c->PG, c->PG, false)) || (rc = alloc_tape(c, c->pol, c->tp_b, c->PG, c->PG, false)) ||
        (rc = alloc_f(c, &c->act, c->PG * cap * Ad)) || (rc = alloc_f(c, &c->logp_next, c->PG * cap)) ||
        (rc = alloc_f(c, &c->logp_pi, c->PG * cap)) || (rc = alloc_f(c, &c->save_y, c->PG * cap * Ad)) ||
        (rc = alloc_f(c, &c->save_std, c->PG * cap * Ad)) || (rc = alloc_f(c, &c->gq, (size_t)c->QG * q.P)) ||
        (rc = alloc_f(c, &c->gp, (size_t)c->PG * p.P)) || (rc = alloc_f(c, &c->alpha_dev, c->PG))) {
        morl_ac_destroy(c);
        return rc;
    }
    if (hipDeviceSynchronize() != hipSuccess) { morl_ac_destroy(c); return fail(MORL_ERR_HIP, "workspace init failed"); }
    *out = c;
    return MORL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// batched layer launches
// ---------------------------------------------------------------------------------------------------------------------
// Engine choice per launch: the LDS-tiled 128 x 128 engine needs >= ~half a chip of tiles to be worth its barriers;
// below that the work is latency-bound and the wave-level 32 x 32 tiles (gemm_wave.h) spread it over 16x more waves.
// Both accumulate in the same order -> identical bits, so the choice never shows in the results.
static int g_ac_gemm_mode = 0;      // 0 auto, 1 always LDS tiles, 2 always wave tiles (morl_ac_set_gemm_mode)
static bool use_wave_tiles(long long tiles128) {
    if (g_ac_gemm_mode == 1) return false;
    if (g_ac_gemm_mode == 2) return true;
    return tiles128 < 128;
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

struct DropSpec {
    bool active = false;              // train-mode dropout
    const uint8_t* ext = nullptr;     // explicit masks of this phase: [G][per_net bytes]
    int64_t ext_net_bytes = 0;
    unsigned long long seed = 0;
};

// forward of G nets (params + g * pstride) on t.x (shared by groups of x_div nets); result in t.out
static int mlp_forward(morl_ac_ctx* c, const Mlp& m, const float* params, int64_t pstride, Tape& t, int rows, int x_div,
                       const DropSpec& ds, hipStream_t s) {
    const long long cap = c->cap;
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
        g.B = params + m.offW[l];
        g.ldb = m.dims[l];
        b.sB = pstride;
        g.bias = params + m.offB[l];
        b.sBias = pstride;
        g.C = last ? t.out : (post ? t.zx[l] : t.h[l]);
        g.ldc = m.ld[l + 1];
        b.sC = cap * m.ld[l + 1];
        g.M = rows; g.N = m.dims[l + 1]; g.K = m.dims[l];
        int rc = (last || post) ? launch_bgemm<true, true, EPI_BIAS>(b, t.G, s, "ac_gemm_fwd")
                                : launch_bgemm<true, true, EPI_BIAS_RELU>(b, t.G, s, "ac_gemm_fwd_relu");
        if (rc) return rc;
        if (post) {
            PostArgs a{};
            a.z = t.zx[l]; a.h = t.h[l]; a.rstd = t.rstd[l]; a.mask = t.mask[l];
            a.ext_mask = (drop && ds.ext) ? ds.ext + ext_off : nullptr;
            a.ext_gstride = ds.ext_net_bytes;
            a.gamma = m.ln ? params + m.offG[l] : nullptr;
            a.pstride = pstride;
            a.gstride = cap * m.ld[l + 1];
            a.cap = c->cap; a.N = m.dims[l + 1]; a.ld = m.ld[l + 1]; a.rows = rows;
            a.ln = m.ln ? 1 : 0; a.drop = drop ? 1 : 0;
            a.drop_p = m.drop; a.inv_keep = 1.0f / (1.0f - m.drop);
            a.seed = ds.seed * 0x100000001B3ull + (unsigned long long)(l + 1) * 0x9E3779B97F4A7C15ull;
            hipLaunchKernelGGL(ac_post_fwd_kernel, dim3((rows + 3) / 4, t.G), dim3(256), 0, s, a);
            LAUNCH_CHECK("ac_post_fwd");
        }
        if (!last) ext_off += (int64_t)rows * m.dims[l + 1];
    }
    return MORL_OK;
}

// backward of the same pass: t.g[L-1] holds dLoss/d(out).  grads ([G][P], fully overwritten) may be NULL (no parameter
// gradients wanted); need_dx -> t.dx = dLoss/d(input rows).  `dropped` = the forward ran with train-mode dropout.
static int mlp_backward(morl_ac_ctx* c, const Mlp& m, const float* params, int64_t pstride, Tape& t, int rows, int x_div,
                        bool dropped, float* grads, bool need_dx, hipStream_t s) {
    const long long cap = c->cap;
    for (int l = m.L - 1; l >= 0; --l) {
        if (l == 0 && !need_dx) break;
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
                a.pstride = m.P; a.gstride = cap * m.ld[l];
                a.N = m.dims[l]; a.ld = m.ld[l]; a.rows = rows;
                hipLaunchKernelGGL(ac_ln_grad_kernel, dim3((a.N + 255) / 256, t.G), dim3(256), 0, s, a);
                LAUNCH_CHECK("ac_ln_grad");
            }
            PostBwdArgs a{};
            a.d = t.g[hl]; a.h = t.h[hl]; a.xhat = t.zx[hl]; a.rstd = t.rstd[hl]; a.mask = t.mask[hl];
            a.gamma = m.ln ? params + m.offG[hl] : nullptr;
            a.pstride = pstride; a.gstride = cap * m.ld[l];
            a.cap = c->cap; a.N = m.dims[l]; a.ld = m.ld[l]; a.rows = rows;
            a.ln = m.ln ? 1 : 0; a.drop = drop ? 1 : 0;
            a.inv_keep = 1.0f / (1.0f - m.drop);
            hipLaunchKernelGGL(ac_post_bwd_kernel, dim3((rows + 3) / 4, t.G), dim3(256), 0, s, a);
            LAUNCH_CHECK("ac_post_bwd");
        }
    }
    if (grads) {
        GemmGroupBatched grp{};
        grp.n = m.L;
        grp.sC = m.P;
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
static int adam(float* params, float* grads, float* m, float* v, long long seg, int G, double lr, const int* steps,
                int step_add, const morl_ac_cfg* cfg, hipStream_t s) {
    const int nblk = std::min(256, stream_grid(seg, 256));
    hipLaunchKernelGGL(ac_adam_kernel, dim3(nblk, G), dim3(256), 0, s, params, (const float*)grads, m, v, seg, steps, step_add,
                       lr, cfg->beta1, cfg->beta2, (float)cfg->eps);
    LAUNCH_CHECK("ac_adam");
    return MORL_OK;
}

static int polyak(const float* src, float* dst, long long n, float tau, hipStream_t s) {
    hipLaunchKernelGGL(polyak_kernel, dim3(stream_grid(n, OPT_THREADS)), dim3(OPT_THREADS), 0, s, src, dst, n, tau, 1.0f - tau);
    LAUNCH_CHECK("ac_polyak");
    return MORL_OK;
}

static int head_forward(morl_ac_ctx* c, Tape& tp, int rows, const float* eps, const morl_ac_state* st,
                        const morl_ac_cfg* cfg, float* action, float* logp, bool save, hipStream_t s) {
    HeadArgs a{};
    a.head = tp.out;
    a.head_gstride = (long long)c->cap * c->pol.ld[c->pol.L];
    a.ldh = c->pol.ld[c->pol.L];
    a.eps = eps;
    a.scale = st->action_scale; a.bias = st->action_bias;
    a.action = action; a.logp = logp;
    a.save_y = save ? c->save_y : nullptr;
    a.save_std = save ? c->save_std : nullptr;
    a.rows = rows; a.Ad = c->d.act_dim; a.G = c->PG; a.algo = c->d.algo;
    a.policy_noise = cfg ? cfg->policy_noise : 0.f;
    a.noise_clip = cfg ? cfg->noise_clip : 0.f;
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
    if (!bt->obs || !bt->actions || !bt->rewards || !bt->next_obs || !bt->dones || !bt->w || !bt->eps_next)
        return fail(MORL_ERR_ARG, "batch has NULL arrays");
    const bool autotune = algo == MORL_AC_MOSAC && cfg->autotune;
    if (autotune && (!st->log_alpha || !st->log_alpha_exp_avg || !st->log_alpha_exp_avg_sq))
        return fail(MORL_ERR_ARG, "autotune needs log_alpha and its Adam state");
    const int iters = (algo == MORL_AC_MOSAC) ? std::max(1, cfg->policy_iters) : 1;
    if (cfg->do_policy && algo != MORL_AC_TD3 && !bt->eps_pi) return fail(MORL_ERR_ARG, "eps_pi is NULL");
    if (cfg->do_policy && autotune && !bt->eps_alpha) return fail(MORL_ERR_ARG, "eps_alpha is NULL");
    if (cfg->n_per < 0 || cfg->n_per > rows) return fail(MORL_ERR_ARG, "n_per %d outside 0..rows", cfg->n_per);
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

    hipLaunchKernelGGL(ac_alpha_prepare_kernel, dim3((c->PG + 63) / 64), dim3(64), 0, s, (const float*)st->log_alpha,
                       cfg->alpha, autotune ? 1 : 0, c->alpha_dev, c->PG);
    LAUNCH_CHECK("ac_alpha_prepare");

    // ---- critic phase: a' ~ pi(s'), target critics at (s', a'), critics at (s, a), TD loss, backward, Adam --------------
    if ((rc = concat(c, c->tp_a.x, P.ld[0], c->PG, rows, bt->next_obs, D, w_rows, wR, nullptr, 0, s))) return rc;
    if ((rc = mlp_forward(c, P, algo == MORL_AC_TD3 ? st->pol_target : st->pol, P.P, c->tp_a, rows, 1, nodrop, s))) return rc;
    if ((rc = head_forward(c, c->tp_a, rows, bt->eps_next, st, cfg, c->act, c->logp_next, false, s))) return rc;
    if ((rc = concat(c, c->tq_a.x, Q.ld[0], c->PG, rows, bt->next_obs, D, c->act, Ad, w_rows, wR, s))) return rc;
    if ((rc = mlp_forward(c, Q, st->q_target, Q.P, c->tq_a, rows, nq, dropspec(0), s))) return rc;
    if ((rc = concat(c, c->tq_b.x, Q.ld[0], c->PG, rows, bt->obs, D, bt->actions, Ad, w_rows, wR, s))) return rc;
    if ((rc = mlp_forward(c, Q, st->q, Q.P, c->tq_b, rows, nq, dropspec(1), s))) return rc;
    {
        CriticArgs a{};
        a.tq = c->tq_a.out; a.q = c->tq_b.out; a.dq = c->tq_b.g[Q.L - 1];
        a.gstride = q_gs; a.ldo = (int)q_ldo;
        a.logp_next = c->logp_next; a.rewards = bt->rewards; a.dones = bt->dones;
        a.w = bt->w; a.w_per_row = c->w_input ? 1 : 0;
        a.alpha_dev = c->alpha_dev;
        a.target_out = out->target_q; a.loss_out = out->critic_loss; a.q_losses = out->q_losses;
        a.priority = (algo == MORL_AC_TD3 && cfg->n_per > 0) ? out->priority : nullptr;
        a.n_per = cfg->n_per;
        a.rows = rows; a.R = R; a.nq = nq; a.algo = algo; a.gamma = cfg->gamma;
        hipLaunchKernelGGL(ac_critic_kernel, dim3(c->PG), dim3(256), 0, s, a);
        LAUNCH_CHECK("ac_critic");
    }
    if ((rc = mlp_backward(c, Q, st->q, Q.P, c->tq_b, rows, nq, Q.drop > 0.f, c->gq, false, s))) return rc;
    if (out->q_grads)
        HIP_TRY(hipMemcpyAsync(out->q_grads, c->gq, (size_t)c->QG * Q.P * sizeof(float), hipMemcpyDeviceToDevice, s));
    if ((rc = adam(st->q, c->gq, st->q_exp_avg, st->q_exp_avg_sq, (long long)nq * Q.P, PG, cfg->q_lr, st->q_steps,
                   st->q_steps ? 1 : cfg->q_step, cfg, s))) return rc;

    // ---- actor phase ---------------------------------------------------------------------------------------------------
    if (cfg->do_policy) {
        for (int it = 0; it < iters; ++it) {
            const float* eps_pi = (algo == MORL_AC_TD3) ? nullptr : bt->eps_pi + (long long)it * c->PG * rows * Ad;
            if ((rc = concat(c, c->tp_b.x, P.ld[0], c->PG, rows, bt->obs, D, w_rows, wR, nullptr, 0, s))) return rc;
            if ((rc = mlp_forward(c, P, st->pol, P.P, c->tp_b, rows, 1, nodrop, s))) return rc;
            if ((rc = head_forward(c, c->tp_b, rows, eps_pi, st, cfg, c->act, c->logp_pi, true, s))) return rc;
            if ((rc = concat(c, c->tq_a.x, Q.ld[0], c->PG, rows, bt->obs, D, c->act, Ad, w_rows, wR, s))) return rc;
            if ((rc = mlp_forward(c, Q, st->q, Q.P, c->tq_a, rows, nq, dropspec(2), s))) return rc;
            {
                ActorLossArgs a{};
                a.q = c->tq_a.out; a.dq = c->tq_a.g[Q.L - 1];
                a.gstride = q_gs; a.ldo = (int)q_ldo;
                a.logp = c->logp_pi; a.w = bt->w; a.w_per_row = c->w_input ? 1 : 0;
                a.alpha_dev = c->alpha_dev;
                a.loss_out = out->policy_loss;
                a.rows = rows; a.R = R; a.nq = nq; a.algo = algo;
                hipLaunchKernelGGL(ac_actor_loss_kernel, dim3(c->PG), dim3(256), 0, s, a);
                LAUNCH_CHECK("ac_actor_loss");
            }
            if ((rc = mlp_backward(c, Q, st->q, Q.P, c->tq_a, rows, nq, Q.drop > 0.f, nullptr, true, s))) return rc;
            {
                HeadBwdArgs a{};
                a.dx_q = c->tq_a.dx; a.dxq_gstride = (long long)c->cap * Q.ld[0];
                a.nq = nq; a.ld_qin = Q.ld[0]; a.col0 = D;
                a.head = c->tp_b.out; a.head_gstride = (long long)c->cap * P.ld[P.L]; a.ldh = P.ld[P.L];
                a.eps = eps_pi; a.save_y = c->save_y; a.save_std = c->save_std; a.logp = c->logp_pi;
                a.scale = st->action_scale; a.alpha_dev = c->alpha_dev;
                a.dhead = c->tp_b.g[P.L - 1];
                a.rows = rows; a.Ad = Ad; a.G = c->PG; a.algo = algo;
                hipLaunchKernelGGL(ac_head_bwd_kernel, dim3((c->PG * rows + 255) / 256), dim3(256), 0, s, a);
                LAUNCH_CHECK("ac_head_bwd");
            }
            if ((rc = mlp_backward(c, P, st->pol, P.P, c->tp_b, rows, 1, false, c->gp, false, s))) return rc;
            if (out->pol_grads)
                HIP_TRY(hipMemcpyAsync(out->pol_grads, c->gp, (size_t)c->PG * P.P * sizeof(float), hipMemcpyDeviceToDevice, s));
            if ((rc = adam(st->pol, c->gp, st->pol_exp_avg, st->pol_exp_avg_sq, P.P, PG, cfg->policy_lr, st->pol_steps,
                           (st->pol_steps ? 1 : cfg->policy_step) + it, cfg, s))) return rc;
            if (autotune) {
                // log-prob of a fresh sample under the UPDATED actor (mosac_continuous_action.py:467-468)
                const float* eps_al = bt->eps_alpha + (long long)it * c->PG * rows * Ad;
                if ((rc = mlp_forward(c, P, st->pol, P.P, c->tp_b, rows, 1, nodrop, s))) return rc;
                if ((rc = head_forward(c, c->tp_b, rows, eps_al, st, cfg, c->act, c->logp_pi, false, s))) return rc;
                hipLaunchKernelGGL(ac_alpha_step_kernel, dim3(c->PG), dim3(256), 0, s, st->log_alpha, st->log_alpha_exp_avg,
                                   st->log_alpha_exp_avg_sq, (const float*)c->logp_pi, rows, cfg->target_entropy,
                                   (const int*)st->pol_steps, (st->pol_steps ? 1 : cfg->policy_step) + it, cfg->alpha_lr,
                                   cfg->beta1, cfg->beta2, (float)cfg->eps, c->alpha_dev, out->alpha_loss);
                LAUNCH_CHECK("ac_alpha_step");
            }
            if (algo == MORL_AC_TD3)
                if ((rc = polyak(st->pol, st->pol_target, (long long)c->PG * P.P, cfg->tau, s))) return rc;
        }
    }
    if (cfg->do_target)
        if ((rc = polyak(st->q, st->q_target, (long long)c->QG * Q.P, cfg->tau, s))) return rc;
    if (st->q_steps || st->pol_steps) {
        hipLaunchKernelGGL(ac_step_advance_kernel, dim3((PG + 63) / 64), dim3(64), 0, s, st->q_steps,
                           cfg->do_policy ? st->pol_steps : (int32_t*)nullptr, PG, iters);
        LAUNCH_CHECK("ac_step_advance");
    }
    if (out->alpha)
        HIP_TRY(hipMemcpyAsync(out->alpha, c->alpha_dev, (size_t)c->PG * sizeof(float), hipMemcpyDeviceToDevice, s));
    return MORL_OK;
}

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
    if ((rc = mlp_forward(c, P, use_target ? st->pol_target : st->pol, P.P, c->tp_a, rows, 1, DropSpec(), s))) return rc;
    return head_forward(c, c->tp_a, rows, mode == 1 ? eps : nullptr, st, cfg, actions_out, logp_out, false, s);
}

extern "C" int morl_ac_q_forward(morl_ac_ctx* c, const morl_ac_state* st, const float* obs, const float* actions,
                                 const float* w, int rows, int use_target, float* q_out, void* stream) {
    int rc = check_state(c, st, rows);
    if (rc) return rc;
    if (!obs || !actions || !q_out) return fail(MORL_ERR_ARG, "obs / actions / q_out is NULL");
    if (c->w_input && !w) return fail(MORL_ERR_ARG, "this critic is weight-conditioned: w is NULL");
    if (use_target && !st->q_target) return fail(MORL_ERR_ARG, "q_target is NULL");
    hipStream_t s = (hipStream_t)stream;
    const Mlp& Q = c->q;
    const int R = c->d.reward_dim;
    c->PG = c->d.population; c->QG = c->PG * c->d.num_q;
    c->tq_a.G = c->QG;
    if ((rc = concat(c, c->tq_a.x, Q.ld[0], c->PG, rows, obs, c->d.obs_dim, actions, c->d.act_dim, c->w_input ? w : nullptr,
                     c->w_input ? R : 0, s))) return rc;
    if ((rc = mlp_forward(c, Q, use_target ? st->q_target : st->q, Q.P, c->tq_a, rows, c->d.num_q, DropSpec(), s))) return rc;
    // compact [QG][cap][ld] -> [QG][rows][R]
    ConcatArgs a{};
    a.src[0] = c->tq_a.out; a.width[0] = R; a.gstride[0] = (long long)c->cap * Q.ld[Q.L]; a.rstride[0] = Q.ld[Q.L];
    a.n_src = 1;
    a.dst = q_out; a.ld = R; a.dst_gstride = (long long)rows * R; a.rows = rows; a.G = c->QG;
    hipLaunchKernelGGL(ac_concat_kernel, dim3(stream_grid((long long)c->QG * rows * R, 256)), dim3(256), 0, s, a);
    LAUNCH_CHECK("ac_q_compact");
    return MORL_OK;
}
