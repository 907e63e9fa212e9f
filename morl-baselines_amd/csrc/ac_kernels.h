// Device kernels of the continuous-action actor-critic updates (CAPQL / MOSAC / GPI-PD continuous, gfx950 wave64).
//
// The networks are small ([256, 256] hidden, 128-256 batch rows): one learner's update is latency-bound, so every
// kernel here is BATCHED over a leading "net" axis -- the twin critics of a learner, and every learner of a MORL/D
// population -- and a whole population advances with the launch count of a single learner.  The dense layers reuse
// the exact-fp32 MFMA tile engine of gemm_f32.h (blockIdx.z = net); everything else (input assembly, Dropout /
// LayerNorm / ReLU, the squashed-Gaussian heads, TD targets, losses and their derivatives) is row-wise work:
// one wave per activation row with shuffle reductions, or one thread per batch row for the O(Ad + R) head math.
#pragma once
#include "gemm_f32.h"
#include "gemm_wave.h"
#include "morl_device.h"
#include "morl_hip.h"

namespace morl {

// ---------------------------------------------------------------------------------------------------------------------
// batched GEMMs: entry z of the batch reads A + (z / a_div) * sA (twin critics share their input rows), B + z * sB ...
// ---------------------------------------------------------------------------------------------------------------------
struct GemmBatched {
    GemmProblem p;
    long long sA, sB, sC, sBias, sMask;
    int a_div;
    // optional second segment: entries z >= split (split > 0) are the same layer shape on other buffers -- two passes
    // that do not depend on each other (target critics at (s', a') and online critics at (s, a)) share one launch
    int split;
    const float* A2;
    const float* B2;
    const float* bias2;
    float* C2;
};

__device__ __forceinline__ GemmProblem gemm_batched_select(const GemmBatched& b, int z) {
    GemmProblem g = b.p;
    if (b.split > 0 && z >= b.split) {
        z -= b.split;
        g.A = b.A2; g.B = b.B2; g.bias = b.bias2; g.C = b.C2;
    }
    g.A += (long long)(z / b.a_div) * b.sA;
    g.B += (long long)z * b.sB;
    g.C += (long long)z * b.sC;
    if (g.bias) g.bias += (long long)z * b.sBias;
    if (g.mask) g.mask += (long long)z * b.sMask;
    return g;
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_batched_kernel(GemmBatched b) {
    const GemmProblem g = gemm_batched_select(b, (int)blockIdx.z);
    const int id = (int)blockIdx.x;
    gemm_tile<A_KC, B_KC, EPI>(g, id / g.tiles_n, id % g.tiles_n, 0);
}

// every weight-gradient GEMM of one backward pass, for every net of the batch, in one launch:
// dW_l[o][i] = sum_rows dZ_l[row][o] * X_l[row][i], db_l[o] = sum_rows dZ_l[row][o]   (K = batch rows, one split)
struct GemmGroupBatched {
    GemmProblem p[MORL_MAX_LAYERS];
    long long sA[MORL_MAX_LAYERS], sB[MORL_MAX_LAYERS];
    int b_div[MORL_MAX_LAYERS];
    long long sC;                       // parameter count of one net
    int tile_start[MORL_MAX_LAYERS + 1];
    int n;
};

__global__ __launch_bounds__(GEMM_THREADS) void gemm_grouped_tn_batched_kernel(GemmGroupBatched grp) {
    kernarg_warm<sizeof(GemmGroupBatched)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    const int id = (int)blockIdx.x, z = (int)blockIdx.z;
    int q = 0;
    while (q + 1 < grp.n && id >= grp.tile_start[q + 1]) ++q;
    const int local = id - grp.tile_start[q];
    GemmProblem g = grp.p[q];
    g.A += (long long)z * grp.sA[q];
    g.B += (long long)(z / grp.b_div[q]) * grp.sB[q];
    g.C += (long long)z * grp.sC;
    g.colsum += (long long)z * grp.sC;
    // double-buffered tiles: the next 32-deep chunk's loads run under the MFMAs of the current one (single-buffered: 69 -> 66 us
    // for the critics of 64 learners)
    gemm_tile_tn_db(g, local / g.tiles_n, local % g.tiles_n, 0);
}

// wave-level variants (gemm_wave.h) for small batches of nets: 4 waves = 4 independent 32 x 32 tiles per workgroup;
// p.tiles_m / p.tiles_n count 32-wide tiles here
template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256) void gemm_wave_batched_kernel(GemmBatched b) {
    const int tile = (int)blockIdx.x * 4 + wave_id();
    if (tile >= b.p.tiles_m * b.p.tiles_n) return;
    const GemmProblem g = gemm_batched_select(b, (int)blockIdx.z);
    gemm_wave_tile<A_KC, B_KC, EPI>(g, tile / g.tiles_n, tile % g.tiles_n);
}

// split-K variants: one 32 x 32 tile per workgroup, the four waves share the contraction (gemm_wave4_tile)
template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256) void gemm_wave4_batched_kernel(GemmBatched b) {
    const GemmProblem g = gemm_batched_select(b, (int)blockIdx.z);
    const int tile = (int)blockIdx.x;
    gemm_wave4_tile<A_KC, B_KC, EPI>(g, tile / g.tiles_n, tile % g.tiles_n);
}

__global__ __launch_bounds__(256) void gemm_wave4_grouped_tn_batched_kernel(GemmGroupBatched grp) {
    kernarg_warm<sizeof(GemmGroupBatched)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    const int id = (int)blockIdx.x, z = (int)blockIdx.z;
    int q = 0;
    while (q + 1 < grp.n && id >= grp.tile_start[q + 1]) ++q;
    const int local = id - grp.tile_start[q];
    GemmProblem g = grp.p[q];
    g.A += (long long)z * grp.sA[q];
    g.B += (long long)(z / grp.b_div[q]) * grp.sB[q];
    g.C += (long long)z * grp.sC;
    g.colsum += (long long)z * grp.sC;
    gemm_wave4_tile<false, false, EPI_STORE>(g, local / g.tiles_n, local % g.tiles_n);
}

// the same launch with the optimiser step in the epilogue (gemm_wave.h: AdamTile): net z of the batch is learner z /
// nets_per_learner, whose device-resident step counter (or the host's step number) gives the bias corrections
struct AdamFuse {
    float* params; float* exp_avg; float* exp_avg_sq;   // [nets][P]
    float* wt;                  // [nets][P] K-major shadow copies, or NULL
    float* target;              // [nets][P] Polyak targets, or NULL
    long long offB[MORL_MAX_LAYERS];   // bias blocks (the weight blocks are grp.p[l].C - the gradient base)
    long long offW[MORL_MAX_LAYERS];
    const int* steps;           // [learners] or NULL
    const float* corr;          // [learners][2] bias-correction scalars left by an earlier kernel of the update, or NULL
    int step_add, nets_per_learner;
    double lr, b1, b2;
    float eps, tau;
    // when this is the last launch of the update that depends on the device-resident step counters (it reads `corr`, not them):
    // their advance, by the first workgroup of each learner
    int* adv_q; int* adv_p;
    int adv_p_by;
    // parameters whose gradients are NOT tiles of this launch (LayerNorm gains / shifts: ac_ln_grad_kernel left them in `grads`
    // before the dX pass last read the parameters): stepped element-wise by one extra workgroup per net
    const float* grads;         // [nets][P]
    long long extra_off[MORL_MAX_LAYERS];
    int extra_len[MORL_MAX_LAYERS];
    int n_extra;
};

__global__ __launch_bounds__(256) void gemm_wave4_grouped_tn_batched_adam_kernel(GemmGroupBatched grp, AdamFuse f) {
    kernarg_warm<sizeof(GemmGroupBatched) + sizeof(AdamFuse)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    const int id = (int)blockIdx.x, z = (int)blockIdx.z;
    int q = 0;
    while (q + 1 < grp.n && id >= grp.tile_start[q + 1]) ++q;
    const int local = id - grp.tile_start[q];
    GemmProblem g = grp.p[q];
    g.A += (long long)z * grp.sA[q];
    g.B += (long long)(z / grp.b_div[q]) * grp.sB[q];
    const long long net = (long long)z * grp.sC;
    AdamTile ad;
    ad.params = f.params + net; ad.exp_avg = f.exp_avg + net; ad.exp_avg_sq = f.exp_avg_sq + net;
    ad.wt = f.wt ? f.wt + net : nullptr;
    ad.target = f.target ? f.target + net : nullptr;
    ad.offW = f.offW[q]; ad.offB = f.offB[q];
    if (id == 0 && threadIdx.x == 0 && f.corr && z % f.nets_per_learner == 0) {
        if (f.adv_q) f.adv_q[z / f.nets_per_learner] += 1;
        if (f.adv_p) f.adv_p[z / f.nets_per_learner] += f.adv_p_by;
    }
    ad.corr = f.corr ? f.corr + 2 * (z / f.nets_per_learner) : nullptr;
    ad.step = f.corr ? 1 : max(1, (f.steps ? f.steps[z / f.nets_per_learner] : 0) + f.step_add);
    ad.lr = f.lr; ad.b1 = f.b1; ad.b2 = f.b2; ad.eps = f.eps; ad.tau = f.tau;
    if (id >= grp.tile_start[grp.n]) {              // (workgroup-uniform) the extra workgroup of this net
        AdamScalars c;
        if (ad.corr != nullptr) {
            c.neg_step_size = ad.corr[0]; c.bc2_sqrt = ad.corr[1];
            c.one_minus_b1 = (float)(1.0 - f.b1); c.b2 = (float)f.b2; c.one_minus_b2 = (float)(1.0 - f.b2); c.eps = f.eps;
        } else {
            c = adam_scalars(ad.step, f.lr, f.b1, f.b2, f.eps);
        }
        for (int x = 0; x < f.n_extra; ++x)
            for (int e = (int)threadIdx.x; e < f.extra_len[x]; e += (int)blockDim.x)
                adam_tile_apply(ad, c, f.extra_off[x] + e, f.extra_off[x] + e, f.grads[net + f.extra_off[x] + e]);
        return;
    }
    gemm_wave4_tile<false, false, EPI_STORE, true>(g, local / g.tiles_n, local % g.tiles_n, &ad);
}

__global__ __launch_bounds__(256) void gemm_wave_grouped_tn_batched_kernel(GemmGroupBatched grp) {
    kernarg_warm<sizeof(GemmGroupBatched)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    const int id = (int)blockIdx.x * 4 + wave_id(), z = (int)blockIdx.z;
    if (id >= grp.tile_start[grp.n]) return;
    int q = 0;
    while (q + 1 < grp.n && id >= grp.tile_start[q + 1]) ++q;
    const int local = id - grp.tile_start[q];
    GemmProblem g = grp.p[q];
    g.A += (long long)z * grp.sA[q];
    g.B += (long long)(z / grp.b_div[q]) * grp.sB[q];
    g.C += (long long)z * grp.sC;
    g.colsum += (long long)z * grp.sC;
    gemm_wave_tile<false, false, EPI_STORE>(g, local / g.tiles_n, local % g.tiles_n);
}

// ---------------------------------------------------------------------------------------------------------------------
// input assembly: dst[g][row][:] = cat(src0, src1, src2) (zero padded to ld).  A source with rstride 0 is one vector
// per net (MOSAC's scalarisation weights are not an input; CAPQL / TD3 weights are per row).
// ---------------------------------------------------------------------------------------------------------------------
struct ConcatArgs {
    const float* src[3];
    int width[3];
    long long gstride[3];
    int rstride[3];
    int n_src;
    float* dst;
    int ld;
    long long dst_gstride;
    int rows, G;
};

__device__ __forceinline__ void ac_concat_body(const ConcatArgs& a, int bx, int nbx) {
    const long long total = (long long)a.G * a.rows * a.ld;
    for (long long e = (long long)bx * blockDim.x + threadIdx.x; e < total; e += (long long)nbx * blockDim.x) {
        const int c = (int)(e % a.ld);
        const long long gr = e / a.ld;
        const int row = (int)(gr % a.rows), g = (int)(gr / a.rows);
        float v = 0.f;
        int c0 = c;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < a.n_src) {
                if (c0 >= 0 && c0 < a.width[s] && a.src[s] != nullptr)      // a NULL source leaves its columns zero
                    v = a.src[s][(long long)g * a.gstride[s] + (long long)row * a.rstride[s] + c0];
                c0 -= a.width[s];
            }
        }
        a.dst[(long long)g * a.dst_gstride + (long long)row * a.ld + c] = v;
    }
}

__global__ __launch_bounds__(256) void ac_concat_kernel(ConcatArgs a) { ac_concat_body(a, (int)blockIdx.x, (int)gridDim.x); }

// all network inputs of one update in a single launch (blockIdx.y = destination matrix)
struct ConcatMulti {
    ConcatArgs c[5];
    int n;
};

__global__ __launch_bounds__(256) void ac_concat_multi_kernel(ConcatMulti m) {
    kernarg_warm<sizeof(ConcatMulti)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    if ((int)blockIdx.y < m.n) ac_concat_body(m.c[blockIdx.y], (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// hidden-layer post-op of common/networks.py:10-48 after the Linear: [Dropout] -> [LayerNorm(eps 1e-5, affine)] -> ReLU.
// One wave per row (N <= 1024).  Saves what the backward needs: xhat (in place of z), rstd, the keep mask.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int POST_MAXJ = 16;   // columns per lane

struct PostArgs {
    float* z;                 // [G][..][ld]  in: Linear output; out: xhat (LayerNorm) or the dropped-out z
    float* h;                 // [G][..][ld]  out: post-ReLU activation
    float* rstd;              // [G][cap]
    unsigned long long* mask; // [G][cap][ceil(N / 64)] keep bits (written when dropout is active): bit l of word j <-> column 64 j + l
                              // -- one wave ballot and ONE 8-byte store per 64 columns (bytes: 16 store instructions per lane and row)
    const uint8_t* ext_mask;  // explicit keep flags [G][rows][N] (parity tests) or NULL -> counter-based RNG
    long long ext_gstride;
    const float* gamma;       // params + offset of this layer's LayerNorm weight; beta follows at +N
    long long pstride;
    long long gstride;        // floats between nets in z / h
    int cap;                  // rows capacity (stride of rstd / mask)
    int N, ld, rows;
    int ln, drop;
    float drop_p, inv_keep;
    unsigned long long seed;
};

// counter-based uniform in [0, 1) for the Dropout keep masks: element `idx` of the stream `seed` (one stream per update, pass and
// layer).  32-bit arithmetic throughout -- three v_mul_lo_u32 per element: the splitmix64 finaliser of rounds 1-4 (three 64-bit
// multiplies = a dozen quarter-rate instructions) was 8.5 of the 54 us of a LayerNorm / Dropout chain launch on MI355X
// (profiles/r05_ln_chain_post_costs.txt).  Two rounds of a multiply-xorshift mixer (Wellons' lowbias32 constants) over the counter
// Weyl-stepped into the MIXED seed (morl_device.h: dropout_seed_mix -- both seed words hashed first); the reference draws its masks from torch's generator (nn.Dropout), so only the statistics matter --
// and that every kernel computing a mask uses THIS function (mlp_chain16.h's post-op stage does).
__device__ __forceinline__ float ac_uniform(unsigned long long seed, unsigned long long idx) {
    unsigned int x = (unsigned int)idx * 0x9E3779B9u + dropout_seed_mix(seed);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ void ac_post_fwd_body(const PostArgs& a) {
    const int row = (int)blockIdx.x * 4 + wave_id();
    const int g = (int)blockIdx.y, lane = lane_id();
    if (row >= a.rows) return;
    float* __restrict__ z = a.z + (long long)g * a.gstride + (long long)row * a.ld;
    float* __restrict__ h = a.h + (long long)g * a.gstride + (long long)row * a.ld;
    float v[POST_MAXJ];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < POST_MAXJ; ++j) {
        const int c = lane + 64 * j;
        float x = 0.f;
        bool kept = false;
        if (c < a.N) {
            x = z[c];
            if (a.drop) {
                bool keep;
                if (a.ext_mask) keep = a.ext_mask[(long long)g * a.ext_gstride + (long long)row * a.N + c] != 0;
                else keep = ac_uniform(a.seed, ((unsigned long long)g * a.cap + row) * a.N + c) >= a.drop_p;
                kept = keep;
                x = keep ? x * a.inv_keep : 0.f;
            }
            sum += x;
        }
        if (a.drop && 64 * j < a.N) {                    // (wave-uniform: the row exists, this 64-column block too)
            const unsigned long long bits = __ballot(kept);
            if (lane == 0) a.mask[((long long)g * a.cap + row) * ((a.N + 63) >> 6) + j] = bits;
        }
        v[j] = x;
    }
    if (!a.ln) {
#pragma unroll
        for (int j = 0; j < POST_MAXJ; ++j) {
            const int c = lane + 64 * j;
            if (c < a.N) h[c] = fmaxf(v[j], 0.f);
        }
        return;
    }
    const float mean = wave_sum(sum) / (float)a.N;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < POST_MAXJ; ++j) {
        const int c = lane + 64 * j;
        if (c < a.N) { const float d = v[j] - mean; sq += d * d; }
    }
    const float var = wave_sum(sq) / (float)a.N;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (lane == 0) a.rstd[(long long)g * a.cap + row] = rstd;
    const float* __restrict__ gam = a.gamma + (long long)g * a.pstride;
    const float* __restrict__ bet = gam + a.N;
#pragma unroll
    for (int j = 0; j < POST_MAXJ; ++j) {
        const int c = lane + 64 * j;
        if (c < a.N) {
            const float xh = (v[j] - mean) * rstd;
            z[c] = xh;
            h[c] = fmaxf(xh * gam[c] + bet[c], 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void ac_post_fwd_kernel(PostArgs a) { ac_post_fwd_body(a); }
// two independent passes of the same layer (paired forward passes, GemmBatched::split): blockIdx.z picks the tape
__global__ __launch_bounds__(256) void ac_post_fwd_pair_kernel(PostArgs a, PostArgs b) { ac_post_fwd_body(blockIdx.z ? b : a); }

struct PostBwdArgs {
    float* d;                 // [G][..][ld]  in: dLoss/dh ; out: dLoss/dz
    const float* h;
    const float* xhat;
    const float* rstd;
    const unsigned long long* mask;      // keep bits, see PostArgs::mask
    const float* gamma;
    long long pstride, gstride;
    int cap, N, ld, rows;
    int ln, drop;
    float inv_keep;
};

__global__ __launch_bounds__(256) void ac_post_bwd_kernel(PostBwdArgs a) {
    const int row = (int)blockIdx.x * 4 + wave_id();
    const int g = (int)blockIdx.y, lane = lane_id();
    if (row >= a.rows) return;
    const long long base = (long long)g * a.gstride + (long long)row * a.ld;
    float* __restrict__ d = a.d + base;
    const float* __restrict__ h = a.h + base;
    float dxh[POST_MAXJ], xh[POST_MAXJ];
    float s1 = 0.f, s2 = 0.f;
    const float* __restrict__ gam = a.ln ? a.gamma + (long long)g * a.pstride : nullptr;
#pragma unroll
    for (int j = 0; j < POST_MAXJ; ++j) {
        const int c = lane + 64 * j;
        float t = 0.f, x = 0.f;
        if (c < a.N) {
            t = (h[c] > 0.f) ? d[c] : 0.f;                       // ReLU
            if (a.ln) {
                x = a.xhat[base + c];
                t *= gam[c];
                s1 += t;
                s2 += t * x;
            }
        }
        dxh[j] = t;
        xh[j] = x;
    }
    float m1 = 0.f, m2 = 0.f, rstd = 1.f;
    if (a.ln) {
        m1 = wave_sum(s1) / (float)a.N;
        m2 = wave_sum(s2) / (float)a.N;
        rstd = a.rstd[(long long)g * a.cap + row];
    }
#pragma unroll
    for (int j = 0; j < POST_MAXJ; ++j) {
        const int c = lane + 64 * j;
        if (c < a.N) {
            float t = dxh[j];
            if (a.ln) t = rstd * (t - m1 - xh[j] * m2);
            if (a.drop) t = ((a.mask[((long long)g * a.cap + row) * ((a.N + 63) >> 6) + j] >> lane) & 1ull) ? t * a.inv_keep : 0.f;
            d[c] = t;
        }
    }
}

// LayerNorm affine gradients: dgamma[c] = sum_rows dy * xhat, dbeta[c] = sum_rows dy, dy = dh * (h > 0).
// Must run BEFORE ac_post_bwd_kernel overwrites dh.  One workgroup of 16 waves per (64 columns, net): lane = column
// (coalesced 256-byte row segments), wave w sums rows w, w + 16, ...; the 16 partials are combined in wave order
// (deterministic).  The naive "one thread walks all rows of its column" form was the slowest kernel of a GPI-PD update
// (122 us for 256 rows: one dependent load chain per column on 512 threads).
struct LnGradArgs {
    const float* d;
    const float* h;
    const float* xhat;
    float* dgamma;            // grads + offset of gamma; dbeta follows at +N
    long long pstride, gstride;
    int N, ld, rows;
};

constexpr int COLRED_WAVES = 16;

__device__ __forceinline__ void ac_ln_grad_body(const LnGradArgs& a);
__global__ __launch_bounds__(64 * COLRED_WAVES) void ac_ln_grad_kernel(LnGradArgs a) { ac_ln_grad_body(a); }
// every hidden layer of a network in one launch (blockIdx.z = layer): the layer-fused backward of a LayerNorm net (mlp_chain16.h)
// leaves all the layers' dLoss/dh behind at once
struct LnGradMulti {
    LnGradArgs a[MORL_MAX_LAYERS];
    int n;
};
__global__ __launch_bounds__(64 * COLRED_WAVES) void ac_ln_grad_multi_kernel(LnGradMulti m) {
    if ((int)blockIdx.z < m.n && (int)blockIdx.x * 64 < m.a[blockIdx.z].N) ac_ln_grad_body(m.a[blockIdx.z]);
}

__device__ __forceinline__ void ac_ln_grad_body(const LnGradArgs& a) {
    __shared__ float s_g[COLRED_WAVES][64], s_b[COLRED_WAVES][64];
    const int lane = lane_id(), wave = wave_id();
    const int c = (int)blockIdx.x * 64 + lane, g = (int)blockIdx.y;
    float sg = 0.f, sb = 0.f;
    if (c < a.N) {
        const long long base = (long long)g * a.gstride + c;
        // eight rows of this wave per trip: their 24 loads are issued together (the plain loop waited for one row's three
        // loads at a time: 16 dependent round trips for 256 rows, 11.5 us), then summed in the same row order as before
        constexpr int U = 8;
        for (int r0 = wave; r0 < a.rows; r0 += COLRED_WAVES * U) {
            float hv[U], dv[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + COLRED_WAVES * u;
                const long long o = base + (long long)(r < a.rows ? r : r0) * a.ld;
                hv[u] = a.h[o]; dv[u] = a.d[o]; xv[u] = a.xhat[o];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r0 + COLRED_WAVES * u >= a.rows) break;
                const float dy = (hv[u] > 0.f) ? dv[u] : 0.f;
                sg += dy * xv[u];
                sb += dy;
            }
        }
    }
    s_g[wave][lane] = sg;
    s_b[wave][lane] = sb;
    __syncthreads();
    if (wave == 0 && c < a.N) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int w = 0; w < COLRED_WAVES; ++w) { tg += s_g[w][lane]; tb += s_b[w][lane]; }
        float* __restrict__ out = a.dgamma + (long long)g * a.pstride;
        out[c] = tg;
        out[a.N + c] = tb;
    }
}

// entropy coefficient of learner g: exp(log_alpha[g]) when it is learnt (mosac...:474 alpha_tensor = log_alpha.exp()),
// the constant otherwise
__device__ __forceinline__ float ac_alpha(const float* __restrict__ log_alpha, float alpha_const, int g) {
    return log_alpha ? expf(log_alpha[g]) : alpha_const;
}

// ---------------------------------------------------------------------------------------------------------------------
// policy heads.  One thread per (learner, batch row); O(Ad) work each.
//   CAPQL  capql.py:118-158      log_std = clamp(raw, -20, 2); log-prob summed per term, clamped to +-1e3
//   MOSAC  mosac_continuous_action.py:98-123   log_std = -5 + 3.5 * (tanh(raw) + 1)
//   TD3    gpi_pd_continuous_action.py:50-58   a = tanh(mean) [+ clamp(noise * policy_noise, +-noise_clip), clamp +-1]
// ---------------------------------------------------------------------------------------------------------------------
constexpr float AC_HALF_LOG_2PI = 0.9189385332046727f;   // log(sqrt(2*pi))

struct HeadArgs {
    const float* head;        // [G][cap][ldh]: mean (cols 0..Ad-1) | raw log_std (cols Ad..2Ad-1)
    long long head_gstride;
    int ldh;
    const float* eps;         // [G][rows][Ad] or NULL (deterministic action)
    const float* scale;       // [Ad]
    const float* bias;        // [Ad]
    float* action;            // [G][rows][Ad]
    float* logp;              // [G][rows] or NULL
    float* save_y;            // [G][rows][Ad] tanh output, or NULL
    float* save_std;          // [G][rows][Ad]
    float* xdst;              // optional: also write the action into columns [xcol0, xcol0 + Ad) of a critic-input matrix
    long long x_gstride;      // [G][cap][xld]
    int xld, xcol0;
    int rows, Ad, G, algo;
    float policy_noise, noise_clip;
};

__device__ __forceinline__ void ac_head_fwd_body(const HeadArgs& a) {
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (e >= a.G * a.rows) return;
    const int g = e / a.rows, row = e % a.rows;
    const float* __restrict__ hd = a.head + (long long)g * a.head_gstride + (long long)row * a.ldh;
    const long long o = ((long long)g * a.rows + row) * a.Ad;
    float* __restrict__ xd = a.xdst ? a.xdst + (long long)g * a.x_gstride + (long long)row * a.xld + a.xcol0 : nullptr;
    if (a.algo == MORL_AC_TD3) {
        for (int j = 0; j < a.Ad; ++j) {
            float t = tanhf(hd[j]);
            if (a.save_y) a.save_y[o + j] = t;
            if (a.eps) {
                const float n = fminf(fmaxf(a.eps[o + j] * a.policy_noise, -a.noise_clip), a.noise_clip);
                t = fminf(fmaxf(t + n, -1.f), 1.f);
            }
            const float act = t * a.scale[j] + a.bias[j];
            a.action[o + j] = act;
            if (xd) xd[j] = act;
        }
        return;
    }
    float lp_gauss = 0.f, lp_corr = 0.f, lp_elem = 0.f;
    for (int j = 0; j < a.Ad; ++j) {
        const float mean = hd[j];
        if (!a.eps) {                                      // deterministic: tanh(mean) * scale + bias
            const float act = tanhf(mean) * a.scale[j] + a.bias[j];
            a.action[o + j] = act;
            if (xd) xd[j] = act;
            continue;
        }
        const float raw = hd[a.Ad + j];
        float ls;
        if (a.algo == MORL_AC_CAPQL) ls = fminf(fmaxf(raw, -20.f), 2.f);
        else ls = -5.f + 0.5f * 7.f * (tanhf(raw) + 1.f);
        const float sd = expf(ls);
        const float x = mean + a.eps[o + j] * sd;
        const float y = tanhf(x);
        const float act = y * a.scale[j] + a.bias[j];
        a.action[o + j] = act;
        if (xd) xd[j] = act;
        const float t = x - mean;
        const float gauss = -(t * t) / (2.f * (sd * sd)) - logf(sd) - AC_HALF_LOG_2PI;
        const float corr = logf(a.scale[j] * (1.f - y * y) + 1e-6f);
        lp_gauss += gauss;
        lp_corr += corr;
        lp_elem += gauss - corr;
        if (a.save_y) { a.save_y[o + j] = y; a.save_std[o + j] = sd; }
    }
    if (a.logp && a.eps) {
        float lp;
        if (a.algo == MORL_AC_CAPQL) lp = fminf(fmaxf(lp_gauss - lp_corr, -1e3f), 1e3f);
        else lp = lp_elem;
        a.logp[(long long)g * a.rows + row] = lp;
    }
}

__global__ __launch_bounds__(256) void ac_head_fwd_kernel(HeadArgs a) { ac_head_fwd_body(a); }
// a' ~ pi(s') and a ~ pi(s) of one update (both actors' pre-activations come out of the same chain launch): blockIdx.y picks
__global__ __launch_bounds__(256) void ac_head_fwd_pair_kernel(HeadArgs a, HeadArgs b) { ac_head_fwd_body(blockIdx.y == 0 ? a : b); }

struct HeadBwdArgs {
    const float* dx_q;        // [G*nq][cap][ld_qin]  dLoss / d(critic input); the action columns start at col0
    long long dxq_gstride;
    int nq, ld_qin, col0;
    const float* head;
    long long head_gstride;
    int ldh;
    const float* eps;
    const float* save_y;
    const float* save_std;
    const float* logp;        // CAPQL: saturated rows pass no log-prob gradient
    const float* scale;
    const float* log_alpha;   // [G] or NULL (constant coefficient)
    float alpha_const;
    float* dhead;             // [G][cap][ldh] out
    int rows, Ad, G, algo;
};

// one (learner, batch row): writes the row of dhead; `v` (optional) also receives its first 16 entries -- the input row of the
// actor's backward chain when the chain's input stage calls this itself (HeadBwdHook)
__device__ __forceinline__ void ac_head_bwd_row(const HeadBwdArgs& a, int g, int row, float* v) {
    const long long o = ((long long)g * a.rows + row) * a.Ad;
    const float* __restrict__ hd = a.head + (long long)g * a.head_gstride + (long long)row * a.ldh;
    float* __restrict__ dh = a.dhead + (long long)g * a.head_gstride + (long long)row * a.ldh;
    float dlogp = 0.f;
    if (a.algo != MORL_AC_TD3) {
        dlogp = ac_alpha(a.log_alpha, a.alpha_const, g) / (float)a.rows;
        if (a.algo == MORL_AC_CAPQL && fabsf(a.logp[(long long)g * a.rows + row]) >= 1e3f) dlogp = 0.f;
    }
    for (int j = 0; j < a.Ad; ++j) {
        float dA = 0.f;
        for (int n = 0; n < a.nq; ++n)
            dA += a.dx_q[(long long)(g * a.nq + n) * a.dxq_gstride + (long long)row * a.ld_qin + a.col0 + j];
        const float y = a.save_y[o + j];
        const float one_m = 1.f - y * y;
        const float sc = a.scale[j];
        if (a.algo == MORL_AC_TD3) {
            const float d = dA * sc * one_m;
            dh[j] = d;
            if (v != nullptr && j < 16) v[j] = d;
            continue;
        }
        const float sd = a.save_std[o + j], ep = a.eps[o + j];
        const float du = dA * sc * one_m + dlogp * (2.f * y * sc * one_m) / (sc * one_m + 1e-6f);
        dh[j] = du;                                         // the Gaussian term's two paths to the mean cancel exactly
        if (v != nullptr && j < 16) v[j] = du;
        const float t = ep * sd;                            // x - mean
        const float var = sd * sd;
        const float dstd = du * ep + dlogp * (-t * ep / var + (t * t) / (var * sd) - 1.f / sd);
        const float dls = dstd * sd;
        const float raw = hd[a.Ad + j];
        float draw;
        if (a.algo == MORL_AC_CAPQL) draw = (raw >= -20.f && raw <= 2.f) ? dls : 0.f;
        else { const float th_ = tanhf(raw); draw = dls * 3.5f * (1.f - th_ * th_); }
        dh[a.Ad + j] = draw;
        if (v != nullptr && a.Ad + j < 16) v[a.Ad + j] = draw;
    }
    for (int j = ((a.algo == MORL_AC_TD3) ? a.Ad : 2 * a.Ad); j < a.ldh; ++j) dh[j] = 0.f;
}

__global__ __launch_bounds__(256) void ac_head_bwd_kernel(HeadBwdArgs a) {
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (e >= a.G * a.rows) return;
    ac_head_bwd_row(a, e / a.rows, e % a.rows, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// block-level deterministic sum (fixed order): every thread gets the total
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ac_block_sum(double v, double* s_red) {
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) s_red[wave_id()] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// TD target + critic loss + dLoss/dQ.  One workgroup per learner, threads stride over the batch rows.
//   CAPQL capql.py:326-334   target = r + (1-d) * gamma * (min_n Qt_n - alpha * logp')      (element-wise min)
//   MOSAC mosac...:436-450   scalarised: t = r.w + (1-d) * gamma * (min_n (Qt_n.w) - alpha * logp'); loss = sum_n mse
//   TD3   gpi_pd_c...:395-415  n* = argmin_n (Qt_n . w_row); target = r + (1-d) * gamma * Qt_n*; PER |q_0 - t| * 0.05 . w
// ---------------------------------------------------------------------------------------------------------------------
// The bias-correction scalars of the Adam step that follows (two fp64 pow()s per learner), computed by an otherwise idle thread
// of the loss kernel in front of the backward pass, for the optimiser step inside the weight-gradient launch (AdamFuse::corr)
struct AdamCorrOut {
    float* out;               // [G][2] or NULL
    const int* steps;         // [G] or NULL
    int step_add;
    double lr, b1, b2;
};
__device__ __forceinline__ void adam_corr_write(const AdamCorrOut& a, int g) {
    if (a.out != nullptr && threadIdx.x == blockDim.x - 1) {
        const AdamScalars c = adam_scalars(max(1, (a.steps ? a.steps[g] : 0) + a.step_add), a.lr, a.b1, a.b2, 0.f);
        a.out[2 * g] = c.neg_step_size;
        a.out[2 * g + 1] = c.bc2_sqrt;
    }
}

struct CriticArgs {
    const float* tq;          // [G*nq][cap][ldo]  target critics at (s', a')
    const float* q;           // [G*nq][cap][ldo]  critics at (s, a)
    float* dq;                // [G*nq][cap][ldo]
    long long gstride;
    int ldo;
    const float* logp_next;   // [G][rows]
    const float* rewards;     // [G][rows][R]
    const float* dones;       // [G][rows]
    const float* w;           // [G][rows][R] or [G][R]
    int w_per_row;
    const float* log_alpha;   // [G] or NULL (constant coefficient)
    float alpha_const;
    float* target_out;        // [G][rows][R] ([G][rows] MOSAC) or NULL
    float* loss_out;          // [G] or NULL
    float* q_losses;          // [G][nq] or NULL
    float* priority;          // [G][n_per] or NULL
    int n_per;
    int rows, R, nq, algo;
    float gamma;
    AdamCorrOut corr;
};

__global__ __launch_bounds__(256) void ac_critic_kernel(CriticArgs a) {
    kernarg_warm<sizeof(CriticArgs)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    __shared__ double s_red[4];
    const int g = (int)blockIdx.x;
    adam_corr_write(a.corr, g);
    const float alpha = (a.algo == MORL_AC_TD3) ? 0.f : ac_alpha(a.log_alpha, a.alpha_const, g);
    double part[4] = {0.0, 0.0, 0.0, 0.0};                  // per-critic squared-error sums (nq <= 4)
    for (int row = (int)threadIdx.x; row < a.rows; row += (int)blockDim.x) {
        const long long ro = (long long)row * a.ldo;
        const float* __restrict__ w = a.w + (a.w_per_row ? ((long long)g * a.rows + row) * a.R : (long long)g * a.R);
        const float* __restrict__ rew = a.rewards + ((long long)g * a.rows + row) * a.R;
        const float nd = 1.f - a.dones[(long long)g * a.rows + row];
        const float lp = (a.algo == MORL_AC_TD3) ? 0.f : a.logp_next[(long long)g * a.rows + row];
        float tgt[MORL_MAX_OBJ];
        if (a.algo == MORL_AC_CAPQL) {
            for (int r = 0; r < a.R; ++r) {
                float m = a.tq[(long long)(g * a.nq) * a.gstride + ro + r];
                for (int n = 1; n < a.nq; ++n) m = fminf(m, a.tq[(long long)(g * a.nq + n) * a.gstride + ro + r]);
                tgt[r] = rew[r] + (nd * a.gamma) * (m - alpha * lp);
            }
        } else if (a.algo == MORL_AC_TD3) {
            int best = 0;
            float bs = 0.f;
            for (int n = 0; n < a.nq; ++n) {
                float s = 0.f;
                for (int r = 0; r < a.R; ++r) s += a.tq[(long long)(g * a.nq + n) * a.gstride + ro + r] * w[r];
                if (n == 0 || s < bs) { bs = s; best = n; }
            }
            for (int r = 0; r < a.R; ++r)
                tgt[r] = rew[r] + (nd * a.gamma) * a.tq[(long long)(g * a.nq + best) * a.gstride + ro + r];
        } else {
            float m = 0.f, sr = 0.f;
            for (int n = 0; n < a.nq; ++n) {
                float s = 0.f;
                for (int r = 0; r < a.R; ++r) s += a.tq[(long long)(g * a.nq + n) * a.gstride + ro + r] * w[r];
                m = (n == 0) ? s : fminf(m, s);
            }
            for (int r = 0; r < a.R; ++r) sr += rew[r] * w[r];
            tgt[0] = sr + (nd * a.gamma) * (m - alpha * lp);
        }
        if (a.target_out) {
            if (a.algo == MORL_AC_MOSAC) a.target_out[(long long)g * a.rows + row] = tgt[0];
            else for (int r = 0; r < a.R; ++r) a.target_out[((long long)g * a.rows + row) * a.R + r] = tgt[r];
        }
        for (int n = 0; n < a.nq; ++n) {
            const float* __restrict__ q = a.q + (long long)(g * a.nq + n) * a.gstride + ro;
            float* __restrict__ dq = a.dq + (long long)(g * a.nq + n) * a.gstride + ro;
            if (a.algo == MORL_AC_MOSAC) {
                float s = 0.f;
                for (int r = 0; r < a.R; ++r) s += q[r] * w[r];
                const float e = s - tgt[0];
                part[n] += (double)e * (double)e;
                const float c = 2.f * e / (float)a.rows;
                for (int r = 0; r < a.R; ++r) dq[r] = c * w[r];
            } else {
                const float c = 2.f / ((float)a.nq * (float)a.rows * (float)a.R);
                for (int r = 0; r < a.R; ++r) {
                    const float e = q[r] - tgt[r];
                    part[n] += (double)e * (double)e;
                    dq[r] = c * e;
                }
            }
            for (int r = a.R; r < a.ldo; ++r) dq[r] = 0.f;
        }
        if (a.priority && row < a.n_per) {
            const float* __restrict__ q0 = a.q + (long long)(g * a.nq) * a.gstride + ro;
            float p = 0.f;
            for (int r = 0; r < a.R; ++r) p += (fabsf(q0[r] - tgt[r]) * 0.05f) * w[r];
            a.priority[(long long)g * a.n_per + row] = p;
        }
    }
    double total = 0.0;
    for (int n = 0; n < a.nq; ++n) {
        const double s = ac_block_sum(part[n], s_red);
        const double denom = (a.algo == MORL_AC_MOSAC) ? (double)a.rows : (double)a.rows * a.R;
        const double l = s / denom;
        if (threadIdx.x == 0 && a.q_losses) a.q_losses[(long long)g * a.nq + n] = (float)l;
        total += l;
    }
    if (threadIdx.x == 0 && a.loss_out)
        a.loss_out[g] = (float)((a.algo == MORL_AC_MOSAC) ? total : total / (double)a.nq);
}

// ---------------------------------------------------------------------------------------------------------------------
// actor loss and its derivative w.r.t. the critic outputs at (s, pi(s)).
//   CAPQL capql.py:341-346   mean(alpha * logp) - mean((min_n Q_n) . w)     (element-wise min over the critics)
//   MOSAC mosac...:454-460   mean(alpha * logp) - mean(min_n (Q_n . w))
//   TD3   gpi_pd_c...:424-426  -mean(((1/nq) sum_n Q_n) . w)
// ---------------------------------------------------------------------------------------------------------------------
struct ActorLossArgs {
    const float* q;           // [G*nq][cap][ldo]
    float* dq;
    long long gstride;
    int ldo;
    const float* logp;        // [G][rows]
    const float* w;
    int w_per_row;
    const float* log_alpha;   // [G] or NULL (constant coefficient)
    float alpha_const;
    float* loss_out;          // [G] or NULL
    int rows, R, nq, algo;
    AdamCorrOut corr;
};

__global__ __launch_bounds__(256) void ac_actor_loss_kernel(ActorLossArgs a) {
    kernarg_warm<sizeof(ActorLossArgs)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    __shared__ double s_red[4];
    const int g = (int)blockIdx.x;
    adam_corr_write(a.corr, g);
    const float alpha = (a.algo == MORL_AC_TD3) ? 0.f : ac_alpha(a.log_alpha, a.alpha_const, g);
    double s_lp = 0.0, s_q = 0.0;
    const float inv_rows = 1.f / (float)a.rows;
    for (int row = (int)threadIdx.x; row < a.rows; row += (int)blockDim.x) {
        const long long ro = (long long)row * a.ldo;
        const float* __restrict__ w = a.w + (a.w_per_row ? ((long long)g * a.rows + row) * a.R : (long long)g * a.R);
        for (int n = 0; n < a.nq; ++n) {
            float* __restrict__ dq = a.dq + (long long)(g * a.nq + n) * a.gstride + ro;
            for (int r = 0; r < a.ldo; ++r) dq[r] = 0.f;
        }
        float val = 0.f;
        if (a.algo == MORL_AC_CAPQL) {
            for (int r = 0; r < a.R; ++r) {
                int best = 0;
                float m = a.q[(long long)(g * a.nq) * a.gstride + ro + r];
                for (int n = 1; n < a.nq; ++n) {
                    const float v = a.q[(long long)(g * a.nq + n) * a.gstride + ro + r];
                    if (v < m) { m = v; best = n; }
                }
                val += m * w[r];
                a.dq[(long long)(g * a.nq + best) * a.gstride + ro + r] = -w[r] * inv_rows;
            }
        } else if (a.algo == MORL_AC_MOSAC) {
            int best = 0;
            for (int n = 0; n < a.nq; ++n) {
                float s = 0.f;
                for (int r = 0; r < a.R; ++r) s += a.q[(long long)(g * a.nq + n) * a.gstride + ro + r] * w[r];
                if (n == 0 || s < val) { val = s; best = n; }
            }
            for (int r = 0; r < a.R; ++r) a.dq[(long long)(g * a.nq + best) * a.gstride + ro + r] = -w[r] * inv_rows;
        } else {
            const float inv_nq = 1.f / (float)a.nq;
            for (int r = 0; r < a.R; ++r) {
                float m = 0.f;
                for (int n = 0; n < a.nq; ++n) m += a.q[(long long)(g * a.nq + n) * a.gstride + ro + r];
                val += (inv_nq * m) * w[r];
                for (int n = 0; n < a.nq; ++n)
                    a.dq[(long long)(g * a.nq + n) * a.gstride + ro + r] = -w[r] * inv_rows * inv_nq;
            }
        }
        s_q += (double)val;
        if (a.algo != MORL_AC_TD3) s_lp += (double)(alpha * a.logp[(long long)g * a.rows + row]);
    }
    const double tq = ac_block_sum(s_q, s_red);
    const double tl = ac_block_sum(s_lp, s_red);
    if (threadIdx.x == 0 && a.loss_out) a.loss_out[g] = (float)((tl - tq) / (double)a.rows);
}

// ---------------------------------------------------------------------------------------------------------------------
// MOSAC entropy coefficient.  export (optional output only): alpha[g] = autotune ? exp(log_alpha[g]) : alpha_const.
// step (mosac...:466-474): alpha_loss = mean(-log_alpha * (logp + target_entropy)); one scalar Adam step; alpha = exp.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void ac_alpha_prepare_kernel(const float* log_alpha, float alpha_const, int autotune, float* alpha_dev,
                                        int G) {
    const int g = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x;
    if (g < G) alpha_dev[g] = autotune ? expf(log_alpha[g]) : alpha_const;
}

__global__ __launch_bounds__(256) void ac_alpha_step_kernel(float* log_alpha, float* m_, float* v_, const float* logp,
                                                            int rows, float target_entropy, const int* steps,
                                                            int step_add, double lr, double db1, double db2, float eps,
                                                            float* alpha_loss_out, int* adv_q = nullptr, int* adv_p = nullptr,
                                                            int adv_p_by = 0) {
    __shared__ double s_red[4];
    const int g = (int)blockIdx.x;
    double s = 0.0;
    for (int row = (int)threadIdx.x; row < rows; row += (int)blockDim.x)
        s += (double)(logp[(long long)g * rows + row] + target_entropy);
    const double tot = ac_block_sum(s, s_red);
    if (threadIdx.x == 0) {
        const int t = max(1, (steps ? steps[g] : 0) + step_add);
        // (the last reader of the update's step counters advances them: see AdamFuse::adv_q)
        if (adv_q) adv_q[g] += 1;
        if (adv_p) adv_p[g] += adv_p_by;
        const float neg_step_size = (float)(-(lr / (1.0 - pow(db1, (double)t))));
        const float bc2_sqrt = (float)sqrt(1.0 - pow(db2, (double)t));
        const float one_minus_b1 = (float)(1.0 - db1), b2 = (float)db2, one_minus_b2 = (float)(1.0 - db2);
        const float mean = (float)(tot / (double)rows);
        const float la = log_alpha[g];
        if (alpha_loss_out) alpha_loss_out[g] = -la * mean;
        const float grad = -mean;
        float m = m_[g], v = v_[g];
        m = fmaf(one_minus_b1, __fsub_rn(grad, m), m);
        v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(one_minus_b2, grad), grad));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
        const float nla = __fadd_rn(la, __fmul_rn(neg_step_size, __fdiv_rn(m, denom)));
        log_alpha[g] = nla;
        m_[g] = m;
        v_[g] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// MOSAC with discrete actions (mosac_discrete_action.py:440-503).  One workgroup per learner, threads stride over rows.
//   critic: p', logp' = softmax / log_softmax of the actor logits at s'; v = sum_a p'_a * (min_n(Qt_n[a] . w) - alpha logp'_a);
//           target = r . w + (1 - d) * gamma * v;  loss = sum_n mse(Q_n[a_taken] . w, target)
//   actor : c_a = alpha logp_a - min_n(Q_n[a] . w);  loss = mean_{rows x A}(p_a c_a);
//           dLoss/dlogit_k = p_k (c_k - sum_a p_a c_a) / (rows * A)        (the alpha * d(logp) terms cancel)
//   alpha : loss = mean_{rows x A}(p_a * (-exp(log_alpha) * (logp_a + target_entropy))), same probabilities -> done in the
//           actor kernel right after the block sum (scalar Adam step on log_alpha)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SACD_MAX_A = 64;

__device__ __forceinline__ void sacd_softmax(const float* __restrict__ logits, int A, float* p, float* lp) {
    float mx = logits[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, logits[a]);
    float se = 0.f;
    for (int a = 0; a < A; ++a) se += expf(logits[a] - mx);
    const float lse = logf(se);
    for (int a = 0; a < A; ++a) {
        lp[a] = (logits[a] - mx) - lse;
        p[a] = expf(lp[a]);
    }
}

struct SacdCriticArgs {
    const float* logits_next; // [G][cap][ldp]
    long long p_gstride;
    int ldp;
    const float* tq;          // [G*2][cap][ldo]  target critics at s' (A*R per row)
    const float* q;           // [G*2][cap][ldo]  critics at s
    float* dq;
    long long gstride;
    int ldo;
    const float* actions;     // [G][rows] action index as float
    const float* rewards;     // [G][rows][R]
    const float* dones;       // [G][rows]
    const float* w;           // [G][R]
    const float* log_alpha;
    float alpha_const;
    float* target_out;        // [G][rows] or NULL
    float* loss_out;          // [G]
    float* q_losses;          // [G][2]
    int rows, A, R;
    float gamma;
};

__global__ __launch_bounds__(256) void sacd_critic_kernel(SacdCriticArgs a) {
    kernarg_warm<sizeof(SacdCriticArgs)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    __shared__ double s_red[4];
    const int g = (int)blockIdx.x;
    const float alpha = ac_alpha(a.log_alpha, a.alpha_const, g);
    const float* __restrict__ w = a.w + (long long)g * a.R;
    double part[2] = {0.0, 0.0};
    for (int row = (int)threadIdx.x; row < a.rows; row += (int)blockDim.x) {
        float p[SACD_MAX_A], lp[SACD_MAX_A];
        sacd_softmax(a.logits_next + (long long)g * a.p_gstride + (long long)row * a.ldp, a.A, p, lp);
        const long long ro = (long long)row * a.ldo;
        float v = 0.f;
        for (int ac = 0; ac < a.A; ++ac) {
            float m = 0.f;
            for (int n = 0; n < 2; ++n) {
                float s = 0.f;
                for (int r = 0; r < a.R; ++r) s += a.tq[(long long)(g * 2 + n) * a.gstride + ro + ac * a.R + r] * w[r];
                m = (n == 0) ? s : fminf(m, s);
            }
            v += p[ac] * (m - alpha * lp[ac]);
        }
        float sr = 0.f;
        for (int r = 0; r < a.R; ++r) sr += a.rewards[((long long)g * a.rows + row) * a.R + r] * w[r];
        const float tgt = sr + ((1.f - a.dones[(long long)g * a.rows + row]) * a.gamma) * v;
        if (a.target_out) a.target_out[(long long)g * a.rows + row] = tgt;
        const int act = (int)a.actions[(long long)g * a.rows + row];
        for (int n = 0; n < 2; ++n) {
            const float* __restrict__ q = a.q + (long long)(g * 2 + n) * a.gstride + ro;
            float* __restrict__ dq = a.dq + (long long)(g * 2 + n) * a.gstride + ro;
            for (int e = 0; e < a.ldo; ++e) dq[e] = 0.f;
            float s = 0.f;
            for (int r = 0; r < a.R; ++r) s += q[act * a.R + r] * w[r];
            const float e = s - tgt;
            part[n] += (double)e * (double)e;
            const float c = 2.f * e / (float)a.rows;
            for (int r = 0; r < a.R; ++r) dq[act * a.R + r] = c * w[r];
        }
    }
    double total = 0.0;
    for (int n = 0; n < 2; ++n) {
        const double l = ac_block_sum(part[n], s_red) / (double)a.rows;
        if (threadIdx.x == 0 && a.q_losses) a.q_losses[(long long)g * 2 + n] = (float)l;
        total += l;
    }
    if (threadIdx.x == 0 && a.loss_out) a.loss_out[g] = (float)total;
}

struct SacdActorArgs {
    const float* logits;      // [G][cap][ldp]
    float* dlogits;           // [G][cap][ldp]
    long long p_gstride;
    int ldp;
    const float* q;           // [G*2][cap][ldo]  (updated) critics at s
    long long gstride;
    int ldo;
    const float* w;           // [G][R]
    float* log_alpha;         // [G] or NULL (constant coefficient)
    float* la_m;
    float* la_v;
    float alpha_const;
    int autotune;
    float target_entropy;
    const int* steps;         // device Adam step counters of the actor or NULL
    int step_add;
    double alpha_lr, b1, b2;
    float eps;
    float* loss_out;          // [G] or NULL
    float* alpha_loss_out;    // [G] or NULL
    int rows, A, R;
};

__global__ __launch_bounds__(256) void sacd_actor_kernel(SacdActorArgs a) {
    kernarg_warm<sizeof(SacdActorArgs)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    __shared__ double s_red[4];
    const int g = (int)blockIdx.x;
    const float alpha = ac_alpha(a.autotune ? a.log_alpha : nullptr, a.alpha_const, g);
    const float* __restrict__ w = a.w + (long long)g * a.R;
    const float inv = 1.f / ((float)a.rows * (float)a.A);
    double s_loss = 0.0, s_ent = 0.0;
    for (int row = (int)threadIdx.x; row < a.rows; row += (int)blockDim.x) {
        float p[SACD_MAX_A], lp[SACD_MAX_A], c[SACD_MAX_A];
        sacd_softmax(a.logits + (long long)g * a.p_gstride + (long long)row * a.ldp, a.A, p, lp);
        const long long ro = (long long)row * a.ldo;
        float cbar = 0.f, ent = 0.f;
        for (int ac = 0; ac < a.A; ++ac) {
            float m = 0.f;
            for (int n = 0; n < 2; ++n) {
                float s = 0.f;
                for (int r = 0; r < a.R; ++r) s += a.q[(long long)(g * 2 + n) * a.gstride + ro + ac * a.R + r] * w[r];
                m = (n == 0) ? s : fminf(m, s);
            }
            c[ac] = alpha * lp[ac] - m;
            cbar += p[ac] * c[ac];
            ent += p[ac] * (lp[ac] + a.target_entropy);
        }
        float* __restrict__ dl = a.dlogits + (long long)g * a.p_gstride + (long long)row * a.ldp;
        for (int ac = 0; ac < a.A; ++ac) dl[ac] = p[ac] * (c[ac] - cbar) * inv;
        for (int e = a.A; e < a.ldp; ++e) dl[e] = 0.f;
        s_loss += (double)cbar;
        s_ent += (double)ent;
    }
    const double tl = ac_block_sum(s_loss, s_red);
    const double te = ac_block_sum(s_ent, s_red);
    if (threadIdx.x == 0) {
        if (a.loss_out) a.loss_out[g] = (float)(tl * (double)inv);
        if (a.autotune) {
            const float la = a.log_alpha[g];
            const float ea = expf(la);
            const float x = (float)(te * (double)inv);           // mean_{rows x A} p * (logp + target_entropy)
            if (a.alpha_loss_out) a.alpha_loss_out[g] = -ea * x;
            const float grad = -ea * x;                           // d/d(log_alpha) of -exp(log_alpha) * x
            const int t = max(1, (a.steps ? a.steps[g] : 0) + a.step_add);
            const float neg_step_size = (float)(-(a.alpha_lr / (1.0 - pow(a.b1, (double)t))));
            const float bc2_sqrt = (float)sqrt(1.0 - pow(a.b2, (double)t));
            const float one_minus_b1 = (float)(1.0 - a.b1), b2 = (float)a.b2, one_minus_b2 = (float)(1.0 - a.b2);
            float m = a.la_m[g], v = a.la_v[g];
            m = fmaf(one_minus_b1, __fsub_rn(grad, m), m);
            v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(one_minus_b2, grad), grad));
            const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), a.eps);
            // NOTE: the update must not be visible to this kernel's own alpha read above: every thread took `alpha` at entry
            a.log_alpha[g] = __fadd_rn(la, __fmul_rn(neg_step_size, __fdiv_rn(m, denom)));
            a.la_m[g] = m;
            a.la_v[g] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// torch _single_tensor_adam over one parameter segment per learner (blockIdx.y); the 1-based step is either the same
// for everybody (steps == NULL) or read from the learner's device-resident counter: t = steps[g] + step_add.
// Same arithmetic as clip_adam_kernel (optim_kernels.h) without the clipping.
// ---------------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------
// K-major shadow copies of the weight matrices.  nn.Linear stores W_l as [out][in] = contraction-contiguous per output
// column; a forward GEMM that reads it that way has to transpose 32 x 64 panels through LDS (gemm_wave.h), which measured
// 13 us against 7.4 us for the otherwise identical dX launch whose weight operand is read along the output index.  The
// update therefore keeps Wt_l [in][out] next to every parameter set: one launch at the start of an update fills them (the
// caller may have changed the parameters between calls) and the Adam kernel refreshes the entries it steps.  Values and
// the MFMA k order are unchanged -> results are bit-identical to the contraction-contiguous read.
// ---------------------------------------------------------------------------------------------------------------------
struct MlpLayout {
    long long offW[MORL_MAX_LAYERS];
    int K[MORL_MAX_LAYERS], N[MORL_MAX_LAYERS];
    int L;
    long long P;             // parameters of one net
};

// element p of a net's flat buffer -> its index in the K-major copy (non-matrix entries keep their place)
__device__ __forceinline__ long long mlp_transposed_index(const MlpLayout& t, long long p) {
    for (int l = 0; l < t.L; ++l) {
        const long long rel = p - t.offW[l];
        if (rel >= 0 && rel < (long long)t.K[l] * t.N[l]) {
            const int n = (int)(rel / t.K[l]), k = (int)(rel - (long long)n * t.K[l]);
            return t.offW[l] + (long long)k * t.N[l] + n;
        }
    }
    return p;
}

struct TransposeMulti {
    const float* src[4];
    float* dst[4];
    MlpLayout lay[4];
    int nets[4];
    int n;
};

// One workgroup = one 64 x 64 tile of one weight matrix of one net, through LDS: rows of the source ([out][in]) are read and
// rows of the copy ([in][out]) are written 256 bytes per wave instruction.  (The first version scattered 4-byte stores at a
// stride of one matrix row: 7 us for a single learner, but 278 us = 1.3 TB/s for the 175 MB of a 64-learner population.)  The
// last workgroup of a net (blockIdx.x == tiles of the net) copies the entries that are not part of a matrix (biases).
constexpr int TR_T = 64;

__device__ __forceinline__ int mlp_tiles(const MlpLayout& t) {
    int n = 0;
    for (int l = 0; l < t.L; ++l) n += ((t.N[l] + TR_T - 1) / TR_T) * ((t.K[l] + TR_T - 1) / TR_T);
    return n;
}

__device__ __forceinline__ void mlp_transpose_tile(const MlpLayout& t, const float* __restrict__ src, float* __restrict__ dst,
                                                   int tile, float (*sT)[TR_T + 1]) {
    const int r = (int)threadIdx.x >> 6, c = (int)threadIdx.x & 63;
    for (int l = 0; l < t.L; ++l) {
        const int N = t.N[l], K = t.K[l];
        const int tk = (K + TR_T - 1) / TR_T, cnt = ((N + TR_T - 1) / TR_T) * tk;
        if (tile >= cnt) { tile -= cnt; continue; }
        const int n0 = (tile / tk) * TR_T, k0 = (tile % tk) * TR_T;
        const float* W = src + t.offW[l];
        float* Wt = dst + t.offW[l];
#pragma unroll
        for (int j = 0; j < TR_T / 4; ++j) {
            const int n = n0 + r + 4 * j, k = k0 + c;
            if (n < N && k < K) sT[r + 4 * j][c] = W[(long long)n * K + k];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TR_T / 4; ++j) {
            const int k = k0 + r + 4 * j, n = n0 + c;
            if (k < K && n < N) Wt[(long long)k * N + n] = sT[c][r + 4 * j];
        }
        return;
    }
    // entries between / after the matrices keep their place
    for (int l = 0; l <= t.L; ++l) {
        const long long lo = (l == 0) ? 0 : t.offW[l - 1] + (long long)t.K[l - 1] * t.N[l - 1];
        const long long hi = (l == t.L) ? t.P : t.offW[l];
        for (long long p = lo + (long long)threadIdx.x; p < hi; p += blockDim.x) dst[p] = src[p];
    }
}

// Small totals (a single learner: 0.15 M parameters per set): one element per thread, scattered stores -- 7 us against the 9 us
// the tiled form needs for its two phases; the tiled form takes over where the scatter's bandwidth matters.
__device__ __forceinline__ void ac_transpose_scatter_body(const TransposeMulti& a, int q, int bx, int nbx) {
    if (q >= a.n) return;
    const MlpLayout& t = a.lay[q];
    const long long total = t.P * a.nets[q];
    for (long long e = (long long)bx * blockDim.x + threadIdx.x; e < total; e += (long long)nbx * blockDim.x) {
        const long long net = e / t.P, p = e - net * t.P;
        a.dst[q][net * t.P + mlp_transposed_index(t, p)] = a.src[q][e];
    }
}
__global__ __launch_bounds__(256) void ac_transpose_scatter_kernel(TransposeMulti a) {
    kernarg_warm<sizeof(TransposeMulti)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    ac_transpose_scatter_body(a, (int)blockIdx.z, (int)blockIdx.x, (int)gridDim.x);
}

// The two independent preparations at the start of a single learner's update in ONE launch (each is a few microseconds of work
// behind a launch boundary of its own): workgroups [0, cx * m.n) assemble the network inputs, the rest scatter the K-major shadow
// copies of the parameter sets.
__global__ __launch_bounds__(256) void ac_inputs_shadows_kernel(ConcatMulti m, TransposeMulti tm, int cx, int tx) {
    kernarg_warm<sizeof(ConcatMulti) + sizeof(TransposeMulti)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    const int b = (int)blockIdx.x;
    if (b < cx * m.n) ac_concat_body(m.c[b / cx], b % cx, cx);
    else ac_transpose_scatter_body(tm, (b - cx * m.n) / tx, (b - cx * m.n) % tx, tx);
}

// grid (max over the sets of tiles + 1, max nets, sets)
__global__ __launch_bounds__(256) void ac_transpose_multi_kernel(TransposeMulti a) {
    kernarg_warm<sizeof(TransposeMulti)>();       // (the block finds its group / piece by walking the argument block: see morl_device.h)
    __shared__ float sT[TR_T][TR_T + 1];
    const int q = (int)blockIdx.z, net = (int)blockIdx.y;
    if (q >= a.n || net >= a.nets[q]) return;
    const MlpLayout& t = a.lay[q];
    if ((int)blockIdx.x > mlp_tiles(t)) return;
    mlp_transpose_tile(t, a.src[q] + (long long)net * t.P, a.dst[q] + (long long)net * t.P, (int)blockIdx.x, sT);
}

__global__ __launch_bounds__(256) void ac_adam_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                      float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                      long long seg, const int* __restrict__ steps, int step_add,
                                                      double lr, double b1, double b2, float eps,
                                                      float* __restrict__ wt = nullptr, MlpLayout lay = MlpLayout{}) {
    const int g = (int)blockIdx.y;
    // the bias corrections need two fp64 pow(): once per workgroup, not per thread (at two elements per thread the pow()s
    // were most of the kernel: 58 us for the 9 M critic parameters of a 64-learner population, ~1 TB/s)
    __shared__ float s_corr[2];
    if (threadIdx.x == 0) {
        const AdamScalars c0 = adam_scalars(max(1, (steps ? steps[g] : 0) + step_add), lr, b1, b2, eps);
        s_corr[0] = c0.neg_step_size;
        s_corr[1] = c0.bc2_sqrt;
    }
    __syncthreads();
    AdamScalars c;
    c.neg_step_size = s_corr[0];
    c.bc2_sqrt = s_corr[1];
    c.one_minus_b1 = (float)(1.0 - b1); c.b2 = (float)b2; c.one_minus_b2 = (float)(1.0 - b2); c.eps = eps;
    const long long base = (long long)g * seg;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < seg; p += (long long)gridDim.x * blockDim.x) {
        const float gr = grads[base + p];
        float m = exp_avg[base + p], v = exp_avg_sq[base + p];
        const float np_ = adam_element(c, params[base + p], gr, m, v);
        params[base + p] = np_;
        exp_avg[base + p] = m;
        exp_avg_sq[base + p] = v;
        if (wt != nullptr) {      // keep the K-major shadow copy current (seg may span several nets of lay.P parameters)
            const long long net = p / lay.P, pp = p - net * lay.P;
            wt[base + net * lay.P + mlp_transposed_index(lay, pp)] = np_;
        }
    }
}

// last kernel of an update: Polyak of the target critics (common/networks.py:120-139; tau == 1 -> copy) fused with the
// advance of the device-resident Adam step counters (nothing reads them after the optimiser kernels)
__global__ __launch_bounds__(256) void ac_polyak_advance_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                long long n, float tau, float one_minus_tau,
                                                                int* q_steps, int* pol_steps, int G, int pol_inc) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (tau == 1.0f) dst[p] = src[p];
        else dst[p] = __fadd_rn(__fmul_rn(dst[p], one_minus_tau), __fmul_rn(tau, src[p]));
    }
    if (blockIdx.x == 0)
        for (int g = (int)threadIdx.x; g < G; g += (int)blockDim.x) {
            if (q_steps) q_steps[g] += 1;
            if (pol_steps) pol_steps[g] += pol_inc;
        }
}

__global__ void ac_step_advance_kernel(int* q_steps, int* pol_steps, int G, int pol_inc) {
    const int g = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x;
    if (g >= G) return;
    if (q_steps) q_steps[g] += 1;
    if (pol_steps) pol_steps[g] += pol_inc;
}

}  // namespace morl
