// Development probe (NOT part of the product): one no-grad forward pass of the flagship Q-network
// (35 -> 256 -> 256 -> 256 -> 256 -> 18, 16 384 rows) with every fp32 GEMM evaluated as SIX split-bf16 products
// (a = a_hi + a_mid + a_lo, 8 + 8 + 8 mantissa bits; hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi) on v_mfma_f32_32x32x16_bf16,
// accumulated in fp32.  Prints the time per pass and the error of sampled rows against a float64 forward on the host, next to
// the error of a plain float32 forward -- the question being whether the library's 61 us fp32-MFMA pass has an fp32-accurate
// successor at the bf16 rate.     hipcc --offload-arch=gfx950 -O3 split_bf16_chain_probe.hip -o split_bf16_chain_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ROWS = 16384, D_IN = 35, H = 256, N_OUT = 18, L = 5;
constexpr int TM = 64;                 // rows per workgroup
constexpr int LDA = H + 8;             // bf16 elements per activation row (+16 bytes: conflict-free 16-byte reads down a column of rows)
constexpr int THREADS = 256;

struct Layer {
    const unsigned short* w[3];        // split weights, [kpad / 8][npad][8] bf16: the 8 contraction indices of a lane are 16 contiguous bytes
    const float* bias;                 // [npad]
    int kpad, npad, relu;              // kpad multiple of 16, npad multiple of 32
};
struct Net { Layer l[L]; };

__device__ __forceinline__ unsigned short f2bf(float x) {      // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    hi = f2bf(x);
    const float r1 = x - bf2f(hi);
    mid = f2bf(r1);
    lo = f2bf(r1 - bf2f(mid));
}

__device__ __forceinline__ f32x16 mfma_bf(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// six products, smallest magnitude first
__device__ __forceinline__ f32x16 six(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
    c = mfma_bf(a[2], b[0], c);   // lo * hi
    c = mfma_bf(a[1], b[1], c);   // mid * mid
    c = mfma_bf(a[0], b[2], c);   // hi * lo
    c = mfma_bf(a[1], b[0], c);   // mid * hi
    c = mfma_bf(a[0], b[1], c);   // hi * mid
    c = mfma_bf(a[0], b[0], c);   // hi * hi
    return c;
}

// MODE 0: the pass.  Decomposition runs (wrong results, timing only): 1 = operand traffic and epilogue without the MFMAs,
// 2 = MFMAs and operand traffic with a trivial epilogue (hi part stored three times, no splitting), 3 = MFMAs only (operands
// loaded once per layer, trivial epilogue)
// 4 = like 3 with the six products of the four accumulators interleaved (no two consecutive MFMAs share an accumulator)
__device__ long long g_clk[4];          // [0..1] shader cycles, [2..3] 100 MHz wall clock, first workgroup
template <int MODE>
__global__ __launch_bounds__(THREADS, 1) void chain_split_bf16(Net net, const float* __restrict__ x, float* __restrict__ q) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter(); g_clk[2] = wall_clock64(); }
    unsigned short* act[3] = {lds, lds + TM * LDA, lds + 2 * TM * LDA};      // hi / mid / lo, [TM][LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * TM;
    // ---- input rows, split, zero-padded to the first layer's kpad
    for (int e = tid; e < TM * net.l[0].kpad; e += THREADS) {
        const int m = e / net.l[0].kpad, k = e % net.l[0].kpad;
        const float v = (k < D_IN) ? x[(size_t)(row0 + m) * D_IN + k] : 0.f;
        unsigned short h, md, lo;
        split3(v, h, md, lo);
        act[0][m * LDA + k] = h; act[1][m * LDA + k] = md; act[2][m * LDA + k] = lo;
    }
    __syncthreads();
    const int rl = lane & 31, kg = lane >> 5;                  // operand row / column within a 32-tile, k-group (8 indices) of a 16-step
    bf16x8 bA[2][3];                                           // first weight set of a layer: fetched before the previous layer's epilogue
    bool have_bA = false;
    for (int s = 0; s < L; ++s) {
        const Layer& ly = net.l[s];
        const int steps = ly.kpad >> 4;
        const bool wide = ly.npad > 32;
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        // wide: wave w owns columns [64w, 64w + 64) (2 column tiles) over all k; narrow (npad == 32): the waves split k, one column tile
        const int n_ct = wide ? 2 : 1;
        const int s_lo = wide ? 0 : (steps * wave) / 4, s_hi = wide ? steps : (steps * (wave + 1)) / 4;
        const int col_base = wide ? wave * 64 : 0;
        // software pipeline: the operands of step st + 1 are in flight while step st multiplies.  Loads are unconditional (clamped
        // step index) so that the compiler can count them (a first version with guarded loads waited for everything: 96 us)
        auto load_b = [&](bf16x8 (&b)[2][3], int st) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[ct][p] = *reinterpret_cast<const bf16x8*>(ly.w[p] + ((size_t)(st * 2 + kg) * ly.npad + col_base + (ct < n_ct ? ct : 0) * 32 + rl) * 8);
        };
        auto load_a = [&](bf16x8 (&a)[2][3], int st) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[rt][p] = *reinterpret_cast<const bf16x8*>(act[p] + (rt * 32 + rl) * LDA + st * 16 + kg * 8);
        };
        auto mul = [&](const bf16x8 (&a)[2][3], const bf16x8 (&b)[2][3]) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    if (ct < n_ct) {
                        if (MODE == 4) continue;
                        if (MODE == 1) acc[rt][ct][0] += (float)a[rt][0][0] + (float)b[ct][0][0] + (float)a[rt][1][1] + (float)b[ct][1][1] + (float)a[rt][2][2] + (float)b[ct][2][2];
                        else acc[rt][ct] = six(b[ct], a[rt], acc[rt][ct]);      // weights are the A operand: D[feature][batch row]
                    }
        };
        bf16x8 bB[2][3], aA[2][3], aB[2][3];
        if (!have_bA) load_b(bA, s_lo);
        load_a(aA, s_lo);
        // this layer's bias quads (wide layers), in flight under the MFMA loop
        float4 bias_q[2][4];
        if (wide)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) bias_q[ct][g4] = *reinterpret_cast<const float4*>(ly.bias + col_base + ct * 32 + 8 * g4 + 4 * kg);
        if (MODE == 3) {
            for (int st = s_lo; st < s_hi; ++st) mul(aA, bA);
        } else if (MODE == 4) {
            constexpr int pa[6] = {2, 1, 0, 1, 0, 0}, pb[6] = {0, 1, 2, 0, 1, 0};
            for (int st = s_lo; st < s_hi; ++st) {
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            if (ct < n_ct) acc[rt][ct] = mfma_bf(bA[ct][pb[p]], aA[rt][pa[p]], acc[rt][ct]);
            }
        } else
        for (int st = s_lo; st < s_hi; st += 2) {
            const int s1 = min(st + 1, s_hi - 1), s2 = min(st + 2, s_hi - 1);
            load_b(bB, s1);
            load_a(aB, s1);
            mul(aA, bA);
            load_b(bA, s2);
            load_a(aA, s2);
            if (st + 1 < s_hi) mul(aB, bB);
        }
        have_bA = false;
        if (MODE != 3 && MODE != 4 && s + 1 < L) {
            // the next layer's first weight set rides under this layer's epilogue and barriers
            const Layer& nl = net.l[s + 1];
            const bool nwide = nl.npad > 32;
            const int nsteps = nl.kpad >> 4;
            const int n_lo = nwide ? 0 : (nsteps * wave) / 4, ncb = nwide ? wave * 64 : 0;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bA[ct][p] = *reinterpret_cast<const bf16x8*>(nl.w[p] + ((size_t)(n_lo * 2 + kg) * nl.npad + ncb + ((nwide && ct == 1) ? 32 : 0) + rl) * 8);
            have_bA = true;
        }
        __syncthreads();                // every wave is past its last read of the activations
        if (!wide) {
            // split-K partials through LDS (the activation buffer is free): [wave][64 rows][32 cols] fp32
            float* red = reinterpret_cast<float*>(lds);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    red[(wave * TM + rt * 32 + rl) * 32 + n] = acc[rt][0][r];
                }
            __syncthreads();
            for (int e = tid; e < TM * 32; e += THREADS) {
                const int m = e >> 5, n = e & 31;
                float v = red[(0 * TM + m) * 32 + n];
                v += red[(1 * TM + m) * 32 + n];
                v += red[(2 * TM + m) * 32 + n];
                v += red[(3 * TM + m) * 32 + n];
                v += ly.bias[n];
                if (n < N_OUT) q[(size_t)(row0 + m) * N_OUT + n] = v;       // (last layer of this probe)
            }
        } else {
            // lane = batch row, registers = features: the four registers of a group are four CONSECUTIVE features -> one 8-byte LDS
            // store per split part and group (the first version, lane = feature, issued 192 two-byte stores per lane and layer:
            // 24 of its 51 us)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int m = rt * 32 + rl;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int f0 = col_base + ct * 32 + 8 * g4 + 4 * kg;
                        const float4 bias = bias_q[ct][g4];
                        const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
                        unsigned short h[4], md[4], lo[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float v = acc[rt][ct][4 * g4 + u] + bv[u];
                            if (ly.relu) v = fmaxf(v, 0.f);
                            if (MODE >= 2) { h[u] = md[u] = lo[u] = (unsigned short)(__float_as_uint(v) >> 16); }
                            else split3(v, h[u], md[u], lo[u]);
                        }
                        *reinterpret_cast<uint2*>(act[0] + m * LDA + f0) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                        *reinterpret_cast<uint2*>(act[1] + m * LDA + f0) = make_uint2(md[0] | ((unsigned)md[1] << 16), md[2] | ((unsigned)md[3] << 16));
                        *reinterpret_cast<uint2*>(act[2] + m * LDA + f0) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
                    }
                }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[1] = __builtin_readcyclecounter(); g_clk[3] = wall_clock64(); }
}

// pure issue-rate check of the instruction: NACC independent accumulators per wave, `blocks` workgroups of 4 waves
template <int NACC>
__global__ __launch_bounds__(THREADS) void mfma_rate(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.001f * (threadIdx.x + e)); y[e] = (__bf16)(0.002f * (threadIdx.x % 13 + e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) acc[u % NACC] = mfma_bf(x, y, acc[u % NACC]);
    }
    float sum = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * THREADS + threadIdx.x] = sum;
}

static unsigned short h_f2bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float h_bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int dims[L + 1] = {D_IN, H, H, H, H, N_OUT};
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    std::vector<std::vector<float>> W(L), Bv(L);
    for (int l = 0; l < L; ++l) {
        W[l].resize((size_t)dims[l + 1] * dims[l]);
        Bv[l].resize(dims[l + 1]);
        const float sc = 1.7f / sqrtf((float)dims[l]);
        for (auto& v : W[l]) v = rnd() * sc;
        for (auto& v : Bv[l]) v = rnd() * 0.1f;
    }
    std::vector<float> X((size_t)ROWS * D_IN);
    for (auto& v : X) v = rnd();
    Net net{};
    for (int l = 0; l < L; ++l) {
        const int K = dims[l], N = dims[l + 1];
        const int kpad = (K + 15) / 16 * 16, npad = (N + 31) / 32 * 32;
        std::vector<unsigned short> s[3];
        for (int p = 0; p < 3; ++p) s[p].assign((size_t)kpad * npad, 0);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) {
                const float w = W[l][(size_t)n * K + k];               // nn.Linear layout [out][in]
                const unsigned short hi = h_f2bf(w);
                const float r1 = w - h_bf2f(hi);
                const unsigned short mid = h_f2bf(r1);
                const unsigned short lo = h_f2bf(r1 - h_bf2f(mid));
                const size_t idx = ((size_t)(k / 8) * npad + n) * 8 + (k % 8);
                s[0][idx] = hi; s[1][idx] = mid; s[2][idx] = lo;
            }
        std::vector<float> bp(npad, 0.f);
        for (int n = 0; n < N; ++n) bp[n] = Bv[l][n];
        for (int p = 0; p < 3; ++p) {
            unsigned short* d;
            CK(hipMalloc(&d, s[p].size() * 2));
            CK(hipMemcpy(d, s[p].data(), s[p].size() * 2, hipMemcpyHostToDevice));
            net.l[l].w[p] = d;
        }
        float* db;
        CK(hipMalloc(&db, npad * 4));
        CK(hipMemcpy(db, bp.data(), npad * 4, hipMemcpyHostToDevice));
        net.l[l].bias = db;
        net.l[l].kpad = kpad; net.l[l].npad = npad; net.l[l].relu = (l < L - 1);
    }
    float *dx, *dq;
    CK(hipMalloc(&dx, X.size() * 4));
    CK(hipMalloc(&dq, (size_t)ROWS * N_OUT * 4));
    CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    const size_t lds_bytes = (size_t)3 * TM * LDA * 2;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 50;
    float ms = 0;
    auto time_mode = [&](auto kernel, const char* what) -> int {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kernel, dim3(ROWS / TM), dim3(THREADS), lds_bytes, 0, net, dx, dq);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3(ROWS / TM), dim3(THREADS), lds_bytes, 0, net, dx, dq);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        long long clk[4];
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
        printf("  %-78s %6.1f us   (first workgroup: %.2f GHz shader clock over its %.1f us)\n", what, t * 1e3 / reps,
               (double)(clk[1] - clk[0]) / ((double)(clk[3] - clk[2]) * 10.0), (double)(clk[3] - clk[2]) / 100.0);
        ms = t;
        return 0;
    };
    {   // issue rate of v_mfma_f32_32x32x16_bf16 from one and from two waves per SIMD
        float* dr;
        CK(hipMalloc(&dr, (size_t)1024 * THREADS * 4));
        auto rate = [&](auto kernel, int blocks, const char* what) -> int {
            const int iters = 512;
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), 0, 0, dr, iters);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), 0, 0, dr, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            const double n = 24.0 * iters;
            printf("  %-60s %6.2f ns per MFMA per wave, %7.1f TFLOP/s (bf16) chip-wide\n", what, t * 1e6 / n,
                   (double)blocks * 4 * n * 32768.0 / (t * 1e-3) / 1e12);
            return 0;
        };
        printf("v_mfma_f32_32x32x16_bf16 issue rate:\n");
        if (rate(mfma_rate<4>, 256, "4 accumulators, 1 wave per SIMD (256 workgroups)")) return 1;
        if (rate(mfma_rate<4>, 512, "4 accumulators, 2 waves per SIMD (512 workgroups)")) return 1;
        if (rate(mfma_rate<2>, 512, "2 accumulators, 2 waves per SIMD")) return 1;
        if (rate(mfma_rate<4>, 1024, "4 accumulators, 4 waves per SIMD (1024 workgroups)")) return 1;
    }
    printf("decomposition (timing only, results of these three are not the forward pass):\n");
    if (time_mode(chain_split_bf16<1>, "operand traffic + epilogue, no MFMAs")) return 1;
    if (time_mode(chain_split_bf16<2>, "MFMAs + operand traffic, trivial epilogue (no splitting)")) return 1;
    if (time_mode(chain_split_bf16<3>, "MFMAs only (operands loaded once per layer, trivial epilogue)")) return 1;
    if (time_mode(chain_split_bf16<4>, "MFMAs only, the products of the four accumulators interleaved")) return 1;
    printf("the pass:\n");
    if (time_mode(chain_split_bf16<0>, "six-product forward pass")) return 1;
    std::vector<float> Q((size_t)ROWS * N_OUT);
    CK(hipMemcpy(Q.data(), dq, Q.size() * 4, hipMemcpyDeviceToHost));
    // float64 and float32 forwards of 256 sampled rows on the host
    double worst64 = 0, worst32 = 0, qmax = 0;
    for (int t = 0; t < 256; ++t) {
        const int row = (t * 6151) % ROWS;
        std::vector<double> a(X.begin() + (size_t)row * D_IN, X.begin() + (size_t)(row + 1) * D_IN);
        std::vector<float> a32(a.begin(), a.end());
        for (int l = 0; l < L; ++l) {
            std::vector<double> o(dims[l + 1]);
            std::vector<float> o32(dims[l + 1]);
            for (int n = 0; n < dims[l + 1]; ++n) {
                double acc = Bv[l][n];
                float acc32 = 0.f;
                for (int k = 0; k < dims[l]; ++k) { acc += (double)W[l][(size_t)n * dims[l] + k] * a[k]; acc32 = fmaf(W[l][(size_t)n * dims[l] + k], a32[k], acc32); }
                acc32 += Bv[l][n];
                if (l < L - 1) { acc = acc > 0 ? acc : 0; acc32 = acc32 > 0 ? acc32 : 0; }
                o[n] = acc; o32[n] = acc32;
            }
            a = o; a32 = o32;
        }
        for (int n = 0; n < N_OUT; ++n) {
            qmax = fmax(qmax, fabs(a[n]));
            worst64 = fmax(worst64, fabs((double)Q[(size_t)row * N_OUT + n] - a[n]));
            worst32 = fmax(worst32, fabs((double)a32[n] - a[n]));
        }
    }
    const double flop = (double)ROWS * 2 * (35.0 * 256 + 3 * 256.0 * 256 + 256.0 * 18);
    printf("split-bf16 x 6 forward pass: %.1f us per pass of %d rows  (%.1f TFLOP/s fp32-equivalent; the library's fp32-MFMA pass: 61 us)\n",
           ms * 1e3 / reps, ROWS, flop / (ms * 1e-3 / reps) / 1e12);
    printf("max |Q - Q_float64| over 256 sampled rows: split-bf16 %.3e, plain float32 (k-ordered fma chain) %.3e   (max |Q| %.3f)\n",
           worst64, worst32, qmax);
    return 0;
}
