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

__global__ __launch_bounds__(THREADS, 1) void chain_split_bf16(Net net, const float* __restrict__ x, float* __restrict__ q) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
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
                    if (ct < n_ct) acc[rt][ct] = six(a[rt], b[ct], acc[rt][ct]);
        };
        bf16x8 bA[2][3], bB[2][3], aA[2][3], aB[2][3];
        load_b(bA, s_lo);
        load_a(aA, s_lo);
        for (int st = s_lo; st < s_hi; st += 2) {
            const int s1 = min(st + 1, s_hi - 1), s2 = min(st + 2, s_hi - 1);
            load_b(bB, s1);
            load_a(aB, s1);
            mul(aA, bA);
            load_b(bA, s2);
            load_a(aA, s2);
            if (st + 1 < s_hi) mul(aB, bB);
        }
        __syncthreads();                // every wave is past its last read of the activations
        if (!wide) {
            // split-K partials through LDS (the activation buffer is free): [wave][64 rows][32 cols] fp32
            float* red = reinterpret_cast<float*>(lds);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    red[(wave * TM + m) * 32 + rl] = acc[rt][0][r];
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
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int col = col_base + ct * 32 + rl;
                    const float bias = ly.bias[col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        float v = acc[rt][ct][r] + bias;
                        if (ly.relu) v = fmaxf(v, 0.f);
                        unsigned short h, md, lo;
                        split3(v, h, md, lo);
                        act[0][m * LDA + col] = h; act[1][m * LDA + col] = md; act[2][m * LDA + col] = lo;
                    }
                }
        }
        __syncthreads();
    }
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
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_split_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(chain_split_bf16, dim3(ROWS / TM), dim3(THREADS), lds_bytes, 0, net, dx, dq);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chain_split_bf16, dim3(ROWS / TM), dim3(THREADS), lds_bytes, 0, net, dx, dq);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
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
