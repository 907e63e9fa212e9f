// Development probe (not part of the product): mlp_chain2 on the flagship shapes -- schedule / stagger variants, pair overlap,
// per-phase cycle stamps, workgroup placement census.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I morl-baselines_amd/csrc tools/probes/chain2_probe.hip -o tools/probes/chain2_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "mlp_chain2.h"
using namespace morl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SCHED>
__global__ __launch_bounds__(CH_THREADS, 2) void chain2_prof_kernel(Chain2Multi m) {
    __shared__ __attribute__((aligned(16))) float sAct[C2_TM * C2_LDK + 8];
    mlp_chain2_persistent<SCHED, true>(m, sAct);
}
// one workgroup per CU allowed to use the whole register file? (no: same code, launch bounds 1 -> 512 VGPR budget)
template <int SCHED>
__global__ __launch_bounds__(CH_THREADS, 1) void chain2_lb1_kernel(Chain2Multi m) {
    __shared__ __attribute__((aligned(16))) float sAct[C2_TM * C2_LDK + 8];
    mlp_chain2_persistent<SCHED, false>(m, sAct);
}
__global__ void census_kernel(unsigned* out, long long* t) {
    if (threadIdx.x == 0) {
        out[blockIdx.x] = c2_cu_key();
        t[blockIdx.x] = wall_clock64();
    }
    // keep the workgroup resident for a while so that the whole grid is co-resident
    __shared__ float s[C2_TM * C2_LDK + 8];
    for (int k = 0; k < 200; ++k) { s[threadIdx.x] = (float)k; __syncthreads(); }
    if (s[threadIdx.x] < 0.f) out[0] = 0;
}

static const int B = 256, W = 64, D = 32, R = 3, rows = B * W;
static const int dims[6] = {35, 256, 256, 256, 256, 18};
static float *obs, *wv, *wt, *wt2, *params, *wbpad, *bias, *q[3], *hbuf, *gbuf, *x0m, *dq;
static unsigned long long* bits[5];
static unsigned* tickets;

static ChainArgs fwd_chain(const float* wtp, int row_order, bool save, float* qout) {
    ChainArgs a{};
    a.n_steps = 5; a.rows = rows; a.in_mode = 0; a.fast = 1; a.obs = obs; a.weights = wv; a.B = B; a.W = W; a.D = D; a.R = R;
    a.row_order = row_order;
    size_t off = 0, poff = 0;
    for (int l = 0; l < 5; ++l) {
        ChainStep& st = a.step[l];
        const int ldn = (dims[l + 1] + 3) / 4 * 4, kpad = (dims[l] + 63) / 64 * 64;
        st.Bmat = wtp + off; st.ldb = ldn; st.K = dims[l]; st.kpad = kpad; st.N = dims[l + 1]; st.bias = bias + l * 256; st.relu = l < 4;
        st.Bt = params + poff; st.ldbt = dims[l];
        off += (size_t)kpad * ldn;
        poff += (size_t)dims[l + 1] * dims[l] + dims[l + 1];
        if (l == 4) { st.out = qout; st.ldout = 18; }
        else if (save) { st.out = hbuf + (size_t)l * rows * 256; st.ldout = 256; st.bits_out = bits[l + 1]; }
    }
    if (save) { a.x0_out = x0m; a.ldx0 = 36; }
    return a;
}
static ChainArgs bwd_chain() {
    ChainArgs a{};
    a.n_steps = 4; a.rows = rows; a.in_mode = 1; a.fast = 1; a.src = dq; a.ldsrc = 20; a.K0 = 18;
    size_t poff[5]; size_t o = 0;
    for (int l = 0; l < 5; ++l) { poff[l] = o; o += (size_t)dims[l + 1] * dims[l] + dims[l + 1]; }
    for (int l = 4, k = 0; l >= 1; --l, ++k) {
        ChainStep& st = a.step[k];
        st.Bmat = (l == 4) ? wbpad : params + poff[l];
        st.ldb = dims[l]; st.K = dims[l + 1]; st.kpad = (dims[l + 1] + 63) / 64 * 64; st.N = dims[l];
        st.Bt = wt; st.ldbt = 256;
        st.bits_in = bits[l];
        st.out = gbuf + (size_t)(l - 1) * rows * 256; st.ldout = 256;
    }
    return a;
}

struct Cfg { int S, stagger, sched, lb1; int nmajor = 0; };

static Chain2Multi make_multi(const std::vector<ChainArgs>& ch, const Cfg& c) {
    Chain2Multi m{};
    m.n = (int)ch.size();
    int units = 0;
    for (int qn = 0; qn < m.n; ++qn) { m.p[qn] = ch[qn]; m.unit_start[qn] = units; units += (ch[qn].rows + 63) / 64; }
    for (int qn = m.n; qn <= CH_MAX_MULTI; ++qn) m.unit_start[qn] = units;
    const int S = std::min(c.S, 2 * units);
    m.full_rounds = units / S;
    m.tail_base = m.full_rounds * S;
    m.tail_units = units - m.tail_base;
    m.tail_halves = (2 * m.tail_units <= S) ? 1 : 0;
    m.stagger = c.stagger;
    m.cu_tickets = tickets;
    return m;
}

static void launch(const Chain2Multi& m, const Cfg& c, int S) {
    if (c.nmajor) {
        hipLaunchKernelGGL(mlp_chain2_n_kernel, dim3(S), dim3(CH_THREADS), 0, 0, m);
    } else if (c.lb1) {
        if (c.sched) hipLaunchKernelGGL(chain2_lb1_kernel<1>, dim3(S), dim3(CH_THREADS), 0, 0, m);
        else hipLaunchKernelGGL(chain2_lb1_kernel<0>, dim3(S), dim3(CH_THREADS), 0, 0, m);
    } else {
        if (c.sched) hipLaunchKernelGGL(mlp_chain2_kernel<1>, dim3(S), dim3(CH_THREADS), 0, 0, m);
        else hipLaunchKernelGGL(mlp_chain2_kernel<0>, dim3(S), dim3(CH_THREADS), 0, 0, m);
    }
}

static int time_cfg(const char* name, const std::vector<ChainArgs>& ch, const Cfg& c, double flop) {
    Chain2Multi m = make_multi(ch, c);
    int units = m.unit_start[m.n];
    const int S = std::min(c.S, 2 * units);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) launch(m, c, S);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 20);
    }
    CK(hipGetLastError());
    printf("%-44s S=%3d stagger=%d sched=%d lb1=%d rounds=%d tail=%d(%s)  %7.1f us  %6.1f TFLOP/s\n", name, S, c.stagger, c.sched, c.lb1,
           m.full_rounds, m.tail_units, m.tail_halves ? "halves" : "whole", best * 1e3, flop / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    CK(hipMalloc(&obs, B * D * 4)); CK(hipMalloc(&wv, W * R * 4));
    const size_t wt_floats = 64 * 256 + 3 * 65536 + 256 * 20, p_floats = 35 * 256 + 256 + 3 * (65536 + 256) + 18 * 256 + 18;
    CK(hipMalloc(&wt, wt_floats * 4 + 64)); CK(hipMalloc(&wt2, wt_floats * 4 + 64)); CK(hipMalloc(&params, p_floats * 4 + 64));
    CK(hipMalloc(&wbpad, 64 * 256 * 4)); CK(hipMalloc(&bias, 5 * 256 * 4));
    for (int k = 0; k < 3; ++k) CK(hipMalloc(&q[k], (size_t)rows * 20 * 4));
    CK(hipMalloc(&hbuf, (size_t)4 * rows * 256 * 4)); CK(hipMalloc(&gbuf, (size_t)4 * rows * 256 * 4));
    CK(hipMalloc(&x0m, (size_t)rows * 36 * 4)); CK(hipMalloc(&dq, (size_t)rows * 20 * 4));
    for (int l = 0; l < 5; ++l) { CK(hipMalloc(&bits[l], (size_t)rows / 64 * 256 * 8)); CK(hipMemset(bits[l], 0xff, (size_t)rows / 64 * 256 * 8)); }
    CK(hipMalloc(&tickets, C2_CU_SLOTS * 4)); CK(hipMemset(tickets, 0, C2_CU_SLOTS * 4));
    {
        std::vector<float> hw(wt_floats);
        for (size_t e = 0; e < hw.size(); ++e) hw[e] = 0.05f * (float)((int)(e * 2654435761u % 1000) - 500) / 500.f;
        for (int k = 35; k < 64; ++k) for (int n = 0; n < 256; ++n) hw[(size_t)k * 256 + n] = 0.f;   // layer 0 rows beyond K
        CK(hipMemcpy(wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(wt2, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hp(p_floats);
        for (size_t e = 0; e < hp.size(); ++e) hp[e] = 0.05f * (float)((int)(e * 40503u % 1000) - 500) / 500.f;
        CK(hipMemcpy(params, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hb(64 * 256, 0.f);
        for (int k = 0; k < 18; ++k) for (int n = 0; n < 256; ++n) hb[k * 256 + n] = 0.01f * (float)((k * 31 + n * 7) % 200 - 100) / 100.f;
        CK(hipMemcpy(wbpad, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> ho(B * D);
        for (size_t e = 0; e < ho.size(); ++e) ho[e] = (float)((int)(e * 40503u % 2000) - 1000) / 1000.f;
        CK(hipMemcpy(obs, ho.data(), ho.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hwv(W * R);
        for (size_t e = 0; e < hwv.size(); ++e) hwv[e] = (float)(e % 7) / 7.f;
        CK(hipMemcpy(wv, hwv.data(), hwv.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(bias, 0, 5 * 256 * 4));
        std::vector<float> hd((size_t)rows * 20);
        for (size_t e = 0; e < hd.size(); ++e) hd[e] = 1e-4f * (float)((int)(e * 2246822519u % 2000) - 1000) / 1000.f;
        CK(hipMemcpy(dq, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    }
    const double F_FWD = (double)rows * 420352.0, F_BWD = (double)rows * 402432.0;

    // ---- placement census: which workgroups share a CU ---------------------------------------------------------------
    {
        unsigned* d; long long* t; CK(hipMalloc(&d, 512 * 4)); CK(hipMalloc(&t, 512 * 8));
        hipLaunchKernelGGL(census_kernel, dim3(512), dim3(256), 0, 0, d, t);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(512); CK(hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, std::vector<int>> by;
        for (int b = 0; b < 512; ++b) by[h[b]].push_back(b);
        int hist[8] = {0};
        for (auto& kv : by) hist[std::min<size_t>(7, kv.second.size())]++;
        printf("census: %zu distinct CU keys; workgroups per key histogram:", by.size());
        for (int k = 1; k < 8; ++k) printf(" %d:%d", k, hist[k]);
        printf("\n  first keys:");
        int shown = 0;
        for (auto& kv : by) { if (shown++ >= 12) break; printf(" %03x{", kv.first); for (int b : kv.second) printf("%d ", b); printf("}"); }
        printf("\n  deltas between co-resident workgroup ids:");
        std::map<int, int> dh;
        for (auto& kv : by) for (size_t k = 1; k < kv.second.size(); ++k) dh[kv.second[k] - kv.second[k - 1]]++;
        for (auto& kv : dh) printf(" %d:x%d", kv.first, kv.second);
        printf("\n");
    }

    std::vector<ChainArgs> fwd3 = {fwd_chain(wt, 0, false, q[0]), fwd_chain(wt2, 0, false, q[1]), fwd_chain(wt, 1, true, q[2])};
    std::vector<ChainArgs> fwd2 = {fwd_chain(wt, 0, false, q[0]), fwd_chain(wt2, 0, false, q[1])};
    std::vector<ChainArgs> fwd1 = {fwd_chain(wt, 0, false, q[0])};
    std::vector<ChainArgs> fwd1s = {fwd_chain(wt, 1, true, q[2])};
    std::vector<ChainArgs> bwd = {bwd_chain()};

    for (int sched = 0; sched < 2; ++sched)
        for (int stg = 0; stg < 4; ++stg) time_cfg("forward x3 (production shape)", fwd3, Cfg{512, stg, sched, 0}, 3 * F_FWD);
    for (int sched = 0; sched < 2; ++sched) {
        if (sched) {
            // the same passes reading the nn.Linear matrices themselves (N-major weight stream, no shadow copy)
            auto nm = [](std::vector<ChainArgs> v) { for (auto& a : v) a.fast = 2; return v; };
            for (int stg : {0, 3}) time_cfg("forward x3, N-major stream", nm(fwd3), Cfg{512, stg, 1, 0, 1}, 3 * F_FWD);
            time_cfg("forward x1 no-save halves, N-major", nm(fwd1), Cfg{512, 0, 1, 0, 1}, F_FWD);
            time_cfg("forward x1 save halves, N-major", nm(fwd1s), Cfg{512, 0, 1, 0, 1}, F_FWD);
        }
        time_cfg("forward x2 no-save, 1 round, 2 WG/CU", fwd2, Cfg{512, 0, sched, 0}, 2 * F_FWD);
        time_cfg("forward x2 no-save, 2 rounds, 1 WG/CU", fwd2, Cfg{256, 0, sched, 0}, 2 * F_FWD);
        time_cfg("forward x2 no-save, 2 rounds, 1 WG/CU lb1", fwd2, Cfg{256, 0, sched, 1}, 2 * F_FWD);
        time_cfg("forward x1 no-save, 64-row, 1 WG/CU", fwd1, Cfg{256, 0, sched, 0}, F_FWD);
        time_cfg("forward x1 no-save, 32-row halves, 2 WG/CU", fwd1, Cfg{512, 0, sched, 0}, F_FWD);
        time_cfg("forward x1 save, 64-row, 1 WG/CU", fwd1s, Cfg{256, 0, sched, 0}, F_FWD);
        time_cfg("forward x1 save, 32-row halves, 2 WG/CU", fwd1s, Cfg{512, 0, sched, 0}, F_FWD);
        time_cfg("backward dX, 32-row halves, 2 WG/CU", bwd, Cfg{512, 0, sched, 0}, F_BWD);
        time_cfg("backward dX, 64-row, 1 WG/CU", bwd, Cfg{256, 0, sched, 0}, F_BWD);
        time_cfg("backward dX, 64-row, 1 WG/CU lb1", bwd, Cfg{256, 0, sched, 1}, F_BWD);
        time_cfg("backward dX, 128 slots 2 rounds", bwd, Cfg{128, 0, sched, 0}, F_BWD);
    }

    // ---- phase stamps: forward x3, production schedule ------------------------------------------------------------------
    for (int variant = 0; variant < 2; ++variant) {
        Cfg c{variant == 0 ? 512 : 256, variant == 0 ? 3 : 0, 1, 0};
        Chain2Multi m = make_multi(variant == 0 ? fwd3 : fwd1, c);
        long long* prof; CK(hipMalloc(&prof, (size_t)512 * 2 * 24 * 8)); CK(hipMemset(prof, 0, (size_t)512 * 2 * 24 * 8));
        m.prof = prof;
        const int S = c.S;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(chain2_prof_kernel<1>, dim3(S), dim3(CH_THREADS), 0, 0, m);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain2_prof_kernel<1>, dim3(S), dim3(CH_THREADS), 0, 0, m);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> hp((size_t)512 * 2 * 24);
        CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
        // per job: stamps 0 start, 1 input done, then per wide step: loop end, barrier, epilogue end (3 stamps), narrow step 1 stamp, final
        long long gmin = -1, gmax = -1;
        for (int b = 0; b < S; ++b) for (int j = 0; j < 2; ++j) {
            const long long* t = &hp[((size_t)b * 2 + j) * 24];
            if (t[0] == 0) continue;
            int last = 0; for (int k = 0; k < 24; ++k) if (t[k]) last = k;
            if (gmin < 0 || t[0] < gmin) gmin = t[0];
            if (t[last] > gmax) gmax = t[last];
        }
        printf("prof variant %d: launch %.1f us, cycle span %lld -> effective clock %.2f GHz\n", variant, ms * 1e3, gmax - gmin,
               (double)(gmax - gmin) / (ms * 1e6));
        for (int b : {0, 1, 255, 256, 257, 511}) {
            if (b >= S) continue;
            for (int j = 0; j < 2; ++j) {
                const long long* t = &hp[((size_t)b * 2 + j) * 24];
                if (t[0] == 0) continue;
                printf("  wg %3d job %d start@%7lld:", b, j, t[0] - gmin);
                for (int k = 1; k < 24 && t[k]; ++k) printf(" %lld", t[k] - t[k - 1]);
                printf("\n");
            }
        }
        CK(hipFree(prof));
    }
    return 0;
}
