// hipsim runtime (TEST INFRASTRUCTURE ONLY) -- fiber scheduler for the emulated workgroup.
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <vector>

namespace hipsim {
LaneCtx* cur = nullptr;
dim3 cur_block_dim, cur_grid_dim;

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
    ucontext_t ctx;
    LaneCtx lane;
    char* stack = nullptr;
    bool done = false;
};
struct WaveState {
    int arrived = 0, nlanes = 0;
    unsigned long long gen = 0;
    CollIn in[64];
    CollOut out[64];
};
ucontext_t sched_ctx;
std::vector<Fiber> fibers;
std::vector<WaveState> waves;
int block_arrived = 0, block_live = 0;
unsigned long long block_gen = 0;
const std::function<void()>* body_fn = nullptr;
Fiber* cur_fiber = nullptr;

void yield() { swapcontext(&cur_fiber->ctx, &sched_ctx); }
void trampoline() {
    (*body_fn)();
    cur_fiber->done = true;
    // a finished work-item no longer takes part in barriers of its block (HIP: UB if others still wait;
    // here we simply shrink the rendezvous so well-formed early-exit kernels complete)
    --block_live;
    if (block_arrived > 0 && block_arrived >= block_live) { block_arrived = 0; ++block_gen; }
    WaveState& w = waves[cur_fiber->lane.wave];
    w.nlanes--;
    if (w.arrived > 0 && w.arrived >= w.nlanes) {
        std::fprintf(stderr, "hipsim: a lane exited while its wave waits in a collective\n");
        std::abort();
    }
    swapcontext(&cur_fiber->ctx, &sched_ctx);
}
}  // namespace

void block_barrier() {
    const unsigned long long g = block_gen;
    if (++block_arrived >= block_live) { block_arrived = 0; ++block_gen; return; }
    while (block_gen == g) yield();
}

CollOut wave_collective(const CollIn& in, void (*fn)(const CollIn*, CollOut*, int)) {
    WaveState& w = waves[cur->wave];
    const int lane = cur->lane;
    w.in[lane] = in;
    const unsigned long long g = w.gen;
    if (++w.arrived >= w.nlanes) {
        fn(w.in, w.out, w.nlanes);
        w.arrived = 0;
        ++w.gen;
    } else {
        while (w.gen == g) yield();
    }
    return w.out[lane];
}

static long long g_launches = 0;      // kernel launches since the library was loaded (tests count the launches of an update)

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
    ++g_launches;
    const int nthreads = (int)(block.x * block.y * block.z);
    cur_block_dim = block;
    cur_grid_dim = grid;
    body_fn = &body;
    if ((int)fibers.size() < nthreads) {
        size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = (char*)std::malloc(kStack);
    }
    const int nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                waves.assign(nwaves, WaveState());
                block_arrived = 0;
                block_live = nthreads;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = fibers[t];
                    f.done = false;
                    f.lane.tid = uint3{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                    f.lane.bid = uint3{bx, by, bz};
                    f.lane.lane = t & 63;
                    f.lane.wave = t >> 6;
                    waves[t >> 6].nlanes++;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &sched_ctx;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
                }
                int live = nthreads;
                while (live > 0) {
                    live = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        cur_fiber = &f;
                        cur = &f.lane;
                        swapcontext(&sched_ctx, &f.ctx);
                        if (!f.done) ++live;
                    }
                }
            }
    cur = nullptr;
}
}  // namespace hipsim

// test hook: how many kernels the emulated library has launched so far (the product library has no such symbol)
extern "C" long long hipsim_launch_count() { return hipsim::g_launches; }
