// Device-resident replay storage: batch gather and the prioritized-replay sum tree.
#pragma once
#include "morl_device.h"

namespace morl {

// sum-tree levels are concatenated root first: level l has 2^l nodes at offset 2^l - 1
__device__ __forceinline__ long long level_off(int l) { return (1ll << l) - 1; }

// ----------------------------------------------------------------------------------------------
// Batch gather (ReplayBuffer.sample's five fancy-index gathers, common/buffer.py:82-91).
// Device storage is one AoS record per transition:  obs[D] | next_obs[D] | reward[R] | done | action[Ad]
// (all fp32; record_floats = 2D+R+1+Ad) so that add() is ONE contiguous H2D copy and a sampled transition
// is ONE contiguous read.  One wave per sampled transition, lanes stride the record (coalesced 4-B
// loads; 272 B per record at D=32, R=3).  HBM-bound: B * record_floats * 4 bytes in, the same out.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_batch_kernel(const float* __restrict__ records, int record_floats,
                                                           long long capacity, const int64_t* __restrict__ idx, int B,
                                                           int D, int R, int Ad, float* __restrict__ obs,
                                                           float* __restrict__ next_obs, float* __restrict__ rewards,
                                                           float* __restrict__ dones, float* __restrict__ actions_f,
                                                           int32_t* __restrict__ actions_i) {
    const int waves_per_block = (int)blockDim.x / kWave;
    const int lane = lane_id();
    for (int b = (int)blockIdx.x * waves_per_block + wave_id(); b < B; b += (int)gridDim.x * waves_per_block) {
        long long t = idx[b];
        if (t < 0) t = 0;
        if (t >= capacity) t = capacity - 1;
        const float* rec = records + (size_t)t * record_floats;
        for (int e = lane; e < record_floats; e += kWave) {
            const float v = rec[e];
            if (e < D) obs[(size_t)b * D + e] = v;
            else if (e < 2 * D) next_obs[(size_t)b * D + (e - D)] = v;
            else if (e < 2 * D + R) rewards[(size_t)b * R + (e - 2 * D)] = v;
            else if (e == 2 * D + R) dones[b] = v;
            else {
                const int a = e - (2 * D + R + 1);
                if (actions_f) actions_f[(size_t)b * Ad + a] = v;
                if (actions_i) actions_i[(size_t)b * Ad + a] = (int32_t)v;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Index selection + gather of one training batch in ONE launch (round 1 ran a sum-tree descent kernel, a gather kernel and
// two host->device copies for the B uniforms and the W x R sampled weights back to back: four dependent ~5 us launches).
// One wave per sampled transition: lane 0 walks the sum tree with its uniform (tree != NULL; SumTree.sample,
// prioritized_buffer.py:30-54, float64 in the reference's order) or takes the host-drawn index (ReplayBuffer.sample,
// buffer.py:80), broadcasts it, and the lanes stride the record.  `u01` / `idx_in` / `aux_src` may be PINNED HOST memory
// (mapped into the device's address space): a few KB read over PCIe inside the kernel instead of separate copy launches.
// Workgroup 0 also moves `aux_floats` floats from aux_src to aux_dst (the step's sampled weight vectors).
// ----------------------------------------------------------------------------------------------
struct SampleGatherArgs {
    const double* tree;        // NULL: indices come from idx_in
    const double* u01;
    const int64_t* idx_in;
    const float* records;
    float *obs, *next_obs, *rewards, *dones, *actions_f;
    int32_t* actions_i;
    int64_t* idx_out;
    const float* aux_src;
    float* aux_dst;
    long long capacity;
    int n_levels, record_floats, B, D, R, Ad, aux_floats;
};

// workgroup `block` of `n_blocks` (so that the routine can run as its own launch or as the leading workgroups of the step's
// prologue launch, beside the shadow-weight tiles)
__device__ __forceinline__ void sample_gather_body(const SampleGatherArgs& a, int block, int n_blocks) {
    const int waves_per_block = (int)blockDim.x / kWave;
    const int lane = lane_id();
    const int D = a.D, R = a.R, Ad = a.Ad;
    if (block == 0 && a.aux_src != nullptr)
        for (int e = (int)threadIdx.x; e < a.aux_floats; e += (int)blockDim.x) a.aux_dst[e] = a.aux_src[e];
    for (int b = block * waves_per_block + wave_id(); b < a.B; b += n_blocks * waves_per_block) {
        long long t = 0;
        if (a.tree != nullptr) {
            // SumTree.sample's descent (prioritized_buffer.py:30-54), five levels per memory round trip: the wave fetches the
            // 62 nodes of the subtree below the current node in ONE load (lane (2^j - 2) + i holds node i of relative level j)
            // and walks it with shuffles -- 4 round trips for a 100 000-leaf tree instead of 17 dependent loads by one lane.
            // Same comparisons and the same fp64 subtractions in the same order: indices stay bit-exact.
            double q = __dadd_rn(0.0, __dmul_rn(__dsub_rn(a.tree[0], 0.0), a.u01[b]));     // (wave-uniform)
            long long node = 0;
            const int ln = lane + 2;
            int j = 1;
            while ((2 << j) <= ln) ++j;
            const int i = ln - (1 << j);
            for (int l = 0; l < a.n_levels - 1;) {
                const int depth = min(5, a.n_levels - 1 - l);
                double v = 0.0;
                if (j <= depth) v = a.tree[level_off(l + j) + (node << j) + i];
                int loc = 0;
                for (int jj = 1; jj <= depth; ++jj) {
                    const double left = __shfl(v, (1 << jj) - 2 + 2 * loc);
                    const bool gt = q > left;
                    loc = 2 * loc + (gt ? 1 : 0);
                    q = __dsub_rn(q, __dmul_rn(left, gt ? 1.0 : 0.0));
                }
                node = (node << depth) + loc;
                l += depth;
            }
            t = node;
        } else {
            t = a.idx_in[b];
        }
        if (lane == 0 && a.idx_out) a.idx_out[b] = t;
        if (t < 0) t = 0;
        if (t >= a.capacity) t = a.capacity - 1;
        const float* rec = a.records + (size_t)t * a.record_floats;
        for (int e = lane; e < a.record_floats; e += kWave) {
            const float v = rec[e];
            if (e < D) a.obs[(size_t)b * D + e] = v;
            else if (e < 2 * D) a.next_obs[(size_t)b * D + (e - D)] = v;
            else if (e < 2 * D + R) a.rewards[(size_t)b * R + (e - 2 * D)] = v;
            else if (e == 2 * D + R) a.dones[b] = v;
            else {
                const int k = e - (2 * D + R + 1);
                if (a.actions_f) a.actions_f[(size_t)b * Ad + k] = v;
                if (a.actions_i) a.actions_i[(size_t)b * Ad + k] = (int32_t)v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sample_gather_kernel(SampleGatherArgs a) { sample_gather_body(a, (int)blockIdx.x, (int)gridDim.x); }

// Generic form for records with extra per-transition fields (CAPQL's ReplayMemory stores the weight vector of the
// episode with every transition, capql.py:40-54): field f of sampled row b = record[idx[b]][offset_f .. +width_f).
constexpr int GATHER_MAX_FIELDS = 8;
struct GatherFields {
    int n;
    int offset[GATHER_MAX_FIELDS];
    int width[GATHER_MAX_FIELDS];
    float* dst[GATHER_MAX_FIELDS];
};

__global__ __launch_bounds__(256) void gather_fields_kernel(const float* __restrict__ records, int record_floats,
                                                            long long capacity, const int64_t* __restrict__ idx, int B,
                                                            GatherFields f) {
    const int waves_per_block = (int)blockDim.x / kWave;
    const int lane = lane_id();
    for (int b = (int)blockIdx.x * waves_per_block + wave_id(); b < B; b += (int)gridDim.x * waves_per_block) {
        long long t = idx[b];
        if (t < 0) t = 0;
        if (t >= capacity) t = capacity - 1;
        const float* rec = records + (size_t)t * record_floats;
        for (int e = lane; e < record_floats; e += kWave) {
            const float v = rec[e];
#pragma unroll
            for (int k = 0; k < GATHER_MAX_FIELDS; ++k)
                if (k < f.n && e >= f.offset[k] && e < f.offset[k] + f.width[k])
                    f.dst[k][(size_t)b * f.width[k] + (e - f.offset[k])] = v;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Sum tree (common/prioritized_buffer.py:12-82), float64 levels concatenated root first: level l has
// 2^l nodes at offset 2^l - 1.  All arithmetic is IEEE float64 in the reference's order, so indices and
// node values are bit-exact.
// ----------------------------------------------------------------------------------------------

// SumTree.sample (:30-54): query = 0 + (root - 0) * u ; per level: go right iff query > left_sum.
__global__ __launch_bounds__(256) void sumtree_sample_kernel(const double* __restrict__ tree, int n_levels,
                                                             const double* __restrict__ u01, int B,
                                                             int64_t* __restrict__ idx) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= B) return;
    double q = __dadd_rn(0.0, __dmul_rn(__dsub_rn(tree[0], 0.0), u01[k]));
    long long node = 0;
    for (int l = 1; l < n_levels; ++l) {
        node *= 2;
        const double left = tree[level_off(l) + node];
        const bool gt = q > left;
        node += gt ? 1 : 0;
        q = __dsub_rn(q, __dmul_rn(left, gt ? 1.0 : 0.0));
    }
    idx[k] = node;
}

// Sequential SumTree.set (:56-67) for the transitions appended since the last sample.  value < 0 means
// "the running max priority" (PrioritizedReplayBuffer.add uses self.min_priority, :143).
// One wave; lane l owns level (n_levels-1-l): the n sets are applied in order, each lane adding the same
// diff to its level's ancestor -- exactly np.add.at(nodes, node_index, diff) per level.
__global__ __launch_bounds__(64) void sumtree_set_kernel(double* __restrict__ tree, int n_levels,
                                                         const int64_t* __restrict__ ptr,
                                                         const double* __restrict__ value, int n,
                                                         const double* __restrict__ running_max) {
    const int lane = lane_id();
    const int leaf_level = n_levels - 1;
    for (int k = 0; k < n; ++k) {
        const long long leaf = ptr[k];
        const double newp = (value == nullptr || value[k] < 0.0) ? *running_max : value[k];
        const double diff = __dsub_rn(newp, tree[level_off(leaf_level) + leaf]);
        // all lanes read the leaf before any lane writes it
        const double d = __shfl(diff, 0);
        if (lane < n_levels) {
            const int l = leaf_level - lane;
            const long long node = leaf >> lane;
            tree[level_off(l) + node] = __dadd_rn(tree[level_off(l) + node], d);
        }
    }
}

// PrioritizedReplayBuffer.update_priorities (:187-195) + SumTree.batch_set (:69-82), one workgroup.
//   pr[k]  = powf(raw[k] + (float)running_max, alpha)     (envelope.py:333, numpy float32 arithmetic)
//   running_max = max(running_max, max_k pr[k])
//   unique indices ascending, first occurrence's priority; diff = pr - leaf; per level add diffs of the
//   children in ascending index order (np.add.at order) -> bit-exact float64 tree.
constexpr int ST_MAX_B = 1024;
constexpr int ST_THREADS = 1024;   // 16 waves: the tree levels are updated concurrently, one wave per level
constexpr int ST_LDS_BYTES = ST_MAX_B * (8 + 8 + 8 + 4) + 64 * 4;   // scratch of sumtree_update_body

struct SumTreeUpdate {
    double* tree;
    const int64_t* idx;
    const float* raw;
    double* running_max;
    double* pr_out;          // or NULL
    int n_levels, B;
    float alpha;
    float clamp_min;         // > 0 (with alpha >= 0): priority = max(raw, clamp_min) ** alpha (GPIPD.update, gpi_pd.py:507-526)
};

// The update as a workgroup-level routine (any block size that is a multiple of 64; `lds` = ST_LDS_BYTES of scratch, 8-byte
// aligned), so that it can run as its own launch or ride as an extra workgroup of another kernel of the step
// (dw_tiles_kernel: the tree update only depends on the TD priorities and nothing of the backward pass depends on it).
// part 1: priorities, running maximum, sort, leaf differences -- leaves s_diff / s_node in LDS (see the offsets below), synchronised
__device__ __forceinline__ void sumtree_prepare(const SumTreeUpdate& a, void* lds) {
    long long* s_key = reinterpret_cast<long long*>(lds);   // (index << 11) | position -> ascending index, first position first
    double* s_diff = reinterpret_cast<double*>(s_key + ST_MAX_B);
    long long* s_node = reinterpret_cast<long long*>(s_diff + ST_MAX_B);
    float* s_pr = reinterpret_cast<float*>(s_node + ST_MAX_B);
    float* s_max = s_pr + ST_MAX_B;
    double* __restrict__ tree = a.tree;
    const int n_levels = a.n_levels, B = a.B;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    const float rmax = (float)(*a.running_max);
    float lmax = -INFINITY;
    for (int k = tid; k < B; k += nt) {
        // alpha < 0: raw already is the priority (plain PrioritizedReplayBuffer.update_priorities); clamp_min > 0: the GPI-PD form
        const float p = (a.alpha < 0.f) ? a.raw[k] : (a.clamp_min > 0.f) ? powf(fmaxf(a.raw[k], a.clamp_min), a.alpha)
                                                                          : powf(__fadd_rn(a.raw[k], rmax), a.alpha);
        s_pr[k] = p;
        if (a.pr_out) a.pr_out[k] = (double)p;
        lmax = fmaxf(lmax, p);
        s_key[k] = (a.idx[k] << 11) | (long long)k;
    }
    lmax = wave_max(lmax);
    if (lane_id() == 0) s_max[wave_id()] = lmax;
    __syncthreads();
    if (tid == 0) {
        float m = s_max[0];
        for (int w = 1; w < nt / 64; ++w) m = fmaxf(m, s_max[w]);
        // python max(self.min_priority, priorities.max()): keeps the old value unless the new one is larger
        if ((double)m > *a.running_max) *a.running_max = (double)m;
    }
    __syncthreads();
    // rank sort (keys are unique: they embed the position): rank = number of smaller keys, one pass over LDS broadcasts.
    // Ranks first (all reads), then the scatter (all writes).
    long long my_key[ST_MAX_B / 64 > 4 ? 4 : ST_MAX_B / 64];     // up to 4 keys per thread (B <= 4 * blockDim)
    int my_rank[4];
    int n_mine = 0;
    for (int k = tid; k < B && n_mine < 4; k += nt, ++n_mine) {
        const long long key = s_key[k];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (s_key[j] < key) ? 1 : 0;
        my_key[n_mine] = key;
        my_rank[n_mine] = rank;
    }
    __syncthreads();
    for (int q = 0; q < n_mine; ++q) s_key[my_rank[q]] = my_key[q];
    __syncthreads();
    // leaf diffs: the first occurrence of an index carries (priority - leaf), its duplicates 0.0 (x + 0.0 == x), and
    // EVERY entry keeps its node id so that s_node is non-decreasing (run boundaries by comparison / binary search)
    const int leaf_level = n_levels - 1;
    for (int k = tid; k < B; k += nt) {
        const long long id = s_key[k] >> 11;
        const bool first = (k == 0) || ((s_key[k - 1] >> 11) != id);
        s_node[k] = id;
        s_diff[k] = first ? __dsub_rn((double)s_pr[(int)(s_key[k] & 2047)], tree[level_off(leaf_level) + id]) : 0.0;
    }
    __syncthreads();
}

// part 2: per level, the head of each run of equal ancestors adds that run's diffs in order (the np.add.at order).  Levels
// touch disjoint memory, so each wave takes its own levels and they proceed concurrently; the critical path is the
// root's single run of B sequential float64 adds, whose LDS reads do not depend on the running sum and pipeline.
__device__ __forceinline__ void sumtree_levels(const SumTreeUpdate& a, const void* lds) {
    const double* s_diff = reinterpret_cast<const double*>(reinterpret_cast<const long long*>(lds) + ST_MAX_B);
    const long long* s_node = reinterpret_cast<const long long*>(s_diff + ST_MAX_B);
    double* __restrict__ tree = a.tree;
    const int n_levels = a.n_levels, B = a.B, leaf_level = n_levels - 1;
    const int nt = (int)blockDim.x;
    // (round 6: the (level, entry) pairs spread over ALL work-items, the root's level first -- one wave per level put the 256 run heads of
    // the leaf level through 64 lanes four at a time, each a dependent global load -> add -> store, and gave two waves two levels: 12 us of
    // the sharded step's clip + Adam launch.  A run is still summed by ONE work-item in entry order: the tree stays bit-exact.)
    // Four pairs per work-item at a time: run boundaries of all four first (LDS), then their four tree loads back to back, then the adds
    // and stores -- one global round trip per batch instead of one per run head.
    const int tid = (int)threadIdx.x, items = n_levels * B;
    constexpr int NB = 4;
    for (int it0 = tid; it0 < items; it0 += NB * nt) {
        bool head[NB];
        int beg[NB], end[NB];
        long long addr[NB];
        double acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int it = it0 + j * nt;
            head[j] = false; beg[j] = 0; end[j] = 0; addr[j] = 0;
            if (it < items) {
                const int up = n_levels - 1 - it / B, k = it % B;
                const long long anc = s_node[k] >> up;
                head[j] = (k == 0) || ((s_node[k - 1] >> up) != anc);
                if (head[j]) {
                    int lo = k + 1, hi = B;                        // first index whose ancestor differs (binary search)
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if ((s_node[mid] >> up) == anc) lo = mid + 1; else hi = mid;
                    }
                    beg[j] = k; end[j] = lo;
                    addr[j] = level_off(leaf_level - up) + anc;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = head[j] ? tree[addr[j]] : 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (head[j]) {
                double x = acc[j];
                int e = beg[j];
                for (; e + 8 <= end[j]; e += 8) {
                    double d[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) d[u] = s_diff[e + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x = __dadd_rn(x, d[u]);
                }
                for (; e < end[j]; ++e) x = __dadd_rn(x, s_diff[e]);
                tree[addr[j]] = x;
            }
    }
}

__device__ __forceinline__ void sumtree_update_body(const SumTreeUpdate& a, void* lds) {
    sumtree_prepare(a, lds);
    sumtree_levels(a, lds);
}

// The same update split over TWO launches (the sharded Envelope step: the priorities are complete only behind the all-reduce, where two
// short launches follow -- the sum of squares of the reduced gradient, clip + Adam -- and one workgroup carrying all 17 us of the update
// made the second of them 17 us long): part 1 as an extra workgroup of the first launch, its result (ST_MAX_B differences + node ids,
// 16 KB) through device memory, part 2 as an extra workgroup of the second.  Same arithmetic in the same order: the tree stays bit-exact.
constexpr int ST_SCRATCH_BYTES = 2 * ST_MAX_B * 8;
__device__ __forceinline__ void sumtree_update_part1(const SumTreeUpdate& a, void* lds, void* scratch) {
    sumtree_prepare(a, lds);
    const long long* src = reinterpret_cast<const long long*>(lds) + ST_MAX_B;        // s_diff then s_node: 2 * ST_MAX_B eight-byte words
    long long* dst = reinterpret_cast<long long*>(scratch);
    for (int e = (int)threadIdx.x; e < 2 * ST_MAX_B; e += (int)blockDim.x) dst[e] = src[e];
}
__device__ __forceinline__ void sumtree_update_part2(const SumTreeUpdate& a, void* lds, const void* scratch) {
    long long* dst = reinterpret_cast<long long*>(lds) + ST_MAX_B;
    const long long* src = reinterpret_cast<const long long*>(scratch);
    for (int e = (int)threadIdx.x; e < 2 * ST_MAX_B; e += (int)blockDim.x) dst[e] = src[e];
    __syncthreads();
    sumtree_levels(a, lds);
}

__global__ __launch_bounds__(ST_THREADS) void sumtree_update_kernel(SumTreeUpdate a) {
    __shared__ __attribute__((aligned(8))) unsigned char lds[ST_LDS_BYTES];
    sumtree_update_body(a, lds);
}

}  // namespace morl
