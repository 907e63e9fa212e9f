// Shared device helpers for the gfx950 kernels of libmorl_hip (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace morl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int c2_u32x4 __attribute__((ext_vector_type(4)));   // payload type of the 16-byte raw buffer builtins

constexpr int kWave = 64;

// Exact-f32 matrix core op: D(32x32) = A(32x2) * B(2x32) + C, one f32 per lane for A and B.
// Lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; D register r of lane l is
// row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), column l & 31.  v_mfma_f32_32x32x2_f32: 64 cycles / SIMD.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// wave index as a scalar (SGPR) value so that wave-dependent loop bounds / branches stay scalar
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// The first BYTES of the kernel's argument block pulled into the scalar cache in (a few batches of) ONE round trip, at kernel
// entry.  A kernel whose prologue reads a large by-value argument block in several DEPENDENT batches -- pointer, then what hangs
// off it, branch, next pointer -- otherwise pays a first-touch miss of the scalar cache per batch (0.3 - 0.5 us each on MI355X;
// the 8-row chain's prologue: 3.6 -> 2.9 us with this and branch-free fetches, profiles/r05_target_rows_phases.txt).
// The 64-bit Dropout stream seed folded to the 32-bit state of the counter hash (ac_kernels.h: ac_uniform, mlp_chain16.h: c16_uniform):
// both words through a hash round of their own BEFORE the counter's Weyl step is added (a plain xor of the words made every pair of seeds
// with hi ^ lo equal share one stream).  Uniform per launch: the compiler keeps it in scalar registers, outside the element loops.
__device__ __forceinline__ unsigned int dropout_seed_mix(unsigned long long seed) {
    unsigned int s = (unsigned int)seed ^ 0x9E3779B9u;
    s ^= s >> 16; s *= 0x7FEB352Du;
    s ^= s >> 15; s *= 0x846CA68Bu;
    s ^= s >> 16;
    unsigned int h = (unsigned int)(seed >> 32) + 0x85EBCA6Bu;
    h ^= h >> 16; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    return s + (h | 1u) * 0xC2B2AE35u;
}

template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if defined(__AMDGCN__)
    const __attribute__((address_space(4))) int* ka = (const __attribute__((address_space(4))) int*)__builtin_amdgcn_kernarg_segment_ptr();
    int t = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) t |= ka[i * 16];
    asm volatile("" ::"s"(t));
#endif
}

// the same for BYTES at a (workgroup-uniform) byte offset of a TOTAL-byte block: the one chain of a multi-chain argument block a
// workgroup reads (lines clamped to the block: nothing beyond the argument segment is touched)
template <int BYTES, int TOTAL>
__device__ __forceinline__ void kernarg_warm_at(int byte_offset) {
#if defined(__AMDGCN__)
    const __attribute__((address_space(4))) char* base = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    const int first = byte_offset & ~63, last = (TOTAL - 1) & ~63;
    int t = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 126) / 64; ++i) {
        const int off = min(first + 64 * i, last);
        t |= *(const __attribute__((address_space(4))) int*)(base + off);
    }
    asm volatile("" ::"s"(t));
#else
    (void)byte_offset;
#endif
}

// Deterministic butterfly sums (same order on every run; all lanes end with the total).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// XCD-aware block id: the dispatcher places block b on XCD b % 8; remap so that consecutive logical
// tiles (which share an operand panel) land on one XCD's L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

}  // namespace morl
