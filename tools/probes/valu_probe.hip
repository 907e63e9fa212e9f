// Development probe (not part of the product): issue cost of the vector-ALU instructions of the bf16 split, one wave per SIMD,
// independent streams (16 accumulators) and dependent chains.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/valu_probe.hip -o tools/probes/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP, bool DEP>
__global__ void k(long long* out, float seed, int iters) {
    float r[16]; f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = seed + i + threadIdx.x; p[i] = f2{seed + i, seed - i}; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = DEP ? 0 : i;
            if (OP == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[j]) : "v"(r[15]));
            if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(p[15]));
            if (OP == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[j]) : "v"(r[15]));
            if (OP == 3) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[j]));
            if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(r[j]));
            if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(r[j]) : "v"(r[15]));
            if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[j]) : "v"(p[15]));
            if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(r[15]), "v"(r[14]));
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[j]) : "v"(r[15]));
            if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(p[15]));
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int i = 0; i < 16; ++i) s += r[i] + p[i].x + p[i].y;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}
template <int OP> int run(const char* name, long long* d) {
    for (int dep = 0; dep < 2; ++dep)
        for (int waves : {1, 2, 4}) {     // waves per SIMD (blocks of 256 * waves work-items on one CU)
            long long h = 0;
            if (dep) hipLaunchKernelGGL((k<OP, true>), dim3(1), dim3(256 * waves), 0, 0, d, 1.5f, 1000);
            else hipLaunchKernelGGL((k<OP, false>), dim3(1), dim3(256 * waves), 0, 0, d, 1.5f, 1000);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
            printf("%-28s %s  %d wave(s)/SIMD: %6.2f cycles per instruction per wave\n", name, dep ? "dependent  " : "independent", waves, (double)h / 16000.0);
        }
    return 0;
}
int main() {
    long long* d; CK(hipMalloc(&d, 64));
    run<0>("v_cvt_pk_bf16_f32", d); run<1>("v_pk_add_f32", d); run<6>("v_pk_add_f32 neg", d); run<2>("v_sub_f32", d); run<3>("v_and_b32 (literal)", d);
    run<4>("v_lshlrev_b32", d); run<5>("v_mov_b32", d); run<7>("v_perm_b32", d); run<8>("v_fma_f32", d); run<9>("v_pk_mul_f32", d);
    return 0;
}
