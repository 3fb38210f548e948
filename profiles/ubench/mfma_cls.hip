// Does the register class of an MFMA's operands decide how much of the wave's own VALU work issues in its shadow?  (round 4)
// One wave per SIMD (256 threads, 100 KB of LDS per workgroup), 256 workgroups; per iteration 16 x { v_mfma_f32_32x32x16_bf16 ; NV filler VALU ops },
// four accumulator chains.  CD: accumulators in AGPRs (1) or arch VGPRs (0); AB: A / B fragments in AGPRs (1) or VGPRs (0).
// Fillers: FK 0 = v_fma_f32 (independent, rotating registers), 1 = v_exp_f32, 2 = one exp + (NV - 1) fma.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_cls.hip -o mfma_cls ; prints nominal cycles (2.4 GHz) per MFMA gap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int CD, int AB> __device__ __forceinline__ void mf(f32x16& c, i32x4 a, i32x4 b) {
    if constexpr (CD && AB) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "a"(b));
    else if constexpr (CD && !AB) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else if constexpr (!CD && AB) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int CD, int AB, int NV, int FK, int NOMFMA, int NCH>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float cc) {
    extern __shared__ char smem[];
    f32x16 acc[4];
    i32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80 + threadIdx.x; b[i] = 0x3f003f00 + i; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = threadIdx.x * 1e-3f + i * 0.01f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (!NOMFMA) mf<CD, AB>(acc[m & (NCH - 1)], a, b);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float& x = e[(m * NV + v) & 7];
                if ((FK == 1) || (FK == 2 && v == 0)) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(cc));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    for (int i = 0; i < 8; ++i) s += e[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + smem[threadIdx.x];
}
template <int CD, int AB, int NV, int FK, int NOMFMA = 0, int NCH = 4> void run(float* d) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<CD, AB, NV, FK, NOMFMA, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CD, AB, NV, FK, NOMFMA, NCH><<<256, 256, 100 * 1024>>>(d, 10, 1.0f);
    hipEventRecord(e0);
    k<CD, AB, NV, FK, NOMFMA, NCH><<<256, 256, 100 * 1024>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s chains=%d CD=%s AB=%s NV=%d filler=%s : %.1f cycles per gap\n", NOMFMA ? "no MFMA" : "MFMA   ", NCH, CD ? "acc" : "vgpr", AB ? "acc" : "vgpr", NV, FK == 0 ? "fma" : (FK == 1 ? "exp" : "exp+fma"),
           ms * 1e-3 * 2.4e9 / (iters * 16.0));
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<1, 1, 0, 0, 0, 4>(d); run<1, 1, 0, 0, 0, 2>(d); run<1, 1, 0, 0, 0, 1>(d); run<0, 1, 0, 0, 0, 2>(d); run<0, 1, 0, 0, 0, 1>(d);
    run<0, 1, 4, 2, 0, 4>(d); run<0, 1, 4, 2, 0, 2>(d); run<0, 1, 4, 2, 0, 1>(d); run<1, 1, 4, 2, 0, 2>(d);
    run<0, 1, 6, 2, 0, 2>(d); run<0, 1, 3, 2, 0, 2>(d); run<0, 1, 2, 2, 0, 2>(d);
    return 0;
}
