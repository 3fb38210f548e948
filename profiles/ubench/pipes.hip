// Round 3: can ONE SIMD run a dense (throughput-bound) VALU stream of one wave beside back-to-back MFMAs of another wave?
// 512-thread workgroups, one per CU: waves 0-3 loop over 28 x v_mfma_f32_32x32x16_bf16 (4 accumulators), waves 4-7 over a softmax-like block of
// INDEPENDENT ops (64 v_exp_f32, 64 v_fma_f32, 32 v_max3-like, 32 v_cvt_pk) — unlike mix32.hip / antiphase.hip, no value chains through the block, so the
// VALU side is bound by issue / transcendental throughput, not by latency.  Modes: MFMA waves only, VALU waves only (the others exit), both.
// Build: hipcc --offload-arch=gfx950 -O3 pipes.hip -o pipes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int DATA>      // MODE 1: MFMA waves, 2: VALU waves, 3: both;  DATA 0: constant operands, 1: per-lane pseudo-random
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float c) {
    const int g = threadIdx.x >> 8;
    if (g == 0) {
        if (!(MODE & 1)) return;
        f32x16 acc[4];
        bf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u);
            a[i] = DATA ? (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 16384.0f)) : (__bf16)1.0f;
            b[i] = DATA ? (__bf16)(((int)((h >> 16) & 0xffff) - 32768) * (1.0f / 16384.0f)) : (__bf16)0.5f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 28; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        if (!(MODE & 2)) return;
        float e[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) e[i] = threadIdx.x * 0.001f + i * 0.01f;
        float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) e[i] = __builtin_amdgcn_exp2f(fmaf(e[i], c, -0.25f));          // 64 independent fma + exp
#pragma unroll
            for (int i = 0; i < 32; ++i) {                                                                // 8 independent chains of 4
                mx[i & 7] = fmaxf(fmaxf(mx[i & 7], e[2 * i]), e[2 * i + 1]);
                unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(e[2 * i]), "v"(e[2 * i + 1]));
                pk[i & 7] ^= r;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += mx[i] + (float)pk[i];
#pragma unroll
        for (int i = 0; i < 64; ++i) s += e[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}
template <int MODE, int DATA> void run(const char* name) {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, DATA>), dim3(256), dim3(512), 0, 0, d, 10, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, DATA>), dim3(256), dim3(512), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.1f ns per iteration (%6.0f nominal 2.4 GHz cycles)\n", name, ms * 1e6 / iters, ms * 1e-3 * 2.4e9 / iters);
    (void)hipFree(d);
}
int main() {
    run<1, 0>("MFMA wave alone (28 MFMA 32x32x16 / iteration), constant data");
    run<1, 1>("MFMA wave alone, pseudo-random data");
    run<2, 0>("VALU wave alone (64 exp + 64 fma + 32 max3 + 32 pack, independent)");
    run<3, 0>("both on each SIMD, free-running, constant data");
    run<3, 1>("both on each SIMD, free-running, pseudo-random data");
    return 0;
}
