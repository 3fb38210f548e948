// MFMA issue-rate microbenchmark for gfx950: SIMD cycles (2.4 GHz assumed) per instruction, 8 independent accumulators,
// 1 and 2 waves per SIMD, for the bf16 shapes a D=40 attention could mix.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[8]; f32x16 big[2];
    bf16x8 a8, b8; bf16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.01f + i); b8[i] = (__bf16)(0.5f - i); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (MODE == 0) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[m & 7], 0, 0, 0);
            if (MODE == 1) { union { bf16x4 b; s16x4 s; } ua, ub; ua.b = a4; ub.b = b4; acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ua.s, ub.s, acc[m & 7], 0, 0, 0); }
            if (MODE == 2) big[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, big[m & 1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 2; ++i) s += big[i][0] + big[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double flop) {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 32 * wps;                       // instructions per SIMD
        printf("%-28s waves/SIMD %d: %6.2f cycles/instr @2.4 GHz  -> %7.1f TFLOP/s chip-wide\n", name, wps, ms * 1e-3 * 2.4e9 / n,
               flop * n * 1024 / (ms * 1e-3) / 1e12);
    }
    (void)hipFree(d);
}
int main() {
    run<0>("mfma_f32_16x16x32_bf16", 16384.0);
    run<1>("mfma_f32_16x16x16_bf16_1k", 8192.0);
    run<2>("mfma_f32_32x32x16_bf16", 32768.0);
    return 0;
}
