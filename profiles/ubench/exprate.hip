// Round 3: transcendental issue rates on one SIMD (2 waves): v_exp_f32 vs v_exp_f16 vs v_rcp_f32.
// Build: hipcc --offload-arch=gfx950 -O3 exprate.hip -o exprate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    float e[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) e[i] = -(threadIdx.x * 0.001f + i * 0.01f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE == 0) e[i] = __builtin_amdgcn_exp2f(e[i]) - 1.5f;
            if (MODE == 2) { asm volatile("v_exp_f16 %0, %0" : "+v"(e[i])); }                      // raw: f16 exp on the low half, no conversions
            if (MODE == 3) { asm volatile("v_exp_f32 %0, %0" : "+v"(e[i])); }                      // raw f32 exp, no other ops
            if (MODE == 4) { asm volatile("v_rcp_f32 %0, %0" : "+v"(e[i])); }
            if (MODE == 5) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[i])); }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 2 waves per SIMD, 32 ops per iteration per wave
    printf("%-52s %6.2f ns per wave-instruction per SIMD (%5.1f nominal 2.4 GHz cycles)\n", name, ms * 1e6 / iters / 64, ms * 1e-3 * 2.4e9 / iters / 64);
    (void)hipFree(d);
}
int main() {
    run<3>("v_exp_f32 (raw, 2 waves / SIMD)");
    run<2>("v_exp_f16 (raw)");
    run<4>("v_rcp_f32 (raw)");
    run<5>("v_fma_f32 (raw)");
    run<0>("exp2f + sub");
    return 0;
}
