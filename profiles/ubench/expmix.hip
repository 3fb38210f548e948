// Can the full-rate VALU take over part of the softmax's exp2 work next to the (quarter-rate) transcendental unit?  Per iteration and lane: 64 exp2 of
// non-positive inputs, NP of them by a degree-4 polynomial (v_fract / v_sub / v_cvt / 4 fma / v_ldexp) and 64 - NP by v_exp_f32, interleaved; with and
// without 28 MFMA 32x32x16 alongside (the D = 40 attention's per-key-block mix).  Two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 expmix.hip -o expmix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ float exp2_poly(float x) {           // x <= 0
    const float f = __builtin_amdgcn_fractf(x);                  // x - floor(x) in [0, 1)
    const float n = x - f;
    float p = fmaf(f, 0.0135557f, 0.0520323f);                   // 2^f, degree 4 (max rel. error ~4e-6)
    p = fmaf(p, f, 0.2413793f);
    p = fmaf(p, f, 0.6930579f);
    p = fmaf(p, f, 1.0f);
    return __builtin_amdgcn_ldexpf(p, (int)n);
}
template <int NP, bool MFMA>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float c) {
    f32x16 acc[4];
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(0.5f - i); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float e[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) e[i] = -(threadIdx.x * 0.001f + i * 0.01f);
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < 64; ++v) {
            if (MFMA && v < 56 && (v & 1) == 0) acc[(v >> 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(v >> 1) & 3], 0, 0, 0);
            const bool poly = NP > 0 && (v * NP) / 64 != ((v + 1) * NP) / 64;
            const float r = poly ? exp2_poly(e[v]) : __builtin_amdgcn_exp2f(e[v]);
            s += r;
            e[v] = e[v] * c - r * 1e-6f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NP, bool MFMA> void run() {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NP, MFMA>), dim3(512), dim3(256), 0, 0, d, 10, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NP, MFMA>), dim3(512), dim3(256), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%2d of 64 exp2 by polynomial, %s: %6.0f nominal cycles per wave-iteration\n", NP, MFMA ? "with 28 MFMA" : "no MFMA     ", ms * 1e-3 * 2.4e9 / iters / 2);
    (void)hipFree(d);
}
int main() {
    run<0, false>(); run<16, false>(); run<24, false>(); run<32, false>(); run<64, false>();
    run<0, true>(); run<16, true>(); run<24, true>(); run<32, true>();
    return 0;
}
