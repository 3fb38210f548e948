// MFMA 32x32x16 x VALU overlap inside one wave, with the D = 40 attention's per-key-block mix (round 2): 28 MFMAs + 64 v_exp_f32 + 64 v_fma_f32 +
// 32 v_max3-like + 32 packs per iteration, blocked (all MFMAs, then all VALU) or interleaved (1 MFMA : ~7 VALU), two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mix32.hip -o mix32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>     // 0 MFMA only, 1 VALU only, 2 blocked, 3 interleaved
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float c) {
    f32x16 acc[4];
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(0.5f - i); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float e[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) e[i] = threadIdx.x * 0.001f + i * 0.01f;
    float mx = 0.f;
    unsigned pk = 0;
    auto valu_slice = [&](int v) __attribute__((always_inline)) {      // v = 0..31: 2 fma, 2 exp, 1 max3, 1 pack
        const int i0 = (2 * v) & 31, i1 = (2 * v + 1) & 31;
        e[i0] = __builtin_amdgcn_exp2f(fmaf(e[i0], c, -0.25f));
        e[i1] = __builtin_amdgcn_exp2f(fmaf(e[i1], c, -0.25f));
        mx = fmaxf(fmaxf(mx, e[i0]), e[i1]);
        unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(e[i0]), "v"(e[i1]));
        pk ^= r;
    };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int m = 0; m < 28; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int v = 0; v < 32; ++v) valu_slice(v);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 28; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
                valu_slice(m);
                if (m < 4) valu_slice(28 + m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = mx + (float)pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, 10, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // two waves per SIMD: cycles of SIMD time per wave-iteration (2.4 GHz nominal)
    printf("%-44s %7.0f cycles per wave-iteration (2 waves/SIMD: %7.0f per SIMD pair-iteration)\n", name, ms * 1e-3 * 2.4e9 / iters / 2, ms * 1e-3 * 2.4e9 / iters);
    (void)hipFree(d);
}
int main() {
    run<0>("28 MFMA 32x32x16 only");
    run<1>("64 exp + 64 fma + 32 max + 32 pack only");
    run<2>("both, blocked");
    run<3>("both, interleaved 1 MFMA : ~7 VALU");
    return 0;
}
