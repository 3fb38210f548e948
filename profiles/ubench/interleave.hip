// Round 3: how much INDEPENDENT VALU work hides behind a wave's own MFMAs?  Per iteration 28 x v_mfma_f32_32x32x16_bf16 and N slices of
// {v_fma_f32 + v_exp_f32, v_fma_f32 + v_exp_f32, v_max3_f32, v_cvt_pk_bf16_f32} on independent registers (no chains through the block), either
// blocked (all MFMAs, then all VALU) or interleaved (PER slices after every MFMA, order pinned with sched_barrier), at one or two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 interleave.hip -o interleave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int WPS>      // MODE 0 blocked, 1 interleaved (1 slice per MFMA + the rest at the end), 2 MFMA only, 3 VALU only
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, float c) {
    f32x16 acc[4];
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u); a[i] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 16384.0f)); b[i] = (__bf16)(((int)((h >> 16) & 0xffff) - 32768) * (1.0f / 16384.0f)); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float e[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) e[i] = -(threadIdx.x * 0.001f + i * 0.01f);
    float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto slice = [&](int v) __attribute__((always_inline)) {      // v = 0..31
        e[2 * v] = __builtin_amdgcn_exp2f(fmaf(e[2 * v], c, -0.25f));
        e[2 * v + 1] = __builtin_amdgcn_exp2f(fmaf(e[2 * v + 1], c, -0.25f));
        mx[v & 7] = fmaxf(fmaxf(mx[v & 7], e[2 * v]), e[2 * v + 1]);
        unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(e[2 * v]), "v"(e[2 * v + 1]));
        pk[v & 7] ^= r;
    };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int m = 0; m < 28; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int v = 0; v < 32; ++v) slice(v);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int m = 0; m < 28; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
                slice(m);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int v = 28; v < 32; ++v) slice(v);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += mx[i] + (float)pk[i];
#pragma unroll
    for (int i = 0; i < 64; ++i) s += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int WPS> void run(const char* name) {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    const int iters = 4000, blocks = 256 * WPS;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %d wave(s)/SIMD: %7.0f nominal cycles of SIMD time per wave-iteration\n", name, WPS, ms * 1e-3 * 2.4e9 / iters / WPS);
    (void)hipFree(d);
}
int main() {
    run<2, 1>("28 MFMA only"); run<3, 1>("32 VALU slices only (64 exp, 64 fma, 32 max3, 32 pack)"); run<0, 1>("blocked"); run<1, 1>("interleaved, 1 slice per MFMA");
    run<2, 2>("28 MFMA only"); run<3, 2>("32 VALU slices only"); run<0, 2>("blocked"); run<1, 2>("interleaved, 1 slice per MFMA");
    return 0;
}
