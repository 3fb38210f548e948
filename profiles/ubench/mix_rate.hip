// MFMA x VALU co-issue microbenchmark for gfx950: does one SIMD overlap the matrix pipe with the VALU /
// transcendental stream (a) across two waves, (b) inside one wave?  Each loop iteration issues NM
// mfma_f32_16x16x32_bf16 (independent accumulators) and NE v_exp_f32 + NF v_fma_f32, either as two blocks
// (MFMA block, then VALU block) or finely interleaved (pinned with sched_barrier).
// Build: hipcc --offload-arch=gfx950 -O3 mix_rate.hip -o mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: MFMA only, 1: VALU only, 2: both, blocked, 3: both, interleaved 1 MFMA : (NE+NF)/NM VALU,
// 4: blocked, but every second workgroup starts half an iteration late (one extra VALU phase up front), so the two waves
//    that share a SIMD run in anti-phase
template <int MODE, int NM, int NE, int NF>
__global__ __launch_bounds__(512) void k(float* out, int iters, float c) {
    f32x4 acc[8];
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(0.5f - i); }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = threadIdx.x * 0.001f + i * 0.01f;
    // anti-phase: with 512-thread workgroups waves w and w+4 share a SIMD; otherwise alternate workgroups
    if (MODE == 4 && (blockDim.x == 512 ? (threadIdx.x >= 256) : (blockIdx.x & 1))) {
#pragma unroll
        for (int v = 0; v < NE; ++v) e[v & 15] = __builtin_amdgcn_exp2f(e[v & 15]);
#pragma unroll
        for (int v = 0; v < NF; ++v) e[v & 15] = fmaf(e[v & 15], c, 0.25f);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int v = 0; v < NE; ++v) e[v & 15] = __builtin_amdgcn_exp2f(e[v & 15]);
#pragma unroll
            for (int v = 0; v < NF; ++v) e[v & 15] = fmaf(e[v & 15], c, 0.25f);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) {
            constexpr int NV = NE + NF;
            int ve = 0, vf = 0;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
                for (int v = (m * NV) / NM; v < ((m + 1) * NV) / NM; ++v) {
                    // spread the exps evenly through the VALU stream
                    if ((v * NE) / NV != ((v + 1) * NE) / NV) { e[ve & 15] = __builtin_amdgcn_exp2f(e[ve & 15]); ++ve; }
                    else { e[(vf + 5) & 15] = fmaf(e[(vf + 5) & 15], c, 0.25f); ++vf; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NM, int NE, int NF>
void run(const char* name) {
    float* d; (void)hipMalloc(&d, 256 * 8 * 1024 * 4);
    const int iters = 4000;
    for (int wps = 1; wps <= 3; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<MODE, NM, NE, NF>), dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0001f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NM, NE, NF>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wps);      // SIMD cycles per wave-iteration
        printf("%-34s waves/SIMD %d: %8.1f cycles per wave-iteration (per SIMD @2.4 GHz)\n", name, wps, cyc);
    }
    (void)hipFree(d);
}

template <int MODE, int NM, int NE, int NF>
void run512(const char* name) {
    float* d; (void)hipMalloc(&d, 256 * 8 * 1024 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NM, NE, NF>), dim3(256), dim3(512), 0, 0, d, 10, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NM, NE, NF>), dim3(256), dim3(512), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s 512-thread WG, 2 waves/SIMD: %8.1f cycles per wave-iteration (per SIMD @2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / ((double)iters * 2));
    (void)hipFree(d);
}

int main() {
    run512<2, 56, 68, 210>("blocked 512");
    run512<4, 56, 68, 210>("blocked 512 anti-phase waves 4-7");
    run512<2, 56, 68, 100>("blocked 512 56/68/100");
    run512<4, 56, 68, 100>("blocked 512 anti-phase 56/68/100");
    // the D=40 attention mix per wave and key block (QT=4): 56 MFMA, 68 exp, 210 plain VALU
    run<0, 56, 68, 210>("mfma only        56/0/0");
    run<1, 56, 68, 210>("valu only        0/68/210");
    run<2, 56, 68, 210>("blocked          56/68/210");
    run<3, 56, 68, 210>("interleaved      56/68/210");
    run<4, 56, 68, 210>("blocked, anti-phase blocks");
    // the pipelined kernel's mix: 56 MFMA, 68 exp, 100 plain VALU
    run<1, 56, 68, 100>("valu only        0/68/100");
    run<2, 56, 68, 100>("blocked          56/68/100");
    run<3, 56, 68, 100>("interleaved      56/68/100");
    run<4, 56, 68, 100>("blocked anti-phase 56/68/100");
    run<1, 56, 64, 0>("exp only         0/64/0");
    run<3, 56, 64, 0>("interleaved      56/64/0");
    run<3, 56, 0, 112>("interleaved      56/0/112");
    return 0;
}
