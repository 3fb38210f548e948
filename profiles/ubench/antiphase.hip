// Round 3: does a barrier-enforced ANTI-PHASE of the two waves of a SIMD overlap the matrix pipe and the VALU?  512-thread workgroups
// (two waves per SIMD from the same workgroup): waves 0-3 run the D = 40 attention's MFMA block (28 MFMA 32x32x16) while waves 4-7 run its
// softmax block (64 exp + 64 fma + 32 max + 32 pack), s_barrier, roles swapped, s_barrier.  MOVE = number of the 32 VALU slices that
// ride inside the MFMA phase (1 slice per MFMA) to balance the two phases.  Reference rows: the free-running blocked / interleaved
// loops of mix32.hip at the same occupancy (256-thread workgroups, two per CU).
// Build: hipcc --offload-arch=gfx950 -O3 antiphase.hip -o antiphase
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct St {
    f32x16 acc[4]; bf16x8 a, b; float e[32]; float mx; unsigned pk; float c;
    __device__ __forceinline__ void init(float cc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(0.5f - i); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) e[i] = threadIdx.x * 0.001f + i * 0.01f;
        mx = 0.f; pk = 0; c = cc;
    }
    __device__ __forceinline__ void slice(int v) {        // 2 fma, 2 exp, 1 max3, 1 pack
        const int i0 = (2 * v) & 31, i1 = (2 * v + 1) & 31;
        e[i0] = __builtin_amdgcn_exp2f(fmaf(e[i0], c, -0.25f));
        e[i1] = __builtin_amdgcn_exp2f(fmaf(e[i1], c, -0.25f));
        mx = fmaxf(fmaxf(mx, e[i0]), e[i1]);
        unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(e[i0]), "v"(e[i1]));
        pk ^= r;
    }
    __device__ __forceinline__ void mfma(int m) { acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0); }
    __device__ __forceinline__ float fold() {
        float s = mx + (float)pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
        for (int i = 0; i < 32; ++i) s += e[i];
        return s;
    }
};

template <int MOVE, int PRIO>
__global__ __launch_bounds__(512, 1) void anti(float* out, int iters, float c) {
    St s; s.init(c);
    const int g = threadIdx.x >> 8;          // waves 0-3 / 4-7: one of each per SIMD
    auto mphase = [&]() __attribute__((always_inline)) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 28; ++m) { s.mfma(m); if (MOVE != 99 && m < MOVE) s.slice(m); __builtin_amdgcn_sched_barrier(0); }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto vphase = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int v = MOVE; v < 32; ++v) s.slice(v);     // MOVE = 99: no VALU work at all
        __builtin_amdgcn_sched_barrier(0);
    };
    if (g == 0) {
        for (int it = 0; it < iters; ++it) { mphase(); __builtin_amdgcn_s_barrier(); vphase(); __builtin_amdgcn_s_barrier(); }
    } else {
        for (int it = 0; it < iters; ++it) { vphase(); __builtin_amdgcn_s_barrier(); mphase(); __builtin_amdgcn_s_barrier(); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.fold();
}

template <int MODE>     // 2 blocked, 3 interleaved: mix32.hip's free-running loops (256 threads, two workgroups per CU)
__global__ __launch_bounds__(256, 2) void freerun(float* out, int iters, float c) {
    St s; s.init(c);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int m = 0; m < 28; ++m) s.mfma(m);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < 32; ++v) s.slice(v);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int m = 0; m < 28; ++m) { s.mfma(m); s.slice(m); if (m < 4) s.slice(28 + m); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.fold();
}

template <typename K> void run(const char* name, K kern, int threads, int blocks) {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.0f nominal cycles (2.4 GHz) of SIMD time per wave-iteration\n", name, ms * 1e-3 * 2.4e9 / iters / 2);
    (void)hipFree(d);
}
int main() {
    run("free-running, blocked (256 thr x 2 WG/CU)", freerun<2>, 256, 512);
    run("free-running, interleaved 1 MFMA : ~7 VALU", freerun<3>, 256, 512);
    run("anti-phase by s_barrier, MOVE 0", anti<0, 0>, 512, 256);
    run("anti-phase by s_barrier, MOVE 0, setprio 1 in the MFMA phase", anti<0, 1>, 512, 256);
    run("anti-phase, MFMA phase only (VALU phase empty: MOVE 32 -> 28 slices... see below)", anti<99, 0>, 512, 256);
    run("anti-phase, 4 VALU slices inside the MFMA phase", anti<4, 0>, 512, 256);
    run("anti-phase, 8 VALU slices inside the MFMA phase", anti<8, 0>, 512, 256);
    run("anti-phase, 12 VALU slices inside the MFMA phase", anti<12, 0>, 512, 256);
    run("anti-phase, 16 VALU slices inside the MFMA phase (= both waves interleave)", anti<16, 0>, 512, 256);
    return 0;
}
