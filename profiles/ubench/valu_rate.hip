// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for the op mix of the softmax
// (fma, exp, max3, cvt_pk, pk_mul) at 1..4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, int iters, float c) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    float m = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = fmaf(a[i], c, 0.5f);                                   // 16 fma
            if (MODE == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);                          // 16 exp
            if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(fmaf(a[i], c, -m));             // 16 fma + 16 exp
            if (MODE == 3) { m = fmaxf(fmaxf(m, a[i]), a[(i + 1) & 15]); a[i] += 1.0f; } // max3 + add
        }
        if (MODE == 4) {                                                                 // packed mul x8 (16 values)
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 8; ++i) { f2 v = {a[2 * i], a[2 * i + 1]}; v = v * (f2){c, c}; a[2 * i] = v[0]; a[2 * i + 1] = v[1]; }
        }
    }
    float s = m;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int instr_per_iter) {
    float* d; (void)hipMalloc(&d, 256 * 8 * 1024 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {                     // waves per SIMD: blocks of 256 threads (4 waves), wps blocks per CU
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0001f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves each issuing iters*instr_per_iter instructions
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-14s waves/SIMD %d: %.2f cycles per wave-instruction (per SIMD, assuming 2.4 GHz)\n", name, wps,
               cyc / ((double)iters * instr_per_iter * wps));
    }
    (void)hipFree(d);
}

int main() {
    run<0>("fma", 16); run<1>("exp", 16); run<2>("fma+exp", 32); run<3>("max3+add", 32); run<4>("pk_mul", 8);
    return 0;
}
