// Weight-streaming access-pattern probe (round 2): how fast can 512-thread workgroups pull a cold [N][K] bf16 weight matrix into LDS by LDS-DMA
// when a K-tile of a 160-row tile is (a) 160 rows x 128 B at a row stride of 2 K bytes (row-major weights, what the GEMM does today) or
// (b) five contiguous 4-KiB blocks (weights pre-tiled as [N/32][K/64][32][64])?  Same bytes, same instruction count, same ring depth.
// Build: hipcc --offload-arch=gfx950 -O3 wstream.hip -o wstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ void dma16(const i32x4 rsrc, int voff, int soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// grid = (N / 160) * S workgroups; workgroup (tn, split) streams rows [160 tn, +160) x k-tiles [split * nk / S, ...)
template <int TILED>
__global__ __launch_bounds__(512, 2) void wstream(const char* W, int N, int K, int S, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = N / 160, tn = blockIdx.x % tiles_n, split = blockIdx.x / tiles_n;
    const int nk_all = K / 64, kb = (int)((long)split * nk_all / S), ke = (int)((long)(split + 1) * nk_all / S);
    const unsigned long long pw = (unsigned long long)W;
    const i32x4 r = {(int)(unsigned)pw, (int)((unsigned)(pw >> 32) & 0xffff), (int)min((long long)N * K * 2, 0x7fffffffLL), 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    int voff[3];
    for (int j = 0; j < 3; ++j) {
        const int row = (tid >> 3) + 64 * j, n = tn * 160 + row, ch = lane & 7;
        if (row >= 160) voff[j] = (int)0x80000000;
        else if (TILED) voff[j] = (n >> 5) * (K / 64) * 4096 + (n & 31) * 128 + ch * 16;
        else voff[j] = n * K * 2 + ch * 16;
    }
    int slot = 0;
    for (int kt = kb; kt < ke; ++kt) {
        const int soff = TILED ? kt * 4096 : kt * 128;
        for (int j = 0; j < 3; ++j) dma16(r, voff[j], soff, lds0 + slot * 32768 + j * 8192);
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");        // three K-tiles in flight
        slot = slot == 3 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = *(float*)smem;
}
int main() {
    const int N = 1280, K = 11520, NB = 24;       // 24 distinct matrices = 708 MB > Infinity Cache
    char* W; float* out;
    CK(hipMalloc(&W, (size_t)NB * N * K * 2)); CK(hipMemset(W, 1, (size_t)NB * N * K * 2)); CK(hipMalloc(&out, 1 << 20));
    CK(hipFuncSetAttribute((const void*)wstream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)wstream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int S : {4, 12, 30, 60}) for (int tiled = 0; tiled < 2; ++tiled) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int grid = (N / 160) * S;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int b = 0; b < NB; ++b) {
                if (tiled) hipLaunchKernelGGL(wstream<1>, dim3(grid), dim3(512), 131072, 0, W + (size_t)b * N * K * 2, N, K, S, out);
                else hipLaunchKernelGGL(wstream<0>, dim3(grid), dim3(512), 131072, 0, W + (size_t)b * N * K * 2, N, K, S, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("S=%2d (%3d workgroups) %s: %.1f us per 29.5 MB matrix = %.0f GB/s\n", S, grid, tiled ? "tiled [N/32][K/64][32][64]" : "row-major [N][K]          ", ms * 1e3 / NB, (double)N * K * 2 / (ms / NB) / 1e6);
    }
    return 0;
}
