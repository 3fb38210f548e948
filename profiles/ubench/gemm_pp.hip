// Prototype of the "ping-pong" GEMM main loop for gfx950 (round 2): C[M][N] = A[M][K] * W[N][K]^T, bf16 in, fp32 accumulate.
//
// 256 x 256 tile, 512 threads = 8 waves (4 along M x 2 along N, 64 x 128 per wave as 4 x 8 MFMA 16x16x32 tiles), BK = 64.
// * Operands go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction); the LDS image keeps the
//   production kernel's 128-B rows with the 16-B-chunk XOR swizzle, applied on the SOURCE address (lane l of an instruction
//   lands at l * 16, so it fetches chunk (l & 7) ^ (row & 7) of row l >> 3).
// * LDS = a ring of five 32-KiB half-tiles in the order A(0) B(0) A(1) B(1) ...; phase p (two per K-tile) issues half-tile
//   p + 4, reads fragments from half-tiles p (and p + 1), and waits with a COUNTED vmcnt until half-tile p + 2 has landed, so
//   two to three half-tiles (64 - 96 KiB per CU) stay in flight across the barriers.
// * Waves 0-3 and 4-7 (one of each per SIMD) run the same phase sequence one barrier apart: while one group issues its 32
//   MFMAs of a phase from registers, the other does its ds_reads / DMA issue / waits for the next phase.
//
// Build: hipcc --offload-arch=gfx950 -O3 gemm_pp.hip -o gemm_pp      Run: ./gemm_pp [M N K] [check]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const i32x4 rsrc, int voff, int soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int nx = 8;
    if (nblk < 2 * nx) return bid;
    const int q = nblk / nx, r = nblk % nx;
    const int xcd = bid % nx, idx = bid / nx;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

constexpr int HALF = 256 * 128;       // bytes of one half-tile (256 rows x 128 B)
constexpr int NSLOT = 5;

// ABL: 0 full, 1 no MFMA (loads + fragment reads only), 2 no global loads (MFMA + fragment reads on whatever LDS holds)
template <int ABL, int GM>
__global__ __launch_bounds__(512, 1) void gemm_pp(const __bf16* __restrict__ A, const __bf16* __restrict__ W, __bf16* __restrict__ C,
                                                  const int M, const int N, const int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g4 = lane >> 4;

    const int tiles_m = M / 256, tiles_n = N / 256, ntiles = tiles_m * tiles_n;
    const int lin = xcd_remap(blockIdx.x, ntiles);
    // grouped order: GM tile rows at a time, column-major inside the group, so the ~32 tiles an XCD runs concurrently form a
    // compact 2-D block (fewer distinct A / W panels per L2)
    const int gsz = GM * tiles_n, grpi = lin / gsz, within = lin - grpi * gsz;
    const int gm_eff = min(GM, tiles_m - grpi * GM);
    const int tm = grpi * GM + within % gm_eff, tn = within / gm_eff;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = K / 64, P = 2 * nk;

    const unsigned long long pa = (unsigned long long)(A + (size_t)m0 * K), pw = (unsigned long long)(W + (size_t)n0 * K);
    const i32x4 rA = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)min((long long)(M - m0) * K * 2, 0x7fffffffLL), 0x00020000};
    const i32x4 rW = {(int)(unsigned)pw, (int)(unsigned)(pw >> 32), (int)min((long long)(N - n0) * K * 2, 0x7fffffffLL), 0x00020000};

    // staging coordinates: instruction j (0..3) of this wave covers rows j*64 + wave*8 + (lane >> 3), LDS position chunk lane & 7
    const int srow = tid >> 3;
    const int gchunk = (lane & 7) ^ (srow & 7);
    int voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) voff[j] = ((srow + 64 * j) * K + gchunk * 8) * 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wofs = wave * 1024;        // + j * 8192 + slot * HALF

    auto issue = [&](int s, int slot_of) __attribute__((always_inline)) {       // half-tile s (in ring slot slot_of): even = A of K-tile s/2, odd = W
        if (ABL == 2) return;
        const int soff = (s >> 1) * 128;
        const unsigned dst = lds0 + slot_of * HALF + wofs;
        if (s & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(rW, voff[j], soff, dst + j * 8192);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(rA, voff[j], soff, dst + j * 8192);
        }
    };

    // fragment read offsets: row r, logical chunk q lives at r * 128 + ((q ^ (r & 7)) << 4); q = ks * 4 + g4
    const int fx = (g4 ^ (l15 & 7)) << 4;
    const int a_off = (wm * 64 + l15) * 128 + fx;           // + i * 2048, ks: ^ 64
    const int b_off = (wn * 128 + l15) * 128 + fx;          // + j * 2048

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");

    bf16x8 af[4][2], bf[4][2];
    int slot = 0;                 // ring slot of half-tile 2t
    for (int t = 0; t < nk; ++t, slot = slot + 2 >= NSLOT ? slot + 2 - NSLOT : slot + 2) {
        const int slot1 = slot + 1 >= NSLOT ? slot + 1 - NSLOT : slot + 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = 2 * t + h;
            // ---------------- memory part ----------------
            {   // half-tile p + 4 goes where half-tile p - 1 was
                const int sl = slot + h - 1;
                if (p + 4 < P) issue(p + 4, sl < 0 ? sl + NSLOT : sl);
            }
            const char* sa = smem + slot * HALF;
            const char* sb = smem + slot1 * HALF;
            if (h == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(sa + ((a_off + i * 2048) ^ (ks * 64)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bf[j][ks] = *(const bf16x8*)(sb + ((b_off + (h * 4 + j) * 2048) ^ (ks * 64)));
            if (p + 4 < P) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- compute part ----------------
            if (ABL != 1) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][h * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][ks], af[i][ks], acc[i][h * 4 + j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(af[i][0]), "v"(af[i][1]), "v"(bf[j][0]), "v"(bf[j][1]));
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
        }
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");

    // epilogue: lane holds row m = .. + l15, columns n = .. + 4 * g4 + r
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + 4 * g4;
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (__bf16)acc[i][j][r];
            *(bf16x4*)(C + (size_t)m * N + n) = o;
        }
    }
}

// plain reference: one thread per output, 16-byte loads
__global__ void gemm_ref(const __bf16* __restrict__ A, const __bf16* __restrict__ W, float* __restrict__ C, int M, int N, int K) {
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
    float s = 0.f;
    for (int k = 0; k < K; k += 8) {
        const bf16x8 a = *(const bf16x8*)(A + (size_t)m * K + k), w = *(const bf16x8*)(W + (size_t)n * K + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)w[e];
    }
    C[(size_t)m * N + n] = s;
}
__global__ void cmp_kernel(const __bf16* C, const float* R, size_t n, float* maxerr, float* maxref) {
    float e = 0.f, r = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        e = fmaxf(e, fabsf((float)C[i] - R[i])); r = fmaxf(r, fabsf(R[i]));
    }
    atomicMax((int*)maxerr, __float_as_int(e)); atomicMax((int*)maxref, __float_as_int(r));
}
__global__ void fill_kernel(__bf16* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (__bf16)((float)(x & 0xffff) / 32768.0f - 1.0f);
    }
}

template <int ABL, int GM>
static float run(const __bf16* A, const __bf16* W, __bf16* C, int M, int N, int K, int reps) {
    const int lds = NSLOT * HALF;
    CK(hipFuncSetAttribute((const void*)gemm_pp<ABL, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int tiles = (M / 256) * (N / 256);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_pp<ABL, GM>), dim3(tiles), dim3(512), lds, 0, A, W, C, M, N, K);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_pp<ABL, GM>), dim3(tiles), dim3(512), lds, 0, A, W, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    int M = 8192, N = 8192, K = 8192;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    const bool check = argc >= 5;
    __bf16 *A, *W, *C; float* R;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    fill_kernel<<<2048, 256>>>(A, (size_t)M * K, 0x1234u);
    fill_kernel<<<2048, 256>>>(W, (size_t)N * K, 0x9876u);
    CK(hipDeviceSynchronize());
    const double fl = 2.0 * M * N * K;
    const double tile_bytes = (double)(M / 256) * (N / 256) * (K / 64) * 65536.0;
    if (check) {
        CK(hipMalloc(&R, (size_t)M * N * 4));
        hipLaunchKernelGGL(gemm_ref, dim3(N / 16, M / 16), dim3(256), 0, 0, A, W, R, M, N, K);
        float* d; CK(hipMalloc(&d, 8));
        std::vector<__bf16> first((size_t)M * N), again((size_t)M * N);
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemset(C, 0xff, (size_t)M * N * 2)); CK(hipMemset(d, 0, 8));
            run<0, 8>(A, W, C, M, N, K, 1);
            cmp_kernel<<<1024, 256>>>(C, R, (size_t)M * N, d, d + 1);
            float h[2]; CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(rep ? again.data() : first.data(), C, (size_t)M * N * 2, hipMemcpyDeviceToHost));
            const bool same = rep == 0 || memcmp(first.data(), again.data(), (size_t)M * N * 2) == 0;
            printf("check %d: max |err| %.4g  (max |ref| %.4g)  %s  %s\n", rep, h[0], h[1], h[0] <= 0.01f * h[1] + 0.02f ? "OK" : "FAIL", same ? "bit-identical" : "DIFFERS FROM RUN 0");
        }
    }
    for (int round = 0; round < 3; ++round) {
        const float t0 = run<0, 8>(A, W, C, M, N, K, 10);
        const float t4 = run<0, 4>(A, W, C, M, N, K, 10);
        const float t1 = run<0, 1>(A, W, C, M, N, K, 10);
        const float tl = run<1, 8>(A, W, C, M, N, K, 10);
        const float tc = run<2, 8>(A, W, C, M, N, K, 10);
        printf("%dx%dx%d  full GM8 %.3f ms %.0f TF | GM4 %.0f TF | GM1 %.0f TF | loads+reads only %.3f ms (%.1f TB/s into LDS) | mfma+reads only %.3f ms (%.0f TF)\n",
               M, N, K, t0, fl / t0 / 1e9, fl / t4 / 1e9, fl / t1 / 1e9, tl, tile_bytes / tl / 1e9, tc, fl / tc / 1e9);
    }
    return 0;
}
