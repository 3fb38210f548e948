// "Ring" GEMM for mid-size plain GEMMs (round 5): 64 x 160 tile, 512 threads = 8 waves as 4 (M) x 2 (N), 16 x 80 per wave, ONE workgroup per CU,
// operands global -> LDS by LDS-DMA into a ring of five 28-KiB K-tile stages — three to four K-tiles (84-112 KiB per CU) in flight at all times.
//
// Why: the SD1.5 32^2 level runs 25 N = K = 1280 projections per step on M = 2048 rows (transformer.py:186-245, 342-377: proj_in, attn1.to_out, attn2.to_q,
// attn2.to_out, proj_out).  The register-staged kernel of gemm.hip keeps ONE K-tile of loads in flight per workgroup: with 64 x 64 tiles (2.5 workgroups per
// CU) the launch is bound by the 205 MB it moves through the L2 -> LDS path (22.8 us, 294 TFLOP/s), with 64 x 160 tiles (45.7 instead of 32 flop per
// operand byte, one round of 256 workgroups) by the latency of each K-tile's loads (25.6 us: one 4-wave workgroup per CU hides nothing;
// profiles/r05/ab_caches_tile64x160.txt).  Both walls fall together only with the wide tile AND several K-tiles in flight, which needs the LDS-DMA
// ring of gemm_pp.inc (no staging registers, counted vmcnt) at a tile small enough to give every CU work at M = 2048.
//
// Structure per K-tile t (one s_barrier per K-tile, every wave runs the same sequence):
//   issue the DMA pieces of K-tile t + 4 into the slot K-tile t - 1 occupied (freed by the barrier that ended iteration t - 1);
//   read the fragments of K-tile t (landed: waited for in iteration t - 1) and run its 10 MFMAs (16 x 80 per wave x 64 k);
//   wait with a COUNTED vmcnt until this wave's pieces of K-tile t + 1 have landed; barrier (everybody's pieces landed, everybody done with t).
// A K-tile is 28 pieces of 1 KiB (8 rows x 128 B): the 8 A pieces go one to each wave, the 20 W pieces 3 to waves 0-3 and 2 to waves 4-7, so a wave
// issues 4 or 3 instructions per K-tile and its vmcnt immediates are compile-time constants per wave group.
// Same 128-B LDS rows and 16-B-chunk XOR swizzle as the other two main loops (applied on the SOURCE address, as in gemm_pp.inc); the output stage is
// the shared gemm_epilogue (bias, residual, per-batch row vector, gate, GroupNorm statistics for the consumer, fp32 output ...).
// Plain mode only (no conv taps, no split-K, no GEGLU pairing, no MX operands, no folded LayerNorm); K % 64 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"

namespace ldx {

typedef __attribute__((ext_vector_type(4))) int gr_i32x4;
static __device__ __forceinline__ gr_i32x4 gr_srd(const void* base, long bytes) {
    const unsigned long long q = (unsigned long long)base;
    const int n = (int)(bytes > 0x7fffffffL ? 0x7fffffffL : (bytes > 0 ? bytes : 0));
    return (gr_i32x4){(int)(unsigned)q, (int)((unsigned)(q >> 32) & 0xffffu), n, 0x00020000};
}
// M0 is written without being declared (gemm_pp.inc explains why that is safe in these kernels)
static __device__ __forceinline__ void gr_dma16(const gr_i32x4 rsrc, int voff, int soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

constexpr int GR_BM = 64, GR_BN = 160, GR_NS = 5;
constexpr int GR_STAGE = (GR_BM + GR_BN) * 128;          // 28 672 B
constexpr int GR_LDS = GR_NS * GR_STAGE;                 // 143 360 B

template <int N> static __device__ __forceinline__ void gr_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// wait until at most `tiles` later K-tiles of this wave (P pieces each) are still in flight
template <int P> static __device__ __forceinline__ void gr_wait_tiles(int tiles) {
    if (tiles >= 3) gr_wait<3 * P>();
    else if (tiles == 2) gr_wait<2 * P>();
    else if (tiles == 1) gr_wait<P>();
    else gr_wait<0>();
}

template <typename T>
__global__ __launch_bounds__(512, 1) void gemm_ring_kernel(const GemmArgs p) {
    constexpr int OOB = (int)0x80000000;
    constexpr int MI = 1, NJ = 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int tiles_n = (p.N + GR_BN - 1) / GR_BN, tiles_m = (p.M + GR_BM - 1) / GR_BM;
    const int lin = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lin / tiles_n, tn = lin - tm * tiles_n;        // an XCD's contiguous range of tiles walks the columns of a few tile rows: shared A panels
    const int m0 = tm * GR_BM, n0 = tn * GR_BN;
    const int nk = p.K / BK;

    const gr_i32x4 rA = gr_srd((const char*)p.A + (long)m0 * p.lda * 2, ((long)(p.M - m0 - 1) * p.lda + p.K) * 2);
    const gr_i32x4 rW = gr_srd((const char*)p.W + (long)n0 * p.K * 2, ((long)(p.N - n0) * p.K) * 2);
    // this wave's DMA pieces: piece = 8 tile rows, lane l -> row 8 piece + (l >> 3), LDS position l & 7 <- source chunk (l & 7) ^ (row & 7)
    const int prow = lane >> 3, gchunk = (lane & 7) ^ prow;       // (8 piece + prow) & 7 == prow
    const int arow = 8 * wave + prow;
    const int a_voff = (m0 + arow < p.M) ? (arow * p.lda + gchunk * 8) * 2 : OOB;
    int w_voff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave + 8 * i;                               // W piece index 0 .. 19 (i = 2 exists for waves 0-3 only)
        const int r = 8 * q + prow;
        w_voff[i] = (q < 20 && n0 + r < p.N) ? (r * p.K + gchunk * 8) * 2 : OOB;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int kt, int slot) __attribute__((always_inline)) {
        const unsigned dst = lds_base + slot * GR_STAGE;
        const int soff = kt * (BK * 2);
        gr_dma16(rA, a_voff, soff, dst + wave * 1024);
        gr_dma16(rW, w_voff[0], soff, dst + (8 + wave) * 1024);
        gr_dma16(rW, w_voff[1], soff, dst + (16 + wave) * 1024);
        if (wave < 4) gr_dma16(rW, w_voff[2], soff, dst + (24 + wave) * 1024);
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Residual operand requested NOW (round 6; the register-staged kernel has done so since round 2): these launches are chains of a few memory round trips
    // (operands, residual, stores), and this one hides under the K loop.  Same predicate as the output stage's vector path.
    uint2 rpre[MI * NJ];
    const bool use_rpre = p.R != nullptr && p.splitk <= 1 && !p.geglu && !p.C8;
    if (use_rpre) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = m0 + wm * 16 + l15, n = n0 + wn * (GR_BN / 2) + j * 16 + 4 * g4;
            rpre[j] = (m < p.M && n + 3 < p.N) ? *(const uint2*)((const T*)p.R + (long)m * p.ldr + n) : make_uint2(0u, 0u);
        }
    }

    // prologue: K-tiles 0 .. NS - 2 in flight, wait for K-tile 0
    const int npre = nk < GR_NS - 1 ? nk : GR_NS - 1;
    for (int t = 0; t < npre; ++t) issue(t, t);
    if (wave < 4) gr_wait_tiles<4>(npre - 1); else gr_wait_tiles<3>(npre - 1);
    asm volatile("s_barrier" ::: "memory");

    // fragment offsets inside a stage: row r, logical chunk q = 4 ks + g4 lives at r * 128 + ((q ^ (r & 7)) << 4)
    const int fx = (g4 ^ (l15 & 7)) << 4;
    const int a_off = (wm * 16 + l15) * 128 + fx;                              // ks = 1: ^ 64
    const int b_off = (GR_BM + wn * (GR_BN / 2) + l15) * 128 + fx;             // + j * 2048

    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        if (t + GR_NS - 1 < nk) issue(t + GR_NS - 1, slot == 0 ? GR_NS - 1 : slot - 1);
        const char* st = smem + slot * GR_STAGE;
        V8 af[2], bf[NJ][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[ks] = as_v8<T>(*(const uint4*)(st + (a_off ^ (ks * 64))));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf[j][ks] = as_v8<T>(*(const uint4*)(st + ((b_off + j * 2048) ^ (ks * 64))));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[0][j] = mfma16(bf[j][ks], af[ks], acc[0][j]);
        // K-tile t + 1 must have landed before the next iteration reads it: the K-tiles issued after it may stay in flight
        const int issued = (t + GR_NS < nk ? t + GR_NS : nk);                 // K-tiles issued so far
        const int later = issued - (t + 2);                                    // issued after K-tile t + 1
        if (t + 1 < nk) { if (wave < 4) gr_wait_tiles<4>(later); else gr_wait_tiles<3>(later); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        slot = slot + 1 == GR_NS ? 0 : slot + 1;
    }
    // (the loop ended on a barrier behind every wave's last fragment reads and with no DMA in flight: the ring is free for the output stage's scratch)
    gemm_epilogue<T, GR_BM, GR_BN, 4, MI, NJ, false, MI * NJ, true>(p, acc, m0, n0, wm, wn, l15, g4, 0, 1, nullptr, nullptr, rpre, use_rpre);
}

// Planner rule (gemm_tile asks it): the shapes the register-staged 64 x 64 tiles serve today — fewer than 400 tiles of 128 x 128 and a short K — whose
// 64 x 160 tiling still gives (nearly) every CU a workgroup.  LDX_RING=0 switches the kernel off, 2 takes every plain GEMM with N % 160 == 0 that fills the chip.
bool gemm_ring_ok(int M, int N, int K, bool plain, int splitk) {
    static const int mode = getenv("LDX_RING") ? atoi(getenv("LDX_RING")) : 1;
    if (!mode || !plain || splitk > 1 || N % GR_BN || K % BK || K < 2 * BK || M <= 0) return false;
    const long t = (long)((M + GR_BM - 1) / GR_BM) * (N / GR_BN);
    if (mode >= 2) return t >= 192;
    return t >= 192 && t <= 512;
}

void launch_gemm_ring(const GemmArgs& a, DType dt, hipStream_t s) {
    gemm_gn_tile_check(a, GR_BM, GR_BN, 1);
    const unsigned tiles = (unsigned)(((a.M + GR_BM - 1) / GR_BM) * ((a.N + GR_BN - 1) / GR_BN));
    if (dt == DT_BF16) {
        static DevOnce once;
        set_dyn_lds(once, (const void*)gemm_ring_kernel<__bf16>, GR_LDS);
        hipLaunchKernelGGL((gemm_ring_kernel<__bf16>), dim3(tiles), dim3(512), GR_LDS, s, a);
    } else {
        static DevOnce once;
        set_dyn_lds(once, (const void*)gemm_ring_kernel<_Float16>, GR_LDS);
        hipLaunchKernelGGL((gemm_ring_kernel<_Float16>), dim3(tiles), dim3(512), GR_LDS, s, a);
    }
}

}  // namespace ldx
