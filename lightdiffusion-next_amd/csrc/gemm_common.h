// Device code shared by the two GEMM main loops (gemm.hip: register-staged 128-row tiles; gemm_pp.hip: 256-row ping-pong tiles):
// K-tile constants, the folded-LayerNorm helpers and the output stage with every fused epilogue.
#pragma once
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"

namespace ldx {

template <typename T> struct DTypeOf;
template <> struct DTypeOf<__bf16> { static constexpr DType v = DT_BF16; };
template <> struct DTypeOf<_Float16> { static constexpr DType v = DT_F16; };

constexpr int BK = 64;
#ifndef LDX_PP_STAMP
#define LDX_PP_STAMP(k)        // profiles/ubench/pp_stamp.hip defines it: per-workgroup wall-clock stamps of a tile's phases
#endif
// gemm_ring.hip: 64 x 160 LDS-DMA ring tiles for mid-size plain GEMMs; gemm_tile (gemm.hip) asks gemm_ring_ok
bool gemm_ring_ok(int M, int N, int K, bool plain, int splitk);
void launch_gemm_ring(const GemmArgs& a, DType dt, hipStream_t s);
// conv_patch.hip: patch-resident 3x3 conv for N = 32 / 64 / 128 (ESRGAN, VAE last level); launch_gemm asks conv_patch_ok
bool conv_patch_ok(const GemmArgs& a);
int conv_patch_gn_chunks(const GemmArgs& a, int HW, int G);
void launch_conv_patch(const GemmArgs& a, DType dt, hipStream_t s);
// BN is a template parameter: 128 (generic) or 160 — every SD1.5 channel count is a multiple of 320, and
// 160-wide tiles remove the half-empty last column tile (and the half-empty last round of workgroups)
// that N = 320 / 960 / 1920 get with 128.
template <int BM, int BN> constexpr int stage_bytes() { return (BM + BN) * BK * 2; }

// Folded LayerNorm, statistics side.  Every wave streams all K columns of its rows through its A fragments, so the row sums come out of
// one extra MFMA per fragment, next to the real ones and with the same exact fp32 accumulation: the wave with wn = 0 issues ones x A^T
// (every accumulator row = sum_k a[m][k]), its neighbour with wn = 1 issues A x A^T (diagonal = sum_k a[m][k]^2); after the K loop the two
// exchange through LDS.  No pass over the activation, nothing crosses workgroups, fixed order.  (First version: v_dot2c_f32 on every
// pair, both waves: 64 VALU dot products per K-tile next to 40 MFMAs made the q|k|v projection VALU-bound, 50 -> 74 us at level 0.)
template <typename T> __device__ __forceinline__ typename Vec<T>::v8 ones_v8() {
    typename Vec<T>::v8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)1.0f;
    return o;
}
// st: LDS scratch of 2 * ROWS floats (free after the K loop); on return s1 / s2 hold mean and 1 / sqrt(var + eps) of the lane's MI rows.
// lnacc[i] of a wn = 0 wave: any element = the row sum; of a wn = 1 wave: element l15 & 3 on the lanes with g4 == l15 >> 2 = the sum of squares.
template <int MI, int ROWS>
__device__ __forceinline__ void ln_exchange(const f32x4 (&lnacc)[MI], float* st, const int wrow0, const int wn, const int l15, const int g4,
                                            const int K, const float eps, float (&mean)[MI], float (&rstd)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wrow0 + i * 16 + l15;
        if (wn == 0) { if (g4 == 0) st[row] = lnacc[i][0]; }
        else if (g4 == (l15 >> 2)) {
            const int r = l15 & 3;
            st[ROWS + row] = r == 0 ? lnacc[i][0] : r == 1 ? lnacc[i][1] : r == 2 ? lnacc[i][2] : lnacc[i][3];
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)K;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wrow0 + i * 16 + l15;
        const float m = st[row] * inv;
        mean[i] = m;
        rstd[i] = rsqrtf(fmaxf(st[ROWS + row] * inv - m * m, 0.f) + eps);
    }
}

// Lean output path (round 6) for the epilogues that make up almost every launch: bias [+ per-image vector] [* gate] [+ residual] -> 16-bit store
// [+ second store of a shared CFG prefix] [+ GroupNorm statistics], on tiles made of whole 16 x 16 blocks (M % 16 == 0, N % 16 == 0: every predicate is
// wave-uniform).  The general stage below handles every operand combination in one body — about a hundred executed instructions per 16 x 16 block, and
// the stage runs at the speed of instruction issue (stamps of profiles/ubench/pp_stamp.hip: a quarter of a K = 3072 tile's life, and neither
// compiling the unused operands out nor coalescing the stores through an LDS transposition — profiles/experiments/gemm_epilogue_rowwise.h.txt —
// changed that by more than 2x).  Here the operand kinds that cost registers (residual rows HR, parked statistics HG) are template parameters and the
// cheap ones (per-image vector, gate, second store) sit behind wave-uniform branches compiled in only with HX: ten to twenty instructions per block.
// Per element the same float operations in the same order as the general stage.  ev_*: the per-column operands in LDS (tile-relative columns);
// rall: the tile's residuals, ALL requested at the top of gemm_epilogue (one exposed round trip instead of one per row group: a row group's half
// microsecond of work hides nothing of a cold line's two).
template <typename T, int MI, int NJ, bool HR, bool HG, bool HX>
__device__ __forceinline__ void gemm_epilogue_lean(const GemmArgs& p, f32x4 (&acc)[MI][NJ], const int wrow0, const int wcol0, const int wcu, const int n0, const int l15,
                                                   const float* ev_bias, const float* ev_rv, const float* ev_gt, const uint2 (&rall)[HR ? MI * NJ : 1]) {
    T* __restrict__ Cp = (T*)p.C;
    const bool has_rv = HX && p.rowvec != nullptr, has_gt = HX && p.gate != nullptr, has_dup = HX && p.dup_rows != 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (wrow0 + i * 16 >= p.M) {        // wave-uniform (M % 16 == 0): a row block past M
            if constexpr (HG) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            continue;
        }
        const long m = wrow0 + i * 16 + l15;
        T* const crow = Cp + m * p.ldc + wcol0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nl = wcol0 - n0 + 16 * j;
            if (wcu + 16 * j >= p.N) {      // wave-uniform (N % 16 == 0; wcu = the wave's first column): a column block past N
                if constexpr (HG) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const float4 b = *(const float4*)(ev_bias + nl);
            float v[4] = {acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w};
            if (has_rv) { const float4 c = *(const float4*)(ev_rv + nl); v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w; }
            if (has_gt) { const float4 c = *(const float4*)(ev_gt + nl); v[0] *= c.x; v[1] *= c.y; v[2] *= c.z; v[3] *= c.w; }
            if constexpr (HR) { float r[4]; unpack4<T>(rall[HR ? i * NJ + j : 0], r); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
            const uint2 pk = pack4<T>(v[0], v[1], v[2], v[3]);
            if constexpr (HG) { float r[4]; unpack4<T>(pk, r); acc[i][j] = (f32x4){r[0], r[1], r[2], r[3]}; }      // the 16-bit values the consumer will read, parked for the statistics pass
            *(uint2*)(crow + 16 * j) = pk;
            if (has_dup) *(uint2*)(crow + p.dup_rows * p.ldc + 16 * j) = pk;
        }
    }
}

// The same for the MX fp8 output (GemmArgs::C8; bias, optional tanh-GELU as a template parameter): a 32-column block = two adjacent 16-column tiles of
// one row, spread over the 4 lanes g4 = 0..3.  M % 16 == 0 and N % 32 == 0: whole blocks, wave-uniform predicates.
template <typename T, int MI, int NJ, int ACT>
__device__ __forceinline__ void gemm_epilogue_lean_c8(const GemmArgs& p, const f32x4 (&acc)[MI][NJ], const int wrow0, const int wcol0, const int wcu, const int n0,
                                                      const int l15, const int g4, const float* ev_bias) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (wrow0 + i * 16 >= p.M) continue;        // wave-uniform
        const long m = wrow0 + i * 16 + l15;
        char* const drow = (char*)p.C8 + m * p.ldc8 + p.c8_col + wcol0;
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp) {
            if (wcu + 32 * jp >= p.N) continue;     // wave-uniform
            const int nl = wcol0 - n0 + 32 * jp;
            const float4 b0 = *(const float4*)(ev_bias + nl), b1 = *(const float4*)(ev_bias + nl + 16);
            float v[8] = {acc[i][2 * jp][0] + b0.x, acc[i][2 * jp][1] + b0.y, acc[i][2 * jp][2] + b0.z, acc[i][2 * jp][3] + b0.w,
                          acc[i][2 * jp + 1][0] + b1.x, acc[i][2 * jp + 1][1] + b1.y, acc[i][2 * jp + 1][2] + b1.z, acc[i][2 * jp + 1][3] + b1.w};
            float amax = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (ACT == 2) v[r] = gelu_tanh_f(v[r]);
                v[r] = to_f32(from_f32<T>(v[r]));       // the value the 16-bit path would have stored (same input to the quantiser)
                amax = fmaxf(amax, fabsf(v[r]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const int e = mx_scale_e8m0(amax);
            const float inv = mx_inv_scale(e);
            *(uint32_t*)(drow + 32 * jp) = mx_pack4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
            *(uint32_t*)(drow + 32 * jp + 16) = mx_pack4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
            if (g4 == 0) { const int kb = (p.c8_col + wcu + 32 * jp) >> 5; ((uint8_t*)p.SC)[((long)(kb >> 2) * p.sc_ld + m) * 4 + (kb & 3)] = (uint8_t)e; }
        }
    }
}

// Output stage shared by the main loops below: split-K partials, MX fp8 output, or the fused 16-bit / fp32 epilogue.
// Lane (l15, g4) of wave (wm, wn) holds, in acc[i][j][r], row m0 + wm*(BM/WM) + 16 i + l15 and column n0 + wn*(BN/2) + 16 j + 4 g4 + r.
// LNF: a LayerNorm is folded into this GEMM (GemmArgs::ln_c1): lnm / lnr hold mean and rstd of the lane's MI rows (ln_row_stats).
template <typename T, int BM, int BN, int WM, int MI, int NJ, bool LNF = false, int NR = 1, bool GNOK = true>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p_in, f32x4 (&acc)[MI][NJ], const int m0, const int n0, const int wm, const int wn,
                                              const int l15, const int g4, const int split, const int S,
                                              const float* lnm, const float* lnr, const uint2 (&rpre)[NR], const bool use_rpre) {
#ifdef LDX_EP_TEST_PLAIN        // profiles/ubench/pp_stamp.hip: the output stage with every optional operand compiled out (how much of its time is code size?)
    GemmArgs p = p_in;
    p.rowvec = nullptr; p.gate = nullptr; p.act = 0; p.oscale = 0.f; p.R2 = nullptr; p.Cf = nullptr; p.dup_rows = 0; p.gn_partial = nullptr; p.R = nullptr; p.C8 = nullptr; p.geglu = 0;
    p.N &= ~3; p.splitk = 1;
#else
    const GemmArgs& p = p_in;
#endif
    // ---- split-K: raw fp32 partials to the workspace.  Without p.sk_count the epilogue happens in splitk_reduce_kernel (a second launch).
    // With it (round 4; gemm_sk_fixup) the LAST workgroup to finish a tile reduces in place and carries on into the fused epilogue below — no
    // reduce launch, and the tile feeds the consumer GroupNorm's statistics like any other.  The recipe is the guide's counter form of the
    // agent-scope hand-off (cdna_hip_programming.md section 5, "In-launch split-K reduction"), in its write-through variant:
    //   * every split workgroup writes its accumulators to a PRIVATE slab of the (tile, split) pair, lane-linear — each wave-instruction is one
    //     contiguous 1 KiB store of 16 bytes per lane — with sc1 (write-through) stores: they reach memory without any L2 write-back fence.
    //     (Round 3 tried the row-major [S][M][N] layout with 8-byte agent-scope atomics — 2.7x the cost per byte and four partial lines per
    //     instruction: 16.4 -> 17.1 ms per step — and a __threadfence() pair, which writes back and invalidates a whole XCD L2 per workgroup.)
    //   * every wave waits for its stores' acknowledgements (vmcnt(0)), the workgroup meets at a barrier, one lane takes a relaxed agent-scope
    //     ticket; the workgroup that draws S - 1 resets the counter and reads ALL S slabs (its own too: the sum does not depend on who is last)
    //     in split order with sc1 loads, 16 bytes per lane, MI * NJ loads in flight per slab.
    // Placement-independent: nothing assumes which XCD a split runs on.
    // Only built with -DLDX_SK_FIXUP_BUILD: compiled into every instantiation the path cost the register-staged kernels 16-73 VGPRs (128 x 128: 176 -> 249,
    // 128 x 160: 124-132 bytes of scratch, 64 x 64: five -> four waves per SIMD) and the VAE decode 3 ms (14.8 -> 17.8), for a scheme that measured slower.
    if (S > 1) {
#ifdef LDX_SK_FIXUP_BUILD
        if (p.sk_count == nullptr)
#endif
        {
            float* __restrict__ ws = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (BM / WM) + i * 16 + l15;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = n0 + wn * (BN / 2) + j * 16 + 4 * g4;
                    if (n + 3 < p.N) *(float4*)(ws + (size_t)m * p.N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    else for (int r = 0; r < 4 && n + r < p.N; ++r) ws[(size_t)m * p.N + n + r] = acc[i][j][r];
                }
            }
            return;
        }
#ifdef LDX_SK_FIXUP_BUILD
        extern __shared__ __attribute__((aligned(16))) char smem_ep[];
        const int tile = (m0 / BM) * ((p.N + BN - 1) / BN) + n0 / BN;
        constexpr int SLAB = BM * BN;                                   // floats
        const int lane_off = (((wm * 2 + wn) * MI * NJ) * 64 + (g4 * 16 + l15)) * 16;      // bytes; + (i * NJ + j) * 1024
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ws + ((size_t)tile * S + split) * SLAB), 0, SLAB * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, acc[i][j]), rs, lane_off + (i * NJ + j) * 1024, 0, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's slab stores are acknowledged
        volatile int& sk_last = *(volatile int*)(smem_ep + 8192);      // in the (idle) staging ring, clear of the statistics stage's first 5 KiB
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.sk_count + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old == (unsigned)(S - 1);
            if (last) __hip_atomic_store(p.sk_count + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left at zero for the next launch
            sk_last = last;
        }
        __syncthreads();
        if (!sk_last) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < S; ++sp) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ws + ((size_t)tile * S + sp) * SLAB), 0, SLAB * 4, 0x00020000);
            f32x4 t[MI][NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    t[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + (i * NJ + j) * 1024, 0, 16 /* sc1 */));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] += t[i][j];
        }
#endif
    }
    // ---- per-column operands through LDS (round 6).  Memory operations of a wave retire in issue order and ONE counter (vmcnt) covers loads and
    // stores on this target, so a global load issued behind a store cannot be consumed before that store is acknowledged: the output loop "load
    // bias / gate / residual of a 16 x 16 tile, compute, store" paid one memory round trip per tile — measured with wall-clock stamps inside the kernel
    // (profiles/ubench/pp_stamp.hip): 10 us for a 256 x 160 tile, 17 us for 256 x 224, 24 us for 256 x 256, i.e. a third of a K = 3072 Flux tile's life
    // and half of a K = 640 UNet tile's.  Now the folded-LayerNorm c1, bias, per-image vector and gate of the tile's BN columns are fetched once by
    // the first BN threads into the (dead) operand ring and read back with ds_read (its own counter), and a row group's residuals are requested
    // together in front of its stores.  Same float operations in the same order per element.
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : 1;
    const int wrow0 = m0 + wm * (BM / WM), wcol0 = n0 + wn * (BN / 2) + 4 * g4;
    // The lean path (gemm_epilogue_lean) and its residual prefetch.  Who takes it: every epilogue made of bias [+ per-image vector] [* gate] [+ residual] [+ second
    // store] [+ statistics] on whole 16 x 16 blocks whose tile lies in one image.  Who does not: LayerNorm-folded tiles (LNF), the register-staged MX kernels (GNOK
    // false and BM < 256: their K loop leaves no register — one spilled VGPR at 128 x 160), GroupNorm-producing tiles wider than 160 columns (GNW below) and 256-wide
    // tiles with a residual (128 accumulators + 64 residual registers spill).  Residual rows: NR > 1 = the caller requested them ahead of its K loop (rpre);
    // otherwise (RALL) they are all requested here, at the top of the stage.
    constexpr bool LEAN_OK = !LNF && (GNOK || BM == 256);
    const bool lean = LEAN_OK && !p.ep_general && p.C && p.bias && !p.C8 && !p.geglu && p.act == 0 && p.oscale == 0.f && !p.R2 && !p.Cf && ((p.M | p.N) & 15) == 0 &&
                      !((p.rowvec || p.gate) && m0 / rpb != (min(m0 + BM, p.M) - 1) / rpb) && !(MI * NJ > 20 && p.gn_partial) && (!p.R || (NR > 1 ? use_rpre : MI * NJ <= 28));
    constexpr bool RALL = LEAN_OK && NR == 1 && MI * NJ <= 28;
    constexpr bool RANY = RALL || (!LNF && NR == MI * NJ);
    uint2 rall[RALL ? MI * NJ : 1];
    if constexpr (RALL) {
        if (lean && p.R) {
            const T* __restrict__ Rq = (const T*)p.R;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    rall[i * NJ + j] = (wrow0 + i * 16 < p.M && n0 + wn * (BN / 2) + 16 * j < p.N) ? *(const uint2*)(Rq + (long)(wrow0 + i * 16 + l15) * p.ldr + wcol0 + 16 * j) : make_uint2(0u, 0u);
        }
    }
    if ((p.rowvec || p.gate) && m0 / rpb != (min(m0 + BM, p.M) - 1) / rpb) {
        // the tile straddles images (tiny latents only): per-lane vectors, one element at a time.  Never GEGLU / MX output (no per-image vector) or a
        // GroupNorm producer (gemm_gn_fuse plans whole tiles of one image).
        const T* __restrict__ Rs = (const T*)p.R;
        T* __restrict__ Cs = (T*)p.C;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wrow0 + i * 16 + l15;
            if (m >= p.M) continue;
            const float* rv = p.rowvec ? p.rowvec + (long)(m / rpb) * p.rowvec_ld : nullptr;
            const float* gt = p.gate ? p.gate + (long)(m / rpb) * p.gate_ld : nullptr;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = wcol0 + 16 * j;
                for (int r = 0; r < 4 && n + r < p.N; ++r) {
                    float x = acc[i][j][r];
                    if (LNF) x = lnr[i] * (x - lnm[i] * p.ln_c1[n + r]);
                    if (p.bias) x += p.bias[n + r];
                    if (rv) x += rv[n + r];
                    if (p.act == 1) x = x / (1.0f + __expf(-1.702f * x));
                    else if (p.act == 2) x = gelu_tanh_f(x);
                    else if (p.act == 3) x = x > 0.f ? x : 0.2f * x;
                    if (gt) x *= gt[n + r];
                    if (p.oscale != 0.f) x *= p.oscale;
                    if (Rs) x += to_f32(Rs[(long)m * p.ldr + n + r]);
                    if (p.R2) x = fmaf(x, p.oscale2, to_f32(((const T*)p.R2)[(long)m * p.ldr2 + n + r]));
                    if (Cs) { Cs[(long)m * p.ldc + n + r] = from_f32<T>(x); if (p.dup_rows) Cs[((long)m + p.dup_rows) * p.ldc + n + r] = from_f32<T>(x); }
                    if (p.Cf) p.Cf[(long)m * p.ldcf + n + r] = x;
                }
            }
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_ev[];
    // [4][BN] floats, 16 KiB into the ring (clear of the statistics stages' first 8 KiB); indexed by the absolute column
    const float* const ev_c1 = (const float*)(smem_ev + 16384);
    const float* const ev_bias = ev_c1 + BN, * const ev_rv = ev_c1 + 2 * BN, * const ev_gt = ev_c1 + 3 * BN;
    {
        float* const ev = (float*)(smem_ev + 16384);
        const long img_row = (p.rowvec || p.gate) ? (long)(m0 / rpb) : 0;
        __syncthreads();            // every wave is past its last fragment read; no DMA is in flight
        for (int t = threadIdx.x; t < BN; t += WM * 128) {
            const int n = n0 + t;
            const bool ok = n < p.N;
            if (LNF) ev[t] = ok ? p.ln_c1[n] : 0.f;
            if (p.bias) ev[BN + t] = ok ? p.bias[n] : 0.f;
            if (p.rowvec) ev[2 * BN + t] = ok ? p.rowvec[img_row * p.rowvec_ld + n] : 0.f;
            if (p.gate) ev[3 * BN + t] = ok ? p.gate[img_row * p.gate_ld + n] : 0.f;
        }
        __syncthreads();
    }
    LDX_PP_STAMP(5);
    // ---- MX fp8 output: a 32-column block = two adjacent 16-column tiles of one row, spread over the 4 lanes g4 = 0..3 ----
    if (p.C8) {
        if constexpr ((BN / 2) % 32 == 0) {
            if (p.bias && !p.ep_general && (p.act == 0 || p.act == 2) && (p.M & 15) == 0 && (p.N & 31) == 0) {
                if (p.act == 2) gemm_epilogue_lean_c8<T, MI, NJ, 2>(p, acc, wrow0, wcol0, n0 + wn * (BN / 2), n0, l15, g4, ev_bias);
                else gemm_epilogue_lean_c8<T, MI, NJ, 0>(p, acc, wrow0, wcol0, n0 + wn * (BN / 2), n0, l15, g4, ev_bias);
                return;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (BM / WM) + i * 16 + l15;
                const bool live = m < p.M;                      // no early exit: every lane takes part in the exchanges
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    const int n = n0 + wn * (BN / 2) + jp * 32 + 4 * g4;
                    const bool nok = n < p.N;                   // N % 32 == 0: a block is inside or outside as a whole
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * jp][r]; v[4 + r] = acc[i][2 * jp + 1][r]; }
                    if (p.bias && nok) {
                        const float4 b0 = *(const float4*)(ev_bias + (n - n0)), b1 = *(const float4*)(ev_bias + (n - n0) + 16);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                    float amax = 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        if (p.act == 1) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
                        else if (p.act == 2) v[r] = gelu_tanh_f(v[r]);
                        else if (p.act == 3) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
                        v[r] = to_f32(from_f32<T>(v[r]));       // the value the 16-bit path would have stored (same input to the quantiser)
                        amax = fmaxf(amax, fabsf(v[r]));
                    }
                    amax = fmaxf(amax, __shfl_xor(amax, 16));
                    amax = fmaxf(amax, __shfl_xor(amax, 32));
                    const int e = mx_scale_e8m0(amax);
                    const float inv = mx_inv_scale(e);
                    if (live && nok) {
                        char* dst = (char*)p.C8 + (long)m * p.ldc8 + p.c8_col + n;
                        *(uint32_t*)dst = mx_pack4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
                        *(uint32_t*)(dst + 16) = mx_pack4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
                        if (g4 == 0) { const int kb = (p.c8_col + n) >> 5; ((uint8_t*)p.SC)[((long)(kb >> 2) * p.sc_ld + m) * 4 + (kb & 3)] = (uint8_t)e; }
                    }
                }
            }
        }
        return;
    }
    // ---- epilogue: lane holds rows m = .. + l15, 4 consecutive columns n = .. + 4*g4 + r ----
    const T* __restrict__ Rp = (const T*)p.R;
    T* __restrict__ Cp = (T*)p.C;
    // GroupNorm statistics of the stored tile (GemmArgs::gn_partial).  Compiled in only for tiles up to 160 columns: keeping the rounded outputs in
    // the accumulator registers makes all of them live through the output stage, which the 192..256-wide ping-pong tiles (128 accumulators, 212 VGPRs)
    // cannot afford without scratch — those take the column-major stage below (GNW).
    constexpr bool GNS = GNOK && NJ <= 5 && !LNF;
    const bool gn = GNS && p.gn_partial != nullptr;
    constexpr bool GNL = GNS;                       // (parking in the lean path for the wide tiles too — GNOK && !LNF — spills 150..1260 bytes: two definitions of 128 accumulators meet in front of the statistics pass)
    const bool gnl = gn;
    // Wide tiles (192..256 columns, NJ = 6..8; round 5): the statistics come from a COLUMN-major output stage instead — for each column tile the lane
    // walks its MI rows, stores them and adds the rounded values in-lane, so an accumulator dies as soon as its column tile is done and nothing is parked
    // (same sums in the same order as the path above: in-lane over i, DPP over the 16 lanes of a column, LDS over the wave rows).  gemm_gn_fuse only plans
    // it for whole tiles (M % BM == 0 within an image, N % BN == 0), so there is no ragged path here.  The VAE's 256 / 512-channel convs are the users.
    constexpr bool GNW = GNOK && NJ > 5 && NJ <= 8 && !LNF && NR == 1;
    if constexpr (GNW) if (p.gn_partial != nullptr) {
        extern __shared__ __attribute__((aligned(16))) char smem_epw[];
        float* red = (float*)smem_epw;                  // [WM][BN][2]; the operand tiles are dead behind the barrier
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + 4 * g4;
            float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
            const float4 bz = p.bias ? *(const float4*)(ev_bias + (n - n0)) : make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 rj[MI], r2j[MI];          // this column tile's residuals, requested together in front of its stores
            if (Rp) {
#pragma unroll
                for (int i = 0; i < MI; ++i) rj[i] = *(const uint2*)(Rp + (long)(wrow0 + i * 16 + l15) * p.ldr + n);
            }
            if (p.R2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) r2j[i] = *(const uint2*)((const T*)p.R2 + (long)(wrow0 + i * 16 + l15) * p.ldr2 + n);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const long m = m0 + wm * (BM / WM) + i * 16 + l15;
                float v[4] = {acc[i][j][0] + bz.x, acc[i][j][1] + bz.y, acc[i][j][2] + bz.z, acc[i][j][3] + bz.w};
                if (p.rowvec) { const float4 b = *(const float4*)(ev_rv + (n - n0)); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
                } else if (p.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
                } else if (p.act == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
                }
                if (p.gate) { const float4 b = *(const float4*)(ev_gt + (n - n0)); v[0] *= b.x; v[1] *= b.y; v[2] *= b.z; v[3] *= b.w; }
                if (p.oscale != 0.f) { v[0] *= p.oscale; v[1] *= p.oscale; v[2] *= p.oscale; v[3] *= p.oscale; }
                if (Rp) { float r[4]; unpack4<T>(rj[i], r); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
                if (p.R2) { float r[4]; unpack4<T>(r2j[i], r);
                            v[0] = fmaf(v[0], p.oscale2, r[0]); v[1] = fmaf(v[1], p.oscale2, r[1]); v[2] = fmaf(v[2], p.oscale2, r[2]); v[3] = fmaf(v[3], p.oscale2, r[3]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float xr = to_f32(from_f32<T>(v[r])); gs[r] += xr; gq[r] = fmaf(xr, xr, gq[r]); }
                if (Cp) { const uint2 pk = pack4<T>(v[0], v[1], v[2], v[3]); *(uint2*)(Cp + m * p.ldc + n) = pk; if (p.dup_rows) *(uint2*)(Cp + (m + p.dup_rows) * p.ldc + n) = pk; }
                if (p.Cf) *(float4*)(p.Cf + m * p.ldcf + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { gs[r] = row16_sum(gs[r]); gq[r] = row16_sum(gq[r]); }
            if (l15 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = wn * (BN / 2) + j * 16 + 4 * g4 + r;
                    red[(wm * BN + col) * 2 + 0] = gs[r];
                    red[(wm * BN + col) * 2 + 1] = gq[r];
                }
            }
        }
        __syncthreads();
        const int cpg = p.gn_cpg, tid = threadIdx.x;
        if (tid < (BN / cpg) * 2) {
            const int gl = tid >> 1, st = tid & 1, nc0 = n0 + gl * cpg;
            float a = 0.f;
            for (int w = 0; w < WM; ++w)
                for (int c = 0; c < cpg; ++c) a += red[(w * BN + gl * cpg + c) * 2 + st];
            const int b = m0 / p.gn_hw, chunk = (m0 - b * p.gn_hw) / BM;
            p.gn_partial[(((long)b * p.gn_nchunk + chunk) * p.gn_G + nc0 / cpg) * 2 + st] = a;
        }
        return;
    }
    bool lean_done = false;
    if constexpr (!LNF) {
        if (lean) {
            const bool hx = p.rowvec || p.gate || p.dup_rows;
            const float* const eb = ev_bias, * const er = ev_rv, * const eg = ev_gt;
            const int wcu = n0 + wn * (BN / 2);
            const uint2 none[1] = {make_uint2(0u, 0u)};
#define LDX_LEAN(HR_, HG_, HX_, RQ_) gemm_epilogue_lean<T, MI, NJ, HR_, HG_, HX_>(p, acc, wrow0, wcol0, wcu, n0, l15, eb, er, eg, RQ_)
#define LDX_LEAN_HR(RQ_) do { if (gnl) { if constexpr (GNL) { if (hx) LDX_LEAN(true, true, true, RQ_); else LDX_LEAN(true, true, false, RQ_); } } \
                              else if (hx) LDX_LEAN(true, false, true, RQ_); else LDX_LEAN(true, false, false, RQ_); } while (0)
            if (Rp) {
                if constexpr (RALL) LDX_LEAN_HR(rall);
                else if constexpr (RANY) LDX_LEAN_HR(rpre);
            } else {
                if (gnl) { if constexpr (GNL) { if (hx) LDX_LEAN(false, true, true, none); else LDX_LEAN(false, true, false, none); } }
                else if (hx) LDX_LEAN(false, false, true, none);
                else LDX_LEAN(false, false, false, none);
            }
#undef LDX_LEAN_HR
#undef LDX_LEAN
            lean_done = true;
        }
    }
    if (!lean_done) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + l15;
        if (m >= p.M) continue;
        const bool rv = p.rowvec != nullptr, gt = p.gate != nullptr;
        const float ln_mean = LNF ? lnm[i] : 0.f, ln_rstd = LNF ? lnr[i] : 1.f;
        constexpr bool GEG = BN == 128 || (BN == 256 && !LNF && NR == 1);      // tiles whose wave columns are whole 64-row GEGLU slabs (and that are instantiated for it)
        if (!GEG || !p.geglu) {
            uint2 rb[NJ], r2b[NJ];          // this row group's residuals, requested together in front of its stores
            if (Rp) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = wcol0 + 16 * j;
                    rb[j] = (NR > 1 && use_rpre) ? rpre[NR > 1 ? i * NJ + j : 0] : n + 3 < p.N ? *(const uint2*)(Rp + (long)m * p.ldr + n) : make_uint2(0u, 0u);
                }
            }
            if (p.R2) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int n = wcol0 + 16 * j; r2b[j] = n + 3 < p.N ? *(const uint2*)((const T*)p.R2 + (long)m * p.ldr2 + n) : make_uint2(0u, 0u); }
            }
            // (A) everything in front of the residuals, in place (LDS operands only: no wait on the loads above or on earlier stores)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = wcol0 + 16 * j;
                if (n + 3 >= p.N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (LNF) { const float4 c = *(const float4*)(ev_c1 + (n - n0));
                                 v[0] = ln_rstd * (v[0] - ln_mean * c.x); v[1] = ln_rstd * (v[1] - ln_mean * c.y);
                                 v[2] = ln_rstd * (v[2] - ln_mean * c.z); v[3] = ln_rstd * (v[3] - ln_mean * c.w); }
                if (p.bias) { const float4 b = *(const float4*)(ev_bias + (n - n0)); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if (rv)     { const float4 b = *(const float4*)(ev_rv + (n - n0));   v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
                } else if (p.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
                } else if (p.act == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
                }
                if (gt)     { const float4 b = *(const float4*)(ev_gt + (n - n0));   v[0] *= b.x; v[1] *= b.y; v[2] *= b.z; v[3] *= b.w; }
                if (p.oscale != 0.f) { v[0] *= p.oscale; v[1] *= p.oscale; v[2] *= p.oscale; v[3] *= p.oscale; }
                acc[i][j] = (f32x4){v[0], v[1], v[2], v[3]};
            }
            // (B1) the residuals, consumed together BEFORE the row group's first store: one wait per row group (a wait placed behind a store would
            // also wait for that store)
            if (Rp) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { float r[4]; unpack4<T>(rb[j], r); acc[i][j][0] += r[0]; acc[i][j][1] += r[1]; acc[i][j][2] += r[2]; acc[i][j][3] += r[3]; }
            }
            if (p.R2) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { float r[4]; unpack4<T>(r2b[j], r);
                    acc[i][j][0] = fmaf(acc[i][j][0], p.oscale2, r[0]); acc[i][j][1] = fmaf(acc[i][j][1], p.oscale2, r[1]);
                    acc[i][j][2] = fmaf(acc[i][j][2], p.oscale2, r[2]); acc[i][j][3] = fmaf(acc[i][j][3], p.oscale2, r[3]); }
            }
            // (B2) rounding and stores only
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = wcol0 + 16 * j;
                if (n + 3 >= p.N) continue;
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (GNS) if (gn) acc[i][j] = (f32x4){to_f32(from_f32<T>(v[0])), to_f32(from_f32<T>(v[1])), to_f32(from_f32<T>(v[2])), to_f32(from_f32<T>(v[3]))};      // the 16-bit values the consumer will read, parked in the (dead) accumulator for the statistics pass below
                if (Cp) {      // plain stores: non-temporal ones cost the step 4 % (consumers find the lines in cache)
                    const uint2 pk = pack4<T>(v[0], v[1], v[2], v[3]);
                    *(uint2*)(Cp + (long)m * p.ldc + n) = pk;
                    if (p.dup_rows) *(uint2*)(Cp + ((long)m + p.dup_rows) * p.ldc + n) = pk;       // second half of a shared CFG prefix (GemmArgs::dup_rows)
                }
                if (p.Cf) *(float4*)(p.Cf + (long)m * p.ldcf + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
            // ragged last columns (N % 4 != 0: ESRGAN's 3-channel output): one element at a time
            if (p.N & 3) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = wcol0 + 16 * j;
                    if (n >= p.N || n + 3 < p.N) continue;
                    for (int r = 0; r < 4 && n + r < p.N; ++r) {
                        float x = acc[i][j][r];
                        if (LNF) x = ln_rstd * (x - ln_mean * ev_c1[n - n0 + r]);
                        if (p.bias) x += ev_bias[n - n0 + r];
                        if (rv) x += ev_rv[n - n0 + r];
                        if (p.act == 1) x = x / (1.0f + __expf(-1.702f * x));
                        else if (p.act == 2) x = gelu_tanh_f(x);
                        else if (p.act == 3) x = x > 0.f ? x : 0.2f * x;
                        if (gt) x *= ev_gt[n - n0 + r];
                        if (p.oscale != 0.f) x *= p.oscale;
                        if (Rp) x += to_f32(Rp[(long)m * p.ldr + n + r]);
                        if (p.R2) x = fmaf(x, p.oscale2, to_f32(((const T*)p.R2)[(long)m * p.ldr2 + n + r]));
                        if (Cp) { Cp[(long)m * p.ldc + n + r] = from_f32<T>(x); if (p.dup_rows) Cp[((long)m + p.dup_rows) * p.ldc + n + r] = from_f32<T>(x); }
                        if (p.Cf) p.Cf[(long)m * p.ldcf + n + r] = x;
                    }
                }
            }
        } else if constexpr (GEG) {
            // slab-interleaved GEGLU: within a 64-column slab j in {0,1} = value columns, j+2 = matching gate columns; a wave owns BN / 128 slabs
#pragma unroll
            for (int sl = 0; sl < BN / 128; ++sl)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int na = n0 + wn * (BN / 2) + sl * 64 + j * 16 + 4 * g4;        // GEMM column of 'a'
                const int ng = na + 32;                                // GEMM column of 'g'
                if (ng >= p.N) continue;
                const int no = (n0 + wn * (BN / 2) + sl * 64) / 2 + j * 16 + 4 * g4;  // output column
                float a[4] = {acc[i][4 * sl + j][0], acc[i][4 * sl + j][1], acc[i][4 * sl + j][2], acc[i][4 * sl + j][3]};
                float g[4] = {acc[i][4 * sl + j + 2][0], acc[i][4 * sl + j + 2][1], acc[i][4 * sl + j + 2][2], acc[i][4 * sl + j + 2][3]};
                if (LNF) {
                    const float4 ca = *(const float4*)(ev_c1 + (na - n0)), cg = *(const float4*)(ev_c1 + (ng - n0));
                    a[0] = ln_rstd * (a[0] - ln_mean * ca.x); a[1] = ln_rstd * (a[1] - ln_mean * ca.y); a[2] = ln_rstd * (a[2] - ln_mean * ca.z); a[3] = ln_rstd * (a[3] - ln_mean * ca.w);
                    g[0] = ln_rstd * (g[0] - ln_mean * cg.x); g[1] = ln_rstd * (g[1] - ln_mean * cg.y); g[2] = ln_rstd * (g[2] - ln_mean * cg.z); g[3] = ln_rstd * (g[3] - ln_mean * cg.w);
                }
                if (p.bias) {
                    const float4 ba = *(const float4*)(ev_bias + (na - n0)), bg = *(const float4*)(ev_bias + (ng - n0));
                    a[0] += ba.x; a[1] += ba.y; a[2] += ba.z; a[3] += ba.w;
                    g[0] += bg.x; g[1] += bg.y; g[2] += bg.z; g[3] += bg.w;
                }
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = a[r] * (p.geglu == 2 ? gelu_tanh_f(g[r]) : gelu_erf_f(g[r]));
                if (Cp) *(uint2*)(Cp + (long)m * p.ldc + no) = pack4<T>(v[0], v[1], v[2], v[3]);
                if (p.Cf) *(float4*)(p.Cf + (long)m * p.ldcf + no) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    }
    if constexpr (GNL) if (gnl && (lean_done || NJ <= 5)) {
        // per-column sums over the tile's rows: in-lane over i (above), DPP over the 16 lanes that share a column, LDS over the WM wave rows;
        // then one thread per (group of the tile, statistic) adds its gn_cpg columns in a fixed order -> deterministic, no atomics
        extern __shared__ __attribute__((aligned(16))) char smem_ep[];
        float* red = (float*)smem_ep;                   // [WM][BN][2]; the operand tiles are dead (barrier below)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {                  // one column tile at a time: 8 live sums instead of 8 * NJ
            float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const bool live = m0 + wm * (BM / WM) + i * 16 + l15 < p.M;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float xr = live ? acc[i][j][r] : 0.f; gs[r] += xr; gq[r] = fmaf(xr, xr, gq[r]); }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { gs[r] = row16_sum(gs[r]); gq[r] = row16_sum(gq[r]); }
            if (l15 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = wn * (BN / 2) + j * 16 + 4 * g4 + r;
                    red[(wm * BN + col) * 2 + 0] = gs[r];
                    red[(wm * BN + col) * 2 + 1] = gq[r];
                }
            }
        }
        __syncthreads();
        const int cpg = p.gn_cpg, tid = threadIdx.x;
        if (tid < (BN / cpg) * 2) {
            const int gl = tid >> 1, st = tid & 1, nc0 = n0 + gl * cpg;
            if (nc0 < p.N) {
                float a = 0.f;
                for (int w = 0; w < WM; ++w)
                    for (int c = 0; c < cpg; ++c) a += red[(w * BN + gl * cpg + c) * 2 + st];
                const int b = m0 / p.gn_hw, chunk = (m0 - b * p.gn_hw) / BM;
                p.gn_partial[(((long)b * p.gn_nchunk + chunk) * p.gn_G + nc0 / cpg) * 2 + st] = a;
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int MI, int NJ, bool LNF = false, bool GNOK = true>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[MI][NJ], const int m0, const int n0, const int wm, const int wn,
                                              const int l15, const int g4, const int split, const int S,
                                              const float* lnm = nullptr, const float* lnr = nullptr) {
    const uint2 none[1] = {make_uint2(0u, 0u)};
    gemm_epilogue<T, BM, BN, WM, MI, NJ, LNF, 1, GNOK>(p, acc, m0, n0, wm, wn, l15, g4, split, S, lnm, lnr, none, false);
}

}  // namespace ldx
