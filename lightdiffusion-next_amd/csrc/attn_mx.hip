// MX fp8 attention for D = 128 heads (Flux joint attention, BlackForest/Flux.py:18-33: 57 calls of [1, 24, 4352, 128] per forward) on the block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 (the only fp8 MFMA that issues at twice the bf16 rate on gfx950) — round 5, VERDICT r4 row N1: until now the
// "fp8" mode of BASELINE config 4 ran QK^T and PV in 16 bit.  Both products now run on e4m3 operands with one E8M0 scale per 32 contraction elements:
//   S^T = K Q^T   contraction over d:    Q, K quantised per (token, head, 32-d block) — exactly ldx_op_mx_quant's format on the [rows][H * 128] matrix
//                                        (qk_norm_rope_mx_kernel: RMSNorm + RoPE + quantisation in one pass, bit-identical to rope -> 16 bit -> mx_quant);
//   O^T = V^T P^T contraction over keys: V quantised per (d, 32-key tile) and stored TRANSPOSED and key-permuted (mx_vt_quant_kernel), P = 2^(s c - m c)
//                                        rounded to e4m3 with the FIXED scale 2^-6 (P <= 2^thr = 4 under the lazy reference maximum, so P 2^6 <= 256 < 448).
// Instruction layout, measured (profiles/ubench/mx_layout32.hip, hypothesis D2 / S1, error 0): lane l (row or column l & 31, half h = l >> 5) supplies
// 32 bytes: registers 0-3 = k 16 h .. 16 h + 15, registers 4-7 = k 32 + 16 h .. 47 + 16 h; the scale a lane supplies covers k 32 h .. 32 h + 31 of its row;
// C / D as v_mfma_f32_32x32x16: lane l holds D[8 (r >> 2) + 4 h + (r & 3)][l & 31].  So, in the transposed formulation of the other attention kernels
// (a lane owns one query), the S^T registers of two 32-key tiles ARE the 32 bytes of a P^T operand once converted, for the key order
//   hardware k (0 .. 63 of a 64-key step)  <->  key 32 (k >> 5) + 8 ((k & 15) >> 2) + 4 ((k >> 4) & 1) + (k & 3),
// which mx_vt_quant_kernel bakes into V^T's byte order; a scale block (k 32 h .. 32 h + 31) is then exactly the 32-key tile 2 s + h: V's blocks are plain
// runs of 32 consecutive keys.
// Kernel: 256 queries per workgroup — eight waves of 32 queries (two per SIMD, the default) or four waves of 64 queries with a skewed schedule —, 128-key blocks,
// K [128][128 B] and V^T [128 d][128 B] tiles + their scale dwords global -> LDS by LDS-DMA (128-B rows, 16-B XOR swizzle on the source side, as gemm_pp.inc),
// double-buffered; per 64-key step 8 + 8 MFMAs of 64 cycles against 64 exponentials per lane: the softmax VALU, not the matrix pipe, bounds it.
// Own parity class (no reference counterpart): tests/test_attn_mx_gpu.py pins the quantisers bit for bit and the attention against
// oracle.mx_attention (the same rule restated in torch) and against fp64 attention of the dequantised operands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"
#include "attn_pipe_common.h"

namespace ldx {

typedef __attribute__((ext_vector_type(4))) int am_i32x4;
static __device__ __forceinline__ am_i32x4 am_srd(const void* base, long bytes) {
    const unsigned long long q = (unsigned long long)base;
    const int n = (int)(bytes > 0x7fffffffL ? 0x7fffffffL : (bytes > 0 ? bytes : 0));
    return (am_i32x4){(int)(unsigned)q, (int)((unsigned)(q >> 32) & 0xffffu), n, 0x00020000};
}
static __device__ __forceinline__ void am_dma16(const am_i32x4 rsrc, int voff, int soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
static __device__ __forceinline__ void am_dma4(const am_i32x4 rsrc, int voff, int soff, unsigned lds) {       // lane l lands at lds + 4 l
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
static __device__ __forceinline__ f32x16 mfma32_mx(i32x8 a, i32x8 b, f32x16 c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}

// S^T MFMAs with the accumulator in ARCH VGPRs (inline asm; see attn512.hip: as builtins hipcc puts S into the accumulator file next to O and moves
// it — and the operand fragments — back and forth: 1152 v_accvgpr copies and 48 scratch accesses per key block in the first build).  The softmax reads S on the
// VALU; O stays in the accumulator file (PV MFMAs are builtins; the rare rescale and the epilogue touch it through asm with "a" operands).
// The scale VGPRs are written by a VALU shift right before: s_nop 1 covers the VALU-write -> MFMA-read wait states hipcc cannot see inside asm.
static __device__ __forceinline__ void am_sacc0(f32x16& d, i32x8 a, i32x8 b, int sa, int sb) {
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
static __device__ __forceinline__ void am_sacc(f32x16& d, i32x8 a, i32x8 b, int sa, int sb) {
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
static __device__ __forceinline__ void am_settle2(f32x16& a, f32x16& b) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b)); }
template <int R0> static __device__ __forceinline__ void am_get8(const f32x16& t, float (&v)[8]) {
    asm volatile("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\tv_accvgpr_read_b32 %3, %11\n\t"
                 "v_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\tv_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15\n\ts_nop 1"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "a"(t[R0]), "a"(t[R0 + 1]), "a"(t[R0 + 2]), "a"(t[R0 + 3]), "a"(t[R0 + 4]), "a"(t[R0 + 5]), "a"(t[R0 + 6]), "a"(t[R0 + 7]));
}

constexpr int AM_D = 128, AM_KB = 128, AM_QB = 256;
constexpr int AM_TILE = 128 * 128;                               // one K or V^T tile: 128 rows x 128 B
constexpr int AM_STAGE = 2 * AM_TILE + 1024, AM_LDS = 2 * AM_STAGE;      // + 128 K-scale dwords + 128 V-scale dwords; 67 584 B
constexpr int AM_PSH = 6;                                        // P is stored as e4m3(P * 2^6) with the hardware scale 2^-6
constexpr float AM_THR = 2.0f;                                   // lazy reference exponent: P <= 2^2

// hardware k of a 64-key step -> key (see the header)
__host__ __device__ constexpr int am_key_of_k(int k) { return 32 * (k >> 5) + 8 * ((k & 15) >> 2) + 4 * ((k >> 4) & 1) + (k & 3); }

// QT = query tiles of 32 per wave.  2: four waves (one per SIMD) x 64 queries, the skewed schedule below.  1: eight waves (two per SIMD) x 32 queries, plain
// QK^T -> softmax -> PV per wave: the partner wave of the SIMD supplies the overlap (each K / V^T fragment then feeds one MFMA instead of two: LDS is far from its limit).
template <typename T, int QT>
__global__ __launch_bounds__(512 / QT, QT == 2 ? 1 : 2) void attn_mx_kernel(const AttnMxArgs p) {
    constexpr int NW = 8 / QT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int OOB = (int)0x80000000;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const int nqb = (p.Nq + AM_QB - 1) / AM_QB;
    const int lin = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int qblk = lin % nqb, hb = lin / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qblk * AM_QB + wave * 32 * QT;
    const float c = p.scale * 1.44269504088896340736f;
    const int nblk = (p.Mk + AM_KB - 1) / AM_KB;

    // ---- Q^T operands: lane (query, half): bytes d = 64 ks + 16 h2 .. + 15 and 64 ks + 32 + 16 h2 .. of the head's 128; one scale dword per (query, head) ----
    i32x8 qf[QT][2];
    int sq[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + 32 * qt + l31;
        const bool ok = q < p.Nq;
        const long row = (long)b * p.Nq + (ok ? q : 0);
        const char* base = (const char*)p.Q8 + row * p.ldq8 + h * AM_D;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 lo = ok ? *(const uint4*)(base + 64 * ks + 16 * h2) : make_uint4(0, 0, 0, 0);
            const uint4 hi = ok ? *(const uint4*)(base + 64 * ks + 32 + 16 * h2) : make_uint4(0, 0, 0, 0);
            qf[qt][ks] = (i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
        sq[qt] = ok ? (int)p.SQ[(long)h * p.sq_ld + row] : 0x7f7f7f7f;
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o[QT][4], osum[QT];
    float mrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = zero16;
        osum[qt] = zero16; mrun[qt] = -INFINITY;
    }

    // ---- staging ----
    const am_i32x4 rK = am_srd((const char*)p.K8 + (long)b * p.Mk * p.ldk8 + h * AM_D, ((long)(p.Mk - 1) * p.ldk8 + AM_D));
    const am_i32x4 rV = am_srd((const char*)p.V8T + ((long)b * p.H + h) * AM_D * p.Lp, (long)AM_D * p.Lp);
    const am_i32x4 rSK = am_srd(p.SK + (long)h * p.sk_ld + (long)b * p.Mk, (long)p.Mk * 4);
    const am_i32x4 rSV = am_srd(p.SV + ((long)b * p.H + h) * (p.Lp / AM_KB) * AM_D, (long)(p.Lp / AM_KB) * AM_D * 4);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // a piece = 8 rows: lane -> row 8 piece + prow, LDS position lane & 7 <- source chunk (lane & 7) ^ swz(row), swz(row) = (row >> 1) & 7: the 16 rows one
    // ds_read_b128 lane group of a 32-row operand touches hold 8 even and 8 odd rows (128-B rows: two rows per 64 banks), and rows of equal parity must sit at
    // different chunk positions.  (row & 7, the 16-row GEMM pattern's swizzle, gave 2-way conflicts on every fragment read: SQ_LDS_BANK_CONFLICT 3x the LDS
    // instruction cycles.)  Pieces of a wave are 4 apart, so (8 piece + prow) >> 1 & 7 = (4 wave + (lane >> 4)) & 7 for all of them.
    const int prow = lane >> 3, gch = ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7)) * 16;
    auto issue = [&](int blk, int stage) __attribute__((always_inline)) {
        const unsigned dst = lds_base + stage * AM_STAGE;
#pragma unroll
        for (int i = 0; i < 16 / NW; ++i) {
            const int pc = wave + NW * i;
            const int key = blk * AM_KB + 8 * pc + prow;
            am_dma16(rK, key < p.Mk ? key * p.ldk8 + gch : OOB, 0, dst + pc * 1024);
            am_dma16(rV, (8 * pc + prow) * p.Lp + blk * AM_KB + gch, 0, dst + AM_TILE + pc * 1024);
        }
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int key = blk * AM_KB + 64 * i + lane; am_dma4(rSK, key < p.Mk ? key * 4 : OOB, 0, dst + 2 * AM_TILE + i * 256); }
        } else if (wave == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) am_dma4(rSV, (blk * AM_KB + 64 * i + lane) * 4, 0, dst + 2 * AM_TILE + 512 + i * 256);
        }
    };
    if (nblk > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // One 64-key step, skewed over the wave's two query tiles so that most of the softmax VALU work has matrix work beside it (the softmax, not the matrix pipe,
    // bounds this kernel: 66 exponentials + ~150 plain VALU per step against 18 MFMAs of 64 cycles; as three fenced phases — QK^T, softmax, PV — the first
    // build ran 275 us at the Flux shape, no faster than the 16-bit kernel):
    //   A  QK^T of query tile 0 (4 MFMAs)
    //   B  QK^T of tile 1 (4 MFMAs), one in front of each quarter of tile 0's softmax
    //   C  PV + row sum of tile 0 (5 MFMAs), one in front of each piece of tile 1's softmax
    //   D  PV + row sum of tile 1 (5 MFMAs)
    // The row sums come from a fifth "d tile" whose V^T operand is all ones (e4m3 1.0, scale 2^0): l = sum of the ROUNDED P, on the matrix pipe, instead of 64
    // VALU adds per step.  S^T MFMAs are asm with VGPR accumulators; everything is pinned in source order with sched_barrier(0).
    const i32x8 ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    // one 128-key block; RAG (compile time): the block holds keys past Mk, whose scores are masked (two copies of the body instead of a run-time test: hipcc
    // turned the test into selects that ran on EVERY block — 130 of 400 VALU instructions per step, PMC)
    auto do_block = [&](const int blk, auto rag_tag) __attribute__((always_inline)) {
        constexpr bool ragged = decltype(rag_tag)::value;
        const int cur = blk & 1;
        if (blk + 1 < nblk) issue(blk + 1, cur ^ 1);
        const char* sK = smem + cur * AM_STAGE;
        const char* sV = sK + AM_TILE;
        const uint32_t* sSK = (const uint32_t*)(sK + 2 * AM_TILE);
        const uint32_t* sSV = sSK + 128;
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {          // NOT unrolled: two copies got different accumulator-file allocations for O, reconciled with 672 v_accvgpr_mov per key block
            // K operands of the step's two key tiles (4 fragments, both query tiles use them) and their scale bytes
            i32x8 kf[2][2];
            int ska[2][2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int krow = 32 * (2 * s + tl) + l31;
                const int skd = (int)sSK[krow];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint4 lo = *(const uint4*)(sK + krow * 128 + (((4 * ks + h2) ^ ((krow >> 1) & 7)) << 4));
                    const uint4 hi = *(const uint4*)(sK + krow * 128 + (((4 * ks + 2 + h2) ^ ((krow >> 1) & 7)) << 4));
                    kf[tl][ks] = (i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                    ska[tl][ks] = skd >> (8 * (2 * ks + h2));
                }
            }
            int sqs[QT][2];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) { sqs[qt][0] = sq[qt] >> (8 * h2); sqs[qt][1] = sq[qt] >> (8 * (2 + h2)); }
            f32x16 S[QT][2];
            auto qk = [&](int qt, int i) __attribute__((always_inline)) {      // MFMA i = 0 .. 3 of query tile qt: (key tile i >> 1, d half i & 1)
                const int tl = i >> 1, ks = i & 1;
                if (ks == 0) am_sacc0(S[qt][tl], kf[tl][0], qf[qt][0], ska[tl][0], sqs[qt][0]);
                else am_sacc(S[qt][tl], kf[tl][1], qf[qt][1], ska[tl][1], sqs[qt][1]);
            };
            float sv[32], off = 0.f;
            i32x8 pk;
            // softmax of one query tile in four pieces: 0 = values + maximum of key tile 0, 1 = key tile 1 + lazy reference update, 2 / 3 = exponentials + e4m3 bytes of tile 0 / 1
            auto sm = [&](int qt, int piece) __attribute__((always_inline)) {
                if (piece == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sv[r] = S[qt][0][r];
                    if constexpr (ragged) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) if (blk * AM_KB + 64 * s + 8 * (e >> 2) + 4 * h2 + (e & 3) >= p.Mk) sv[e] = -INFINITY;
                        asm volatile("" ::: "memory");          // keeps the branch a branch: if-converted, the masking ran on every block (170 of 405 VALU instructions per step, PMC)
                    }
                } else if (piece == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sv[16 + r] = S[qt][1][r];
                    if constexpr (ragged) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) if (blk * AM_KB + 64 * s + 32 + 8 * (e >> 2) + 4 * h2 + (e & 3) >= p.Mk) sv[16 + e] = -INFINITY;
                        asm volatile("" ::: "memory");
                    }
                    float mx = sv[0];
#pragma unroll
                    for (int e = 1; e < 32; ++e) mx = fmaxf(mx, sv[e]);
                    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                    // reference exponent: an INTEGER in the log2 domain (ceil of the scaled maximum when it is taken), so that P = 2^(s c - mref) differs from the
                    // rule's 2^(s c - ceil(max c)) by a power of two and the e4m3 rounding of P is the same bits (a real-valued reference shifts the rounding
                    // grid: first build 2.7e-2 away from the rule, as far as from exact attention).  Lazy: it moves only when a block exceeds it by > thr.
                    const float mxc = mx * c;
                    const bool need = (mxc - mrun[qt]) > AM_THR;                  // first block: mrun = -inf -> true
                    if (__builtin_amdgcn_ballot_w64(need) != 0) {
                        const float mnew = fmaxf(mrun[qt], ceilf(mxc));
                        const float alpha = (mnew == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(mrun[qt] - mnew);
                        mrun[qt] = mnew;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) { ap_scale_acc8<0>(o[qt][dt], alpha); ap_scale_acc8<8>(o[qt][dt], alpha); }
                        ap_scale_acc8<0>(osum[qt], alpha); ap_scale_acc8<8>(osum[qt], alpha);
                    }
                    off = (mrun[qt] == -INFINITY) ? (float)AM_PSH : (float)AM_PSH - mrun[qt];
                } else {
                    const int w0 = 4 * (piece - 2);
#pragma unroll
                    for (int w = w0; w < w0 + 4; ++w) {
                        float pe[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) pe[e] = __builtin_amdgcn_exp2f(fmaf(sv[4 * w + e], c, off));
                        int v = __builtin_amdgcn_cvt_pk_fp8_f32(pe[0], pe[1], 0, false);
                        v = __builtin_amdgcn_cvt_pk_fp8_f32(pe[2], pe[3], v, true);
                        pk[w] = v;
                    }
                }
            };
            // V^T operands of the step's four d tiles: read during phase B (one per softmax quarter), so that no PV MFMA waits out an LDS round trip
            // (PMC on the first skewed build: 31 % of the wave cycles parked in s_waitcnt)
            i32x8 vf[4];
            int sva[4];
            auto vload = [&](int i) __attribute__((always_inline)) {
                const int vrow = 32 * i + l31;
                const int svd = (int)sSV[vrow];
                const uint4 lo = *(const uint4*)(sV + vrow * 128 + (((4 * s + h2) ^ ((vrow >> 1) & 7)) << 4));
                const uint4 hi = *(const uint4*)(sV + vrow * 128 + (((4 * s + 2 + h2) ^ ((vrow >> 1) & 7)) << 4));
                vf[i] = (i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                sva[i] = svd >> (8 * (2 * s + h2));
            };
            auto pv = [&](int qt, const i32x8& pbq, int i) __attribute__((always_inline)) {      // i = 0 .. 3: d tile i; 4: the ones tile (row sums)
                if (i == 4) osum[qt] = mfma32_mx(ones8, pbq, osum[qt], 127, 127 - AM_PSH);
                else o[qt][i] = mfma32_mx(vf[i], pbq, o[qt][i], sva[i], 127 - AM_PSH);
            };
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (QT == 2) {
                // ---- A ----
#pragma unroll
                for (int i = 0; i < 4; ++i) qk(0, i);
                am_settle2(S[0][0], S[0][1]);
                __builtin_amdgcn_sched_barrier(0);
                // ---- B ----
#pragma unroll
                for (int i = 0; i < 4; ++i) { qk(QT - 1, i); vload(i); sm(0, i); __builtin_amdgcn_sched_barrier(0); }
                const i32x8 pb0 = pk;
                am_settle2(S[QT - 1][0], S[QT - 1][1]);
                __builtin_amdgcn_sched_barrier(0);
                // ---- C ----
#pragma unroll
                for (int i = 0; i < 4; ++i) { pv(0, pb0, i); sm(QT - 1, i); __builtin_amdgcn_sched_barrier(0); }
                pv(0, pb0, 4);
                const i32x8 pb1 = pk;
                __builtin_amdgcn_sched_barrier(0);
                // ---- D ----
#pragma unroll
                for (int i = 0; i < 5; ++i) pv(QT - 1, pb1, i);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { qk(0, i); vload(i); }
                am_settle2(S[0][0], S[0][1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) sm(0, i);
                const i32x8 pb0 = pk;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 5; ++i) pv(0, pb0, i);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    };
    const int nfull = p.Mk / AM_KB;
    for (int blk = 0; blk < nfull; ++blk) do_block(blk, std::false_type{});
    if (nblk > nfull) do_block(nfull, std::true_type{});

    // ---- finalize: O / l, 16-bit rows or MX fp8 rows + one scale dword per (row, head) (the d tile dt IS the head's 32-d block dt) ----
    int lane_e = lane;
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(lane_e));           // wait states between the last PV MFMA and the accumulator reads; opaque lane id (attn512.hip)
    const int l31e = lane_e & 31, h2e = lane_e >> 5;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l;
        { float lo[8]; am_get8<0>(osum[qt], lo); l = lo[0]; }        // every row of the ones tile holds the row sum of the rounded P (scale 2^-6 applied by the MFMA)
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        const int q = q0 + 32 * qt + l31e;
        const long row = (long)b * p.Nq + q;
        uint32_t sc = 0;
        char* o8 = p.O8 ? (char*)p.O8 + row * p.ldo8 + h * AM_D : nullptr;
        T* __restrict__ Op = p.O ? (T*)p.O + row * p.ldo + h * AM_D : nullptr;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float v[16];
            { float lo[8], hi[8]; am_get8<0>(o[qt][dt], lo); am_get8<8>(o[qt][dt], hi);
#pragma unroll
              for (int r = 0; r < 8; ++r) { v[r] = lo[r] * inv; v[8 + r] = hi[r] * inv; } }
            if (p.O8) {
                float amax = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { v[r] = to_f32(from_f32<T>(v[r])); amax = fmaxf(amax, fabsf(v[r])); }
                auto am = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
                amax = fmaxf(__uint_as_float(am[0]), __uint_as_float(am[1]));
                const int e = mx_scale_e8m0(amax);
                const float is = mx_inv_scale(e);
                sc |= (uint32_t)e << (8 * dt);
                if (q < p.Nq) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
                        *(uint32_t*)(o8 + dt * 32 + 8 * rq + 4 * h2e) = mx_pack4(v[4 * rq] * is, v[4 * rq + 1] * is, v[4 * rq + 2] * is, v[4 * rq + 3] * is);
                }
            } else if (q < p.Nq) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) *(uint2*)(Op + dt * 32 + 8 * rq + 4 * h2e) = pack4<T>(v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]);
            }
        }
        if (p.O8 && q < p.Nq && h2e == 0) p.SO[(long)h * p.so_ld + row] = sc;
    }
}

// V [rows b L + token][head h at columns 128 h] 16 bit  ->  V8T [B][H][128 d][Lp] e4m3 (within every 64-key step the byte order is the MFMA's hardware k:
// byte k holds key am_key_of_k(k)) + SV [B][H][Lp / 128][128 d] dwords (byte t = E8M0 scale of the 32 keys 32 t .. 32 t + 31 of that 128-key block).  The MX rule of
// mx.hip on blocks of 32 consecutive KEYS of one d: scale = 2^ceil(log2(amax / 448)), e4m3fn round to nearest even; tokens past L hold zeros (scale 2^-126).
template <typename T>
__global__ __launch_bounds__(256) void mx_vt_quant_kernel(const MxVtArgs p) {
    __shared__ __attribute__((aligned(16))) T sv[128][AM_D + 8];           // +8: 16-bit columns of consecutive d are read conflict-free anyway; keeps rows 16-B aligned
    const int tid = threadIdx.x;
    const int blk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const T* __restrict__ V = (const T*)p.V + (long)b * p.L * p.ldv + h * AM_D;
    {   // the block's 128 x 128 values: eight UNCONDITIONAL 16-byte loads per thread issued together (a token past L re-reads the last one and is zeroed
        // by a select; behind `tok < L ? load : 0` hipcc emitted one load + wait per iteration: eight memory latencies in a row per workgroup)
        uint4 u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j, row = i >> 4, ch = i & 15, tok = blk * 128 + row;
            u[j] = *(const uint4*)(V + (long)(tok < p.L ? tok : p.L - 1) * p.ldv + ch * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j, row = i >> 4, ch = i & 15, tok = blk * 128 + row;
            *(uint4*)(&sv[row][ch * 8]) = tok < p.L ? u[j] : make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();
    const int d = tid & 127, s = tid >> 7;                    // this thread: column d, 64-key step s of the block
    float x[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) x[k] = to_f32(sv[64 * s + am_key_of_k(k)][d]);       // hardware-k order
    uint32_t out[16];
    uint32_t sc2 = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                             // hardware block t of the step = k 32 t .. 32 t + 31 = the 32 keys of tile 2 s + t
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) amax = fmaxf(amax, fabsf(x[32 * t + k]));
        const int e = mx_scale_e8m0(amax);
        const float is = mx_inv_scale(e);
        sc2 |= (uint32_t)e << (8 * t);
#pragma unroll
        for (int w = 0; w < 8; ++w) out[8 * t + w] = mx_pack4(x[32 * t + 4 * w] * is, x[32 * t + 4 * w + 1] * is, x[32 * t + 4 * w + 2] * is, x[32 * t + 4 * w + 3] * is);
    }
    char* dst = (char*)p.V8T + (((long)b * p.H + h) * AM_D + d) * p.Lp + blk * 128 + 64 * s;
#pragma unroll
    for (int w = 0; w < 4; ++w) *(uint4*)(dst + 16 * w) = make_uint4(out[4 * w], out[4 * w + 1], out[4 * w + 2], out[4 * w + 3]);
    *(uint16_t*)((char*)(p.SV + (((long)b * p.H + h) * (p.Lp / 128) + blk) * AM_D + d) + 2 * s) = (uint16_t)sc2;
}

// qk_norm_rope_kernel (norm.hip) with the MX quantiser behind it: per-head RMSNorm of q and k (QKNorm, Flux.py:148-200) and RoPE (:73-82), the 16-bit
// rounding the 16-bit path stores, then blocks of 32 d -> e4m3 + E8M0 (4 consecutive threads of 8 elements).  Outputs in ldx_op_mx_quant's format:
// Q8 / K8 [rows][H * 128] bytes, SQ / SK dwords [H][s_ld] (byte j of dword [h][row] = d block j).  The 16-bit q / k columns of QKV are NOT rewritten.
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_mx_kernel(const QkRopeArgs p) {
    const int cpt = 16;                                        // D = 128: 16 chunks of 8 per head
    const long total = (long)p.rows * 2 * p.H * cpt;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < total;
    const long i = live ? idx : total - 1;
    const int ch = (int)(i % cpt);
    long r = i / cpt;
    const int h = (int)(r % p.H); r /= p.H;
    const int which = (int)(r & 1);                            // 0 = q, 1 = k
    const long row = r >> 1;
    const T* __restrict__ ptr = (const T*)p.QKV + row * p.ld + which * (p.H * p.D) + h * p.D + ch * 8;
    float x[8];
    unpack8<T>(*(const uint4*)ptr, x);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
    for (int o = cpt >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rr = rsqrtf(ss / (float)p.D + p.eps);
    const float* sc = (which ? p.kscale : p.qscale) + ch * 8;
    const float4 s0 = *(const float4*)sc, s1 = *(const float4*)(sc + 4);
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const int tok = (int)(row % p.L);
    const long to = (long)tok * (p.D >> 1) + ch * 4;
    const float4 cs = *(const float4*)(p.cosT + to), sn = *(const float4*)(p.sinT + to);
    const float cv[4] = {cs.x, cs.y, cs.z, cs.w}, nv[4] = {sn.x, sn.y, sn.z, sn.w};
    float o[8], amax = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q] * rr * sv[2 * q], bb = x[2 * q + 1] * rr * sv[2 * q + 1];
        o[2 * q] = to_f32(from_f32<T>(cv[q] * a - nv[q] * bb));          // the value the 16-bit path stores
        o[2 * q + 1] = to_f32(from_f32<T>(nv[q] * a + cv[q] * bb));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(o[e]));
    amax = fmaxf(amax, dpp_f<0xB1>(amax));                     // the block's other three chunks: lanes ^ 1, ^ 2 of the aligned quad
    amax = fmaxf(amax, dpp_f<0x4E>(amax));
    const int e8 = mx_scale_e8m0(amax);
    const float is = mx_inv_scale(e8);
    if (!live) return;
    char* y = (char*)(which ? p.K8 : p.Q8) + (row + p.row8) * p.ld8 + h * p.D + ch * 8;
    *(uint2*)y = make_uint2(mx_pack4(o[0] * is, o[1] * is, o[2] * is, o[3] * is), mx_pack4(o[4] * is, o[5] * is, o[6] * is, o[7] * is));
    if ((ch & 3) == 0) *((uint8_t*)((which ? p.SK : p.SQ) + (long)h * p.s8_ld + row + p.row8) + (ch >> 2)) = (uint8_t)e8;
}

bool attn_mx_ok(const AttnMxArgs& a) {
    return a.Q8 && a.K8 && a.V8T && a.SQ && a.SK && a.SV && (a.O || (a.O8 && a.SO)) && a.B > 0 && a.H > 0 && a.Nq > 0 && a.Mk > 0 && a.ldq8 % 16 == 0 && a.ldk8 % 16 == 0 &&
           a.Lp % AM_KB == 0 && a.Lp >= a.Mk && (!a.O || a.ldo % 4 == 0) && (!a.O8 || a.ldo8 % 4 == 0);
}
template <typename T, int QT>
static void launch_attn_mx_inst(const AttnMxArgs& a, unsigned grid, hipStream_t s) {
    static DevOnce once;
    set_dyn_lds(once, (const void*)attn_mx_kernel<T, QT>, AM_LDS);
    hipLaunchKernelGGL((attn_mx_kernel<T, QT>), dim3(grid), dim3(512 / QT), AM_LDS, s, a);
}
void launch_attn_mx(const AttnMxArgs& a, DType dt, hipStream_t s) {
    const unsigned grid = (unsigned)(((a.Nq + AM_QB - 1) / AM_QB) * a.H * a.B);
    // measured at the Flux shape [1, 24, 4352, 128], same box (profiles/r05/attn_mx_variants.txt): QT = 1 (eight waves x 32 queries, two per SIMD) 203 us, QT = 2 (four
    // waves x 64 queries, skewed schedule) 226 us, 16-bit attn128p 282-287 us.  LDX_ATTN_MX_QT=2 selects the other one (read once).
    static const int qt = getenv("LDX_ATTN_MX_QT") ? atoi(getenv("LDX_ATTN_MX_QT")) : 1;
    if (dt == DT_BF16) { if (qt == 1) launch_attn_mx_inst<__bf16, 1>(a, grid, s); else launch_attn_mx_inst<__bf16, 2>(a, grid, s); }
    else { if (qt == 1) launch_attn_mx_inst<_Float16, 1>(a, grid, s); else launch_attn_mx_inst<_Float16, 2>(a, grid, s); }
}
void launch_mx_vt_quant(const MxVtArgs& a, DType dt, hipStream_t s) {
    const dim3 grid((unsigned)(a.Lp / 128), (unsigned)a.H, (unsigned)a.B);
    if (dt == DT_BF16) hipLaunchKernelGGL((mx_vt_quant_kernel<__bf16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mx_vt_quant_kernel<_Float16>), grid, dim3(256), 0, s, a);
}
void launch_qk_norm_rope_mx(const QkRopeArgs& a, DType dt, hipStream_t s) {
    const long total = (long)a.rows * 2 * a.H * 16;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dt == DT_BF16) hipLaunchKernelGGL((qk_norm_rope_mx_kernel<__bf16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((qk_norm_rope_mx_kernel<_Float16>), grid, dim3(256), 0, s, a);
}

}  // namespace ldx
