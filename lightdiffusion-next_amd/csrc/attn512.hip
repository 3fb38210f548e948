// Flash attention for heads of D = 512 on gfx950: the VAE mid-block AttnBlock (src/Attention/Attention.py:127-178 -> pytorch_attention,
// AttentionMethods.py:175-197: ONE head, D = C = 512, N = h w = 16 384 tokens at 1024^2, 65 536 at 2048^2; VariationalAE.py:532-567 calls it in
// Decoder.forward's mid block, :378-413 in the encoder's).  Until round 5 this ran as GEMM (q k^T) -> softmax_rows -> GEMM (p v) over query chunks with
// the N x N scores going through HBM (1 GB of traffic at 1024^2, 16 GB at 2048^2).  Here the scores never leave the CU.
//
// Shape of the kernel (32x32x16 MFMAs, the transposed formulation of attention.hip: S^T = K Q^T, O^T = V^T P^T, a lane owns one query):
//  * one wave per SIMD (256 threads, __launch_bounds__(256, 1): the whole 512-entry register file): a wave owns 32 queries x all 512 d —
//    Q^T fragments in 128 VGPRs, the 16 O^T tiles in 256 accumulator registers.  At D = 512 the softmax is 16 exponentials per lane against
//    64 MFMAs per 32 keys, so the kernel lives on the matrix pipe and the LDS reads (64 KiB of K / V^T fragments per wave and key block);
//  * key blocks of 32: K [32][512] and V [32][512] row-major as the projections wrote them (no V^T GEMM: the transposing read
//    ds_read_b64_tr_b16 gathers the V^T fragments), double-buffered in 2 x 66.5 KiB of LDS, rows padded to 1040 / 1088 B (odd multiples of 16 / 64 B:
//    conflict-free for the b128 and the transposing reads, same rules as attention.hip);
//  * no register is left for staging, so K / V go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, inline asm as in gemm_pp.inc): one
//    1-KiB instruction per row, 16 per wave and block, issued at the top of the iteration that computes the previous block; rows past Mk use an
//    out-of-range offset and land as zeros;
//  * 128 queries per workgroup is only N / 128 workgroups (128 at 1024^2: half the chip), so the KEYS are split over `nsplit` workgroups per query
//    block when that leaves CUs idle; each writes its un-normalised O (fp32), running maximum and denominator, and attn512_merge_kernel folds them
//    (64 MB of partials at 1024^2 with two splits, against the 1 GB the N x N path moved).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"
#include "attn_pipe_common.h"

namespace ldx {

typedef __attribute__((ext_vector_type(4))) int a5_i32x4;
typedef __attribute__((ext_vector_type(4))) short a5_s16x4;
static __device__ __forceinline__ a5_i32x4 a5_srd(const void* base, long bytes) {
    const unsigned long long q = (unsigned long long)base;
    const int n = (int)(bytes > 0x7fffffffL ? 0x7fffffffL : (bytes > 0 ? bytes : 0));
    return (a5_i32x4){(int)(unsigned)q, (int)((unsigned)(q >> 32) & 0xffffu), n, 0x00020000};
}
// lane l lands at lds + 16 l; M0 is written without being declared (see the note in gemm_pp.inc: nothing else in this kernel lives in M0)
static __device__ __forceinline__ void a5_dma16(const a5_i32x4 rsrc, int voff, int soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
static __device__ __forceinline__ uint2 a5_read_tr16(const char* p) {
    const a5_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) a5_s16x4*)p);
    union { a5_s16x4 v; uint2 u; } x; x.v = v; return x.u;
}

// S^T MFMAs with the accumulator in ARCH VGPRs (inline asm, every operand "v"): as builtins hipcc gives S two of the 16 accumulator-file tiles and swaps the
// displaced O tiles through VGPRs around every block (690 v_accvgpr copies per block in the first build).  The softmax reads S on the VALU, O never leaves
// the accumulator file.  An asm MFMA is opaque to hipcc's hazard recogniser: a5_settle() supplies the wait states between the last MFMA and the first VALU read.
template <typename T> static __device__ __forceinline__ void a5_sacc0(f32x16& d, ap_i32x4 a, ap_i32x4 b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
template <typename T> static __device__ __forceinline__ void a5_sacc(f32x16& d, ap_i32x4 a, ap_i32x4 b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
static __device__ __forceinline__ void a5_settle(f32x16& a, f32x16& b) { if (&a == &b) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a)); else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b)); }      // 16-pass MFMA result -> VALU read

// eight registers of an O tile out of the accumulator file, as asm with "a" inputs: read by plain C++ after the key loop, hipcc decides at the loop exit
// which parts of O to move to VGPRs and spills what does not fit
template <int R0> static __device__ __forceinline__ void a5_get8(const f32x16& t, float (&v)[8]) {
    asm volatile("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\tv_accvgpr_read_b32 %3, %11\n\t"
                 "v_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\tv_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15\n\ts_nop 1"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "a"(t[R0]), "a"(t[R0 + 1]), "a"(t[R0 + 2]), "a"(t[R0 + 3]), "a"(t[R0 + 4]), "a"(t[R0 + 5]), "a"(t[R0 + 6]), "a"(t[R0 + 7]));
}

#ifndef A5_KPF
#define A5_KPF 8
#endif
#ifndef A5_NCH
#define A5_NCH 1
#endif
constexpr int A5_D = 512, A5_KV = 32, A5_QB = 128;
constexpr int A5_KROW = A5_D * 2 + 16, A5_VROW = A5_D * 2 + 64;               // 1040 = 65 x 16 B, 1088 = 17 x 64 B
constexpr int A5_KB = A5_KV * A5_KROW, A5_VB = A5_KV * A5_VROW, A5_STAGE = A5_KB + A5_VB, A5_LDS = 2 * A5_STAGE;      // 136 192 B
constexpr int A5_WS_ROW = A5_D + 4;                                            // floats per (split, query) row of the split workspace: O[512], m, l, pad

// ABL (timing-only builds, wrong results; LDX_ATTN512_ABL, read once): 1 = no LDS-DMA inside the key loop (the two stages keep blocks 0 / 1), 2 = no V^T fragment
// reads / PV MFMAs, 4 = no K fragment reads / QK^T MFMAs
// KPF: K fragments are read KPF k-steps ahead of the MFMA that consumes them; NCH: accumulator chains of the S^T contraction (1 or 2)
template <typename T, int ABL = 0, int KPF = A5_KPF, int NCH = A5_NCH>
__global__ __launch_bounds__(256, 1) void attn512_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h2 = lane >> 5, l15 = lane & 15, g16 = lane >> 4;
    const int nqb = (p.Nq + A5_QB - 1) / A5_QB;
    const int S = p.nsplit > 1 ? p.nsplit : 1;
    int lin = blockIdx.x;
    const int qblk = lin % nqb; lin /= nqb;
    const int split = lin % S, hb = lin / S;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qblk * A5_QB + wave * 32;
    const T* __restrict__ Qp = (const T*)p.Q + (long)b * p.Nq * p.ldq + h * A5_D;
    const T* __restrict__ Kp = (const T*)p.K + (long)b * p.Mk * p.ldk + h * A5_D;
    const T* __restrict__ Vp = (const T*)p.V + (long)b * p.Mk * p.ldv + h * A5_D;
    const float c = p.scale * 1.44269504088896340736f;
    const int nblk_all = (p.Mk + A5_KV - 1) / A5_KV;
    const int kb0 = (int)((long)split * nblk_all / S), kb1 = (int)((long)(split + 1) * nblk_all / S);
    const int nblk = kb1 - kb0;

    // ---- Q^T fragments: lane (query l31, half h2) holds d = 16 ks + 8 h2 .. + 7 for ks = 0 .. 31 ----
    V8 qf[32];
    {
        const int q = q0 + l31;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            uint4 u = make_uint4(0, 0, 0, 0);
            if (q < p.Nq) u = *(const uint4*)(Qp + (long)q * p.ldq + (2 * ks + h2) * 8);
            qf[ks] = as_v8<T>(u);
        }
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) o[dt] = zero16;
    float mrun = -INFINITY, lsum = 0.f;

    // ---- staging: wave w moves rows 8 w .. 8 w + 7 of the K block and of the V block, one 1-KiB LDS-DMA instruction per row ----
    constexpr int OOB = (int)0x80000000;
    const a5_i32x4 rK = a5_srd(Kp, ((long)(p.Mk - 1) * p.ldk + A5_D) * 2);
    const a5_i32x4 rV = a5_srd(Vp, ((long)(p.Mk - 1) * p.ldv + A5_D) * 2);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int blk, int stage) __attribute__((always_inline)) {
        const int key0 = (kb0 + blk) * A5_KV + wave * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = key0 + j;                                  // wave-uniform
            const int voff = key < p.Mk ? lane * 16 : OOB;
            const unsigned dst = lds_base + stage * A5_STAGE + (wave * 8 + j) * A5_KROW;
            const unsigned dsv = lds_base + stage * A5_STAGE + A5_KB + (wave * 8 + j) * A5_VROW;
            a5_dma16(rK, voff, key < p.Mk ? key * p.ldk * 2 : 0, dst);
            a5_dma16(rV, voff, key < p.Mk ? key * p.ldv * 2 : 0, dsv);
        }
    };
    if (nblk > 0) issue(0, 0);
    asm volatile("" : "+v"(qf[0][0]));          // (keeps the Q loads ahead of the first wait in program order)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1;
        if (blk + 1 < nblk && (!(ABL & 1) || blk == 0)) issue(blk + 1, cur ^ 1);
        const char* sK = smem + cur * A5_STAGE;
        const char* sV = sK + A5_KB;

        // ---- S^T[key][query] = K Q^T over d: 32 MFMAs, two accumulator chains (even / odd k-steps), K fragments read 4 k-steps ahead ----
        f32x16 s0, s1;
        if constexpr (ABL & 4) { s0 = zero16; s1 = zero16; asm volatile("" : "+v"(s0), "+v"(s1)); } else
        {
            ap_i32x4 kfr[KPF];
#pragma unroll
            for (int d = 0; d < KPF; ++d) kfr[d] = __builtin_bit_cast(ap_i32x4, *(const uint4*)(sK + l31 * A5_KROW + (2 * d + h2) * 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const ap_i32x4 qb = ap_bits(qf[ks]);
                if (ks == 0) a5_sacc0<T>(s0, kfr[0], qb);
                else if (NCH == 2 && ks == 1) a5_sacc0<T>(s1, kfr[1 % KPF], qb);
                else if (NCH == 2 && (ks & 1)) a5_sacc<T>(s1, kfr[ks % KPF], qb);
                else a5_sacc<T>(s0, kfr[ks % KPF], qb);
                if (ks + KPF < 32) kfr[ks % KPF] = __builtin_bit_cast(ap_i32x4, *(const uint4*)(sK + l31 * A5_KROW + (2 * (ks + KPF) + h2) * 16));
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NCH == 2) a5_settle(s0, s1); else a5_settle(s0, s0);
        }
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = (NCH == 2 || (ABL & 4)) ? s0[r] + s1[r] : s0[r];
        if ((kb0 + blk + 1) * A5_KV > p.Mk) {              // ragged last key block (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((kb0 + blk) * A5_KV + 8 * (r >> 2) + 4 * h2 + (r & 3) >= p.Mk) sv[r] = -INFINITY;
            asm volatile("" ::: "memory");
        }
        // ---- online softmax over the block's 32 keys (16 per half-wave lane) ----
        float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sv[8], sv[9]), fmaxf(sv[10], sv[11])), fmaxf(fmaxf(sv[12], sv[13]), fmaxf(sv[14], sv[15]))));
        {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // LAZY reference maximum: mrun only moves when a block's maximum exceeds it by more than thr log2 units (P <= 2^thr: harmless in fp32 sums
        // and in the 16-bit P, whose precision is relative); rescaling O costs 768 accumulator-file instructions, and with an eager maximum some
        // query of the wave moved it in 4 of 10 blocks at N = 16 384 (measured, N = 16 384: 1.00 -> 0.63 ms, N = 65 536: 13.7 -> 7.8 ms = 1.13 PFLOP/s; profiles/r05/attn512_ablations_*.txt: the rescale, not the MFMAs, was the kernel).
        const float thr = ApT<T>::thr;
        const bool need = (mx - mrun) * c > thr;                     // first block: mrun = -inf -> true
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
            const float mnew = fmaxf(mrun, mx);
            const float alpha = (mnew == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(mrun * c - mnew * c);
            mrun = mnew;
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) { ap_scale_acc8<0>(o[dt], alpha); ap_scale_acc8<8>(o[dt], alpha); }      // asm on the AGPRs: no VGPR use of O on the common path
            lsum *= alpha;
        }
        const float mc = (mrun == -INFINITY) ? 0.f : mrun * c;
        V8 pf[2];
        float part = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            V8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(sv[8 * st + e], c, -mc));
                part += pe;
                f[e] = (T)pe;
            }
            pf[st] = f;
        }
        lsum += part;

        // ---- O^T[d][query] += V^T P^T: 2 k-steps of 16 keys x 16 d tiles ----
        if constexpr (!(ABL & 2))
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const char* vp = sV + (16 * st + 4 * (g16 >> 1) + (l15 >> 2)) * A5_VROW + (dt * 32 + 16 * (g16 & 1) + (l15 & 3) * 4) * 2;
                U128 vf;
                vf.d[0] = a5_read_tr16(vp);
                vf.d[1] = a5_read_tr16(vp + 8 * A5_VROW);
                o[dt] = mfma32(as_v8<T>(vf.u), pf[st], o[dt]);
            }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- finalize ----  (lane ids re-derived from an opaque copy: left alone, hipcc computes the output addresses ahead of the key loop and spills them)
    int lane_e = lane;
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(lane_e));          // also the wait states between the last PV MFMA and the accumulator reads below
    const int l31e = lane_e & 31, h2e = lane_e >> 5;
    const float l = xrow32_sum(lsum);
    const int q = q0 + l31e;
    if (q >= p.Nq) return;
    if (S == 1) {
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        T* __restrict__ Op = (T*)p.O + ((long)b * p.Nq + q) * p.ldo + h * A5_D;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            float lo[8], hi[8];
            a5_get8<0>(o[dt], lo); a5_get8<8>(o[dt], hi);
            *(uint2*)(Op + dt * 32 + 4 * h2e) = pack4<T>(lo[0] * inv, lo[1] * inv, lo[2] * inv, lo[3] * inv);
            *(uint2*)(Op + dt * 32 + 8 + 4 * h2e) = pack4<T>(lo[4] * inv, lo[5] * inv, lo[6] * inv, lo[7] * inv);
            *(uint2*)(Op + dt * 32 + 16 + 4 * h2e) = pack4<T>(hi[0] * inv, hi[1] * inv, hi[2] * inv, hi[3] * inv);
            *(uint2*)(Op + dt * 32 + 24 + 4 * h2e) = pack4<T>(hi[4] * inv, hi[5] * inv, hi[6] * inv, hi[7] * inv);
        }
    } else {
        float* w = p.split_ws + ((long)split * p.B * p.H * p.Nq + (long)hb * p.Nq + q) * A5_WS_ROW;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            float lo[8], hi[8];
            a5_get8<0>(o[dt], lo); a5_get8<8>(o[dt], hi);
            *(float4*)(w + dt * 32 + 4 * h2e) = make_float4(lo[0], lo[1], lo[2], lo[3]);
            *(float4*)(w + dt * 32 + 8 + 4 * h2e) = make_float4(lo[4], lo[5], lo[6], lo[7]);
            *(float4*)(w + dt * 32 + 16 + 4 * h2e) = make_float4(hi[0], hi[1], hi[2], hi[3]);
            *(float4*)(w + dt * 32 + 24 + 4 * h2e) = make_float4(hi[4], hi[5], hi[6], hi[7]);
        }
        if (h2e == 0) { w[A5_D] = mrun; w[A5_D + 1] = l; }
    }
}

// out[q][d] = sum_s w_s O_s[q][d] / sum_s w_s l_s,  w_s = 2^((m_s - max_s m_s) c); one thread per 4 d of a query row
template <typename T>
__global__ __launch_bounds__(256) void attn512_merge_kernel(const AttnArgs p) {
    const long rows = (long)p.B * p.H * p.Nq;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = idx >> 7;                     // 128 threads per row
    const int d = (int)(idx & 127) * 4;
    if (row >= rows) return;
    const float c = p.scale * 1.44269504088896340736f;
    const int S = p.nsplit;
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, p.split_ws[((long)s * rows + row) * A5_WS_ROW + A5_D]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float l = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* w = p.split_ws + ((long)s * rows + row) * A5_WS_ROW;
        const float ms = w[A5_D];
        const float ws = (ms == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((ms - m) * c);
        const float4 v = *(const float4*)(w + d);
        acc.x = fmaf(ws, v.x, acc.x); acc.y = fmaf(ws, v.y, acc.y); acc.z = fmaf(ws, v.z, acc.z); acc.w = fmaf(ws, v.w, acc.w);
        l = fmaf(ws, w[A5_D + 1], l);
    }
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    const long hb = row / p.Nq, q = row - hb * p.Nq;
    const int h = (int)(hb % p.H), b = (int)(hb / p.H);
    T* __restrict__ Op = (T*)p.O + ((long)b * p.Nq + q) * p.ldo + h * A5_D;
    *(uint2*)(Op + d) = pack4<T>(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

bool attn512_ok(const AttnArgs& a) {
    static const bool off = getenv("LDX_ATTN512") && atoi(getenv("LDX_ATTN512")) == 0;
    return !off && a.D == A5_D && !a.causal && !a.bias && !a.O8 && a.Nq > 0 && a.Mk > 0 && a.B > 0 && a.H > 0 && a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0;
}
// key splits for a launch: enough workgroups for the chip (>= 224 of 256 CUs), never more than 8, each split at least 16 key blocks
int attn512_splits(const AttnArgs& a) {
    static const int force = getenv("LDX_ATTN512_SPLITS") ? atoi(getenv("LDX_ATTN512_SPLITS")) : 0;
    const long wgs = (long)((a.Nq + A5_QB - 1) / A5_QB) * a.H * a.B;
    const int nblk = (a.Mk + A5_KV - 1) / A5_KV;
    int s = force > 0 ? force : (int)((255 + wgs) / wgs);
    if (s > 8) s = 8;
    while (s > 1 && nblk / s < 16) --s;
    if (force > 0 && s > nblk) s = nblk;
    return s < 1 ? 1 : s;
}
size_t attn512_ws_floats(const AttnArgs& a, int nsplit) { return nsplit > 1 ? (size_t)nsplit * a.B * a.H * a.Nq * A5_WS_ROW : 0; }

void launch_attn512(const AttnArgs& a0, DType dt, hipStream_t s) {
    AttnArgs a = a0;
    if (a.nsplit > 1 && !a.split_ws) a.nsplit = 1;           // no workspace: one workgroup walks all keys of its query block
    if (a.nsplit < 1) a.nsplit = 1;
    const unsigned grid = (unsigned)(((a.Nq + A5_QB - 1) / A5_QB) * a.nsplit * a.H * a.B);
    const long mrows = (long)a.B * a.H * a.Nq;
    static const int abl = getenv("LDX_ATTN512_ABL") ? atoi(getenv("LDX_ATTN512_ABL")) : 0;
    if (abl && dt == DT_BF16) {
#define A5_ABL_CASE(N) case N: { static DevOnce o##N; set_dyn_lds(o##N, (const void*)attn512_kernel<__bf16, N>, A5_LDS); hipLaunchKernelGGL((attn512_kernel<__bf16, N>), dim3(grid), dim3(256), A5_LDS, s, a); return; }
        switch (abl) { A5_ABL_CASE(1) A5_ABL_CASE(2) A5_ABL_CASE(4) A5_ABL_CASE(6) A5_ABL_CASE(7) default: break; }
#undef A5_ABL_CASE
    }
    static const int var = getenv("LDX_ATTN512_VAR") ? atoi(getenv("LDX_ATTN512_VAR")) : 0;      // experiment: KPF * 10 + NCH
    if (var && dt == DT_BF16) {
#define A5_VAR_CASE(K, C) case K * 10 + C: { static DevOnce o; set_dyn_lds(o, (const void*)attn512_kernel<__bf16, 0, K, C>, A5_LDS); hipLaunchKernelGGL((attn512_kernel<__bf16, 0, K, C>), dim3(grid), dim3(256), A5_LDS, s, a); \
            if (a.nsplit > 1) hipLaunchKernelGGL((attn512_merge_kernel<__bf16>), dim3((unsigned)((mrows * 128 + 255) / 256)), dim3(256), 0, s, a); return; }
        switch (var) { A5_VAR_CASE(4, 2) A5_VAR_CASE(8, 2) A5_VAR_CASE(16, 2) A5_VAR_CASE(4, 1) A5_VAR_CASE(16, 1) A5_VAR_CASE(12, 2) default: break; }
#undef A5_VAR_CASE
    }
    if (dt == DT_BF16) {
        static DevOnce once;
        set_dyn_lds(once, (const void*)attn512_kernel<__bf16>, A5_LDS);
        hipLaunchKernelGGL((attn512_kernel<__bf16>), dim3(grid), dim3(256), A5_LDS, s, a);
        if (a.nsplit > 1) hipLaunchKernelGGL((attn512_merge_kernel<__bf16>), dim3((unsigned)((mrows * 128 + 255) / 256)), dim3(256), 0, s, a);
    } else {
        static DevOnce once;
        set_dyn_lds(once, (const void*)attn512_kernel<_Float16>, A5_LDS);
        hipLaunchKernelGGL((attn512_kernel<_Float16>), dim3(grid), dim3(256), A5_LDS, s, a);
        if (a.nsplit > 1) hipLaunchKernelGGL((attn512_merge_kernel<_Float16>), dim3((unsigned)((mrows * 128 + 255) / 256)), dim3(256), 0, s, a);
    }
}

}  // namespace ldx
