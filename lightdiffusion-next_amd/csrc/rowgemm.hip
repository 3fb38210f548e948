// Row-block GEMM for the C = 320 and C = 640 levels of the SD1.5 UNet:  Y[m][0:N) = pro(X[m][0:C)) . W^T + bias (+ R[m][:]),  N = C or 3 C, with the
// prologue pro = identity | LayerNorm | GroupNorm-apply computed by the workgroup that owns the rows (128 at C = 320, 64 at C = 640).  Same scheme as xattn_block.hip /
// ff_block.hip: pro(X) lives in LDS as the B operand of every MFMA, weight rows go from L2 straight into A-operand registers (wave w owns output
// features 40 w .. 40 w + 39 of each 320-wide pass), so the N = K = 320 projections of a transformer block stop paying a LayerNorm / GroupNorm
// launch and a normalised copy of the activation in front of them:
//   LayerNorm + q|k|v projection   (transformer.py:199-204 norm1 + attn1.to_q/k/v; Attention.py:100-113)        pro = 1, N = 960
//   GroupNorm + proj_in            (transformer.py:361-367 norm + proj_in, a 1x1 conv = a GEMM in NHWC)          pro = 2, N = 320
//   attn1.to_out + residual        (transformer.py:205-209)                                                        pro = 0, N = 320, R = Y
//   proj_out + x_in                (transformer.py:372-377), with the next GroupNorm's statistics from the output stage  pro = 0, N = C, gn_out
// The GroupNorm prologue reads the per-(image, chunk, group) partial sums its producer wrote (GemmArgs::gn_partial / splitk_reduce_gn_kernel) and
// folds them exactly as gn_apply_kernel does (8 slices in chunk order, then the slices in order; y = fma(x, rstd * gamma, beta - mean * rstd * gamma)),
// so the normalised 16-bit values — and therefore the GEMM — are the ones the separate launches produce.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"
#include "rowblock_store.h"

namespace ldx {

// Geometry per channel count CC (= K): the workgroup's rows always hold 128 * 320 elements, so that A fits LDS next to the small tables
//   CC = 320: 128 rows, 4 lanes per row in the prologue, 8 query tiles, grid (M / 128, 1)
//   CC = 640:  64 rows, 8 lanes per row,               4 query tiles, grid (M / 64, 2): the two workgroups of a row block own the two halves of
//              the output features (each runs the prologue; every weight row is still fetched by one wave per row block)
template <int CC> struct RgGeom {
    static constexpr int BM = 128 * 320 / CC, QT = BM / 16, LPR = 512 / BM, NH = CC / 320;
    static constexpr int AROW = CC * 2 + 16, ABYTES = BM * AROW;
    static constexpr int TAB = 2 * CC * 4 + 8 * 64 * 4 + 64 * 4;          // scale / shift + fold scratch
    static constexpr int LDS = ABYTES + TAB + RB_STAGE_BYTES;                // + the output stage's tile (rowblock_store.h)
};

template <typename T, int PRO, int CC>
__global__ __launch_bounds__(512, 1) void rowgemm_kernel(const RowGemmArgs p) {
    using G = RgGeom<CC>;
    constexpr int RG_C = CC, RG_BM = G::BM, RG_AROW = G::AROW, RG_ABYTES = G::ABYTES, QT = G::QT, LPR = G::LPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    char* sA = smem;
    float* sSc = (float*)(smem + RG_ABYTES);        // LayerNorm: gamma / beta;  GroupNorm: per-channel scale / shift of this image
    float* sSh = sSc + RG_C;
    float* sRed = sSh + RG_C;                       // [8][64] partial folds, then [64] = mean[32] | rstd[32]
    float* sMR = sRed + 8 * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15_0 = lane & 15, g4_0 = lane >> 4;
    const long m0 = (long)blockIdx.x * RG_BM;
    const T* __restrict__ Xp = (const T*)p.X;
    const T* __restrict__ W = (const T*)p.W;

    const int nown = p.N / G::NH;                       // output features of this workgroup: [blockIdx.y * nown, + nown), in passes of 320 (40 per wave)
    const int npass = nown / 320;
    constexpr int NKS = RG_C / 32, PD = 3;
    uint4 wf[PD + 1][3];
    // Round 6: a wave's loads and stores share one in-order counter, so the first weight fragments of pass p + 1 used to queue behind the output stores of pass p
    // (and those of pass 0 were only requested after the prologue's row loads and barrier).  They are now requested BEFORE that pass's stores / before the prologue.
    auto wload = [&](int wrow0_, int l15_, int g4_, int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int r = 16 * t + l15_;
            wf[slot][t] = (r < 40) ? *(const uint4*)(W + (long)(wrow0_ + r) * RG_C + ks * 32 + g4_ * 8) : make_uint4(0, 0, 0, 0);
        }
    };
#pragma unroll
    for (int ks = 0; ks < PD; ++ks) wload((int)blockIdx.y * nown + wave * 40, l15_0, g4_0, ks, ks);

    // ---- prologue: BM rows -> 16-bit A in LDS ----
    if (PRO == 1) { for (int i = tid; i < RG_C; i += 512) { sSc[i] = p.g[i]; sSh[i] = p.b[i]; } }
    if (PRO == 2) {
        const int b = (int)(m0 / p.HW);                                  // HW % BM == 0: the rows of a workgroup belong to one image
        const int g = tid & 31, st = (tid >> 5) & 1, sl = tid >> 6;       // 8 slices x 32 groups x {sum, sum of squares}
        float a = 0.f;
        {
            const float* pp = p.partial + ((long)b * p.nchunk) * p.G * 2 + g * 2 + st;
            int ck = sl;
            for (; ck + 56 < p.nchunk; ck += 64) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pp[(long)(ck + 8 * u) * p.G * 2];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; ck < p.nchunk; ck += 8) a += pp[(long)ck * p.G * 2];
        }
        sRed[sl * 64 + g * 2 + st] = a;
        __syncthreads();
        if (tid < 32) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) { su += sRed[s8 * 64 + tid * 2]; sq += sRed[s8 * 64 + tid * 2 + 1]; }
            const float n = (float)p.HW * (float)(RG_C / 32);
            const float mean = su / n;
            const float var = fmaxf(sq / n - mean * mean, 0.f);
            sMR[tid] = mean; sMR[32 + tid] = rsqrtf(var + p.eps);
        }
        __syncthreads();
        for (int c = tid; c < RG_C; c += 512) {
            const int g2 = c / (RG_C / 32);
            const float a2 = sMR[32 + g2] * p.g[c];
            sSc[c] = a2; sSh[c] = p.b[c] - sMR[g2] * a2;
        }
    }
    {
        const int row = tid / LPR, part = tid % LPR;
        const long m = m0 + row;
        uint4 raw[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) raw[j] = (m < p.M && !(p.abl & 4)) ? *(const uint4*)(Xp + m * p.ldx + (part + LPR * j) * 8) : make_uint4(0, 0, 0, 0);
        if (PRO == 0) {
#pragma unroll
            for (int j = 0; j < 10; ++j) *(uint4*)(sA + row * RG_AROW + (part + LPR * j) * 16) = raw[j];
        } else {
            float x[80];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                float f[8];
                unpack8<T>(raw[j], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[8 * j + e] = f[e];
            }
            float mean = 0.f, rstd = 1.f;
            if (PRO == 1) {
                float su = 0.f;
#pragma unroll
                for (int e = 0; e < 80; ++e) su += x[e];
                su += dpp_f<0xB1>(su); su += dpp_f<0x4E>(su); if (LPR == 8) su += dpp_f<0x141>(su);
                mean = su * (1.0f / RG_C);
                float sq = 0.f;
#pragma unroll
                for (int e = 0; e < 80; ++e) { const float d = x[e] - mean; sq = fmaf(d, d, sq); }
                sq += dpp_f<0xB1>(sq); sq += dpp_f<0x4E>(sq); if (LPR == 8) sq += dpp_f<0x141>(sq);
                rstd = rsqrtf(sq * (1.0f / RG_C) + p.eps);
            }
            __syncthreads();                             // gamma / beta (LayerNorm) or scale / shift (GroupNorm) in LDS
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int c0 = (part + LPR * j) * 8;
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    f[e] = (PRO == 1) ? fmaf((x[8 * j + e] - mean) * rstd, sSc[c0 + e], sSh[c0 + e]) : fmaf(x[8 * j + e], sSc[c0 + e], sSh[c0 + e]);
                *(uint4*)(sA + row * RG_AROW + c0 * 2) = pack8<T>(f);
            }
        }
    }
    __syncthreads();

    // ---- N / 320 passes: Y^T[320 pass + 40 wave ..][q] = W rows . A^T, + bias (+ residual) ----
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        int l15 = l15_0, g4 = g4_0;                       // opaque per pass: otherwise LICM precomputes every per-lane address of the pass (weights,
        asm volatile("" : "+v"(l15), "+v"(g4));           // residual, stores) ahead of the loop and they spill
        const int wrow0 = (int)blockIdx.y * nown + pass * 320 + wave * 40;
        f32x4 acc[3][QT];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(p.abl & 2))
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + PD < NKS) wload(wrow0, l15, g4, ks + PD, (ks + PD) % (PD + 1));
            V8 af[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) af[qt] = as_v8<T>(*(const uint4*)(sA + (16 * qt + l15) * RG_AROW + (ks * 32 + g4 * 8) * 2));
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const V8 w8 = as_v8<T>(wf[ks % (PD + 1)][t]);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc[t][qt] = mfma16(w8, af[qt], acc[t][qt]);
            }
            __builtin_amdgcn_sched_barrier(0);            // without it hipcc hoists the fragment reads of later k-steps (1 KiB of scratch)
        }
        __builtin_amdgcn_sched_barrier(0);                // the residual loads stay behind the MFMA loop (hoisted above it they cost 48 registers there: scratch)
        uint2 rr[3][QT];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const long m = m0 + 16 * qt + l15;
                const int nl = 16 * t + 4 * g4;
                rr[t][qt] = (p.R && m < p.M && nl < 40 && !(p.abl & 1)) ? *(const uint2*)((const T*)p.R + m * p.ldr + wrow0 + nl) : make_uint2(0u, 0u);
            }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nl = 16 * t + 4 * g4;
            const int n = wrow0 + (nl < 40 ? nl : 0);
            const float4 bo = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float r4[4];
                unpack4<T>(rr[t][qt], r4);
                rr[t][qt] = pack4<T>(acc[t][qt][0] + bo.x + r4[0], acc[t][qt][1] + bo.y + r4[1], acc[t][qt][2] + bo.z + r4[2], acc[t][qt][3] + bo.w + r4[3]);
            }
        }
        if (p.gn_out) {       // statistics of the stored values for the consumer GroupNorm (the layout gn_apply_kernel folds), fixed order: deterministic
            float cs[3][4], cq[3][4];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { cs[t][r] = 0.f; cq[t][r] = 0.f; }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    float v4[4];
                    unpack4<T>(rr[t][qt], v4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { cs[t][r] += v4[r]; cq[t][r] = fmaf(v4[r], v4[r], cq[t][r]); }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { cs[t][r] = row16_sum(cs[t][r]); cq[t][r] = row16_sum(cq[t][r]); }
                const int nl = 16 * t + 4 * g4;
                if (l15 == 0 && nl < 40) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sSc[(wave * 40 + nl + r) * 2] = cs[t][r]; sSc[(wave * 40 + nl + r) * 2 + 1] = cq[t][r]; }
                }
            }
            __syncthreads();
            const int cpg = p.N / 32;
            if (tid < (320 / cpg) * 2) {
                const int gl = tid >> 1, st = tid & 1;
                float a = 0.f;
                for (int c = 0; c < cpg; ++c) a += sSc[(gl * cpg + c) * 2 + st];
                const int bimg = (int)(m0 / p.HW), chunk = (int)((m0 - (long)bimg * p.HW) / RG_BM);
                const int g0 = ((int)blockIdx.y * nown + pass * 320) / cpg;
                p.gn_out[(((long)bimg * p.gn_nchunk + chunk) * 32 + g0 + gl) * 2 + st] = a;
            }
        }
        if (pass + 1 < npass) {                           // the next pass's first weight fragments go out ahead of this pass's stores
#pragma unroll
            for (int ks = 0; ks < PD; ++ks) wload(wrow0 + 320, l15, g4, ks, ks);
        }
        if (!(p.abl & 8)) rb_store_rows<T, QT>((T*)p.Y, p.ldy, m0, p.M, (int)blockIdx.y * nown + pass * 320, wave, l15, g4, tid, rr, smem + RG_ABYTES + G::TAB, p.dup_rows);
    }
}

bool rowgemm_ok(const RowGemmArgs& a) {
    static const bool off = getenv("LDX_ROWGEMM") && atoi(getenv("LDX_ROWGEMM")) == 0;
    static const bool off640 = getenv("LDX_ROWGEMM640") && atoi(getenv("LDX_ROWGEMM640")) == 0;
    if (off || (a.K != 320 && a.K != 640) || (a.K == 640 && off640) || a.N <= 0 || a.N % a.K || a.M <= 0 || a.ldx % 8 || a.ldy % 8 || (a.R && a.ldr % 4) || a.pro < 0 || a.pro > 2) return false;
    const int bm = 128 * 320 / a.K;
    if (a.pro >= 1 && (!a.g || !a.b)) return false;
    if (a.pro == 2 && (!a.partial || a.G != 32 || a.HW % bm || a.M % a.HW || a.nchunk < 1 || a.nchunk > GN_NCHUNK)) return false;
    return true;
}
template <typename T, int PRO, int CC>
static void launch_rowgemm_inst(const RowGemmArgs& a, hipStream_t s) {
    using G = RgGeom<CC>;
    static DevOnce once;
    set_dyn_lds(once, (const void*)rowgemm_kernel<T, PRO, CC>, G::LDS);
    hipLaunchKernelGGL((rowgemm_kernel<T, PRO, CC>), dim3((unsigned)((a.M + G::BM - 1) / G::BM), G::NH), dim3(512), G::LDS, s, a);
}
template <typename T>
static void launch_rowgemm_t(const RowGemmArgs& a, hipStream_t s) {
    if (a.K == 320) {
        if (a.pro == 0) launch_rowgemm_inst<T, 0, 320>(a, s); else if (a.pro == 1) launch_rowgemm_inst<T, 1, 320>(a, s); else launch_rowgemm_inst<T, 2, 320>(a, s);
    } else {
        if (a.pro == 0) launch_rowgemm_inst<T, 0, 640>(a, s); else if (a.pro == 1) launch_rowgemm_inst<T, 1, 640>(a, s); else launch_rowgemm_inst<T, 2, 640>(a, s);
    }
}
void launch_rowgemm(const RowGemmArgs& a0, DType dt, hipStream_t s) {
    static const int abl = getenv("LDX_RG_ABL") ? atoi(getenv("LDX_RG_ABL")) : 0;
    RowGemmArgs a = a0; a.abl = abl;
    if (dt == DT_BF16) launch_rowgemm_t<__bf16>(a, s); else launch_rowgemm_t<_Float16>(a, s);
}

}  // namespace ldx
