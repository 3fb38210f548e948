// Row-block GEMM for the C = 320 level of the SD1.5 UNet:  Y[m][0:N) = pro(X[m][0:320)) . W^T + bias (+ R[m][:]),  N = 320 or 960, with the
// prologue pro = identity | LayerNorm | GroupNorm-apply computed by the workgroup that owns the 128 rows.  Same scheme as xattn_block.hip /
// ff_block.hip: pro(X) lives in LDS as the B operand of every MFMA, weight rows go from L2 straight into A-operand registers (wave w owns output
// features 40 w .. 40 w + 39 of each 320-wide pass), so the N = K = 320 projections of a transformer block stop paying a LayerNorm / GroupNorm
// launch and a normalised copy of the activation in front of them:
//   LayerNorm + q|k|v projection   (transformer.py:199-204 norm1 + attn1.to_q/k/v; Attention.py:100-113)        pro = 1, N = 960
//   GroupNorm + proj_in            (transformer.py:361-367 norm + proj_in, a 1x1 conv = a GEMM in NHWC)          pro = 2, N = 320
//   attn1.to_out + residual        (transformer.py:205-209)                                                        pro = 0, N = 320, R = Y
// The GroupNorm prologue reads the per-(image, chunk, group) partial sums its producer wrote (GemmArgs::gn_partial / splitk_reduce_gn_kernel) and
// folds them exactly as gn_apply_kernel does (8 slices in chunk order, then the slices in order; y = fma(x, rstd * gamma, beta - mean * rstd * gamma)),
// so the normalised 16-bit values — and therefore the GEMM — are the ones the separate launches produce.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"

namespace ldx {

constexpr int RG_C = 320, RG_BM = 128;
constexpr int RG_AROW = RG_C * 2 + 16;
constexpr int RG_ABYTES = RG_BM * RG_AROW;
constexpr int RG_LDS = RG_ABYTES + 2 * RG_C * 4 + 8 * 64 * 4 + 64 * 4;

template <typename T, int PRO>
__global__ __launch_bounds__(512, 1) void rowgemm_kernel(const RowGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    char* sA = smem;
    float* sSc = (float*)(smem + RG_ABYTES);        // LayerNorm: gamma / beta;  GroupNorm: per-channel scale / shift of this image
    float* sSh = sSc + RG_C;
    float* sRed = sSh + RG_C;                       // [8][64] partial folds, then [64] = mean[32] | rstd[32]
    float* sMR = sRed + 8 * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const long m0 = (long)blockIdx.x * RG_BM;
    const T* __restrict__ Xp = (const T*)p.X;
    const T* __restrict__ W = (const T*)p.W;

    // ---- prologue: 128 rows -> 16-bit A in LDS ----
    if (PRO == 1) { for (int i = tid; i < RG_C; i += 512) { sSc[i] = p.g[i]; sSh[i] = p.b[i]; } }
    if (PRO == 2) {
        const int b = (int)(m0 / p.HW);                                  // HW % 128 == 0: the rows of a workgroup belong to one image
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
        const int row = tid >> 2, part = tid & 3;
        const long m = m0 + row;
        uint4 raw[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) raw[j] = (m < p.M) ? *(const uint4*)(Xp + m * p.ldx + (part + 4 * j) * 8) : make_uint4(0, 0, 0, 0);
        if (PRO == 0) {
#pragma unroll
            for (int j = 0; j < 10; ++j) *(uint4*)(sA + row * RG_AROW + (part + 4 * j) * 16) = raw[j];
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
                su += dpp_f<0xB1>(su); su += dpp_f<0x4E>(su);
                mean = su * (1.0f / RG_C);
                float sq = 0.f;
#pragma unroll
                for (int e = 0; e < 80; ++e) { const float d = x[e] - mean; sq = fmaf(d, d, sq); }
                sq += dpp_f<0xB1>(sq); sq += dpp_f<0x4E>(sq);
                rstd = rsqrtf(sq * (1.0f / RG_C) + p.eps);
            }
            __syncthreads();                             // gamma / beta (LayerNorm) or scale / shift (GroupNorm) in LDS
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int c0 = (part + 4 * j) * 8;
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
    const int npass = p.N / RG_C;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        const int wrow0 = pass * RG_C + wave * 40;
        constexpr int NKS = RG_C / 32, PD = 3;
        f32x4 acc[3][8];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) acc[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint4 wf[PD + 1][3];
        auto wload = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int r = 16 * t + l15;
                wf[slot][t] = (r < 40) ? *(const uint4*)(W + (long)(wrow0 + r) * RG_C + ks * 32 + g4 * 8) : make_uint4(0, 0, 0, 0);
            }
        };
#pragma unroll
        for (int ks = 0; ks < PD; ++ks) wload(ks, ks);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + PD < NKS) wload(ks + PD, (ks + PD) % (PD + 1));
            V8 af[8];
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) af[qt] = as_v8<T>(*(const uint4*)(sA + (16 * qt + l15) * RG_AROW + (ks * 32 + g4 * 8) * 2));
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const V8 w8 = as_v8<T>(wf[ks % (PD + 1)][t]);
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) acc[t][qt] = mfma16(w8, af[qt], acc[t][qt]);
            }
            __builtin_amdgcn_sched_barrier(0);            // without it hipcc hoists the fragment reads of later k-steps (1 KiB of scratch)
        }
        uint2 rr[3][8];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) {
                const long m = m0 + 16 * qt + l15;
                const int nl = 16 * t + 4 * g4;
                rr[t][qt] = (p.R && m < p.M && nl < 40) ? *(const uint2*)((const T*)p.R + m * p.ldr + wrow0 + nl) : make_uint2(0u, 0u);
            }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nl = 16 * t + 4 * g4;
            if (nl >= 40) continue;
            const int n = wrow0 + nl;
            const float4 bo = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) {
                const long m = m0 + 16 * qt + l15;
                if (m >= p.M) continue;
                float r4[4];
                unpack4<T>(rr[t][qt], r4);
                *(uint2*)((T*)p.Y + m * p.ldy + n) = pack4<T>(acc[t][qt][0] + bo.x + r4[0], acc[t][qt][1] + bo.y + r4[1], acc[t][qt][2] + bo.z + r4[2], acc[t][qt][3] + bo.w + r4[3]);
            }
        }
    }
}

bool rowgemm_ok(const RowGemmArgs& a) {
    static const bool off = getenv("LDX_ROWGEMM") && atoi(getenv("LDX_ROWGEMM")) == 0;
    if (off || a.K != RG_C || a.N <= 0 || a.N % RG_C || a.M <= 0 || a.ldx % 8 || a.ldy % 4 || (a.R && a.ldr % 4) || a.pro < 0 || a.pro > 2) return false;
    if (a.pro >= 1 && (!a.g || !a.b)) return false;
    if (a.pro == 2 && (!a.partial || a.G != 32 || a.HW % RG_BM || a.M % a.HW || a.nchunk < 1 || a.nchunk > GN_NCHUNK)) return false;
    return true;
}
template <typename T>
static void launch_rowgemm_t(const RowGemmArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.M + RG_BM - 1) / RG_BM));
    if (a.pro == 0) { static DevOnce once; set_dyn_lds(once, (const void*)rowgemm_kernel<T, 0>, RG_LDS); hipLaunchKernelGGL((rowgemm_kernel<T, 0>), grid, dim3(512), RG_LDS, s, a); }
    else if (a.pro == 1) { static DevOnce once; set_dyn_lds(once, (const void*)rowgemm_kernel<T, 1>, RG_LDS); hipLaunchKernelGGL((rowgemm_kernel<T, 1>), grid, dim3(512), RG_LDS, s, a); }
    else { static DevOnce once; set_dyn_lds(once, (const void*)rowgemm_kernel<T, 2>, RG_LDS); hipLaunchKernelGGL((rowgemm_kernel<T, 2>), grid, dim3(512), RG_LDS, s, a); }
}
void launch_rowgemm(const RowGemmArgs& a, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_rowgemm_t<__bf16>(a, s); else launch_rowgemm_t<_Float16>(a, s);
}

}  // namespace ldx
