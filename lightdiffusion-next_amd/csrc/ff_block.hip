// Feed-forward sub-block of a BasicTransformerBlock as ONE kernel (SD1.5 level 0: C = 320, inner = 1280):
//     h += W2 . ( a * gelu_erf(g) ) + b2,   [a | g] = W1 . LayerNorm(h) + b1          (transformer.py:19-70 FeedForward, :240-244;
//                                                                                        cond/Activation.py:6-31 GEGLU)
// The unfused plan runs LayerNorm, the GEGLU projection (M x 2560 x 320 GEMM whose 84 MB output is the largest activation of the step) and the
// down projection (K = 1280, bias, residual): 186 us at M = 32768.  Same scheme as xattn_block.hip: a 512-thread workgroup owns 128 rows,
// LayerNorm(h) stays in LDS as the B operand of every first-projection MFMA, weight rows go from L2 straight into A-operand registers
// (four k-steps in flight), and the [128][1280] GEGLU activation never leaves the CU:
//   per chunk of 128 inner features (10 chunks):
//     a) wave w: [a | g]^T[16 + 16][128 q] for inner features 16 w .. 16 w + 15 of the chunk (K = 320)  ->  GEGLU in registers -> 16-bit G[q][i] in LDS
//     b) after one barrier, wave w: Y^T[40 w .. +40][128 q] += W2[:, chunk] . G^T  (K = 128), accumulators persistent over the chunks
//   G is double-buffered, so one barrier per chunk orders "all of G(c) written" before b(c) and "b(c) done" before G(c + 2) is written
//   (a wave writes G(c + 2) only after passing the barrier of chunk c + 1, which every wave reaches after its b(c)).
// W1 is the engine's GEGLU layout: slabs of 64 rows = 32 value rows then the 32 gate rows of the same inner features; same roundings as the
// unfused path (the GEGLU output is rounded to 16 bit where the GEMM would have stored it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"
#include "rowblock_store.h"

namespace ldx {

constexpr int FB_C = 320, FB_I = 1280, FB_BM = 128, FB_CH = 128, FB_NCH = FB_I / FB_CH;
constexpr int FB_AROW = FB_C * 2 + 16;            // 656 B
constexpr int FB_GROW = FB_CH * 2 + 16;           // 272 B
constexpr int FB_ABYTES = FB_BM * FB_AROW, FB_GBYTES = FB_BM * FB_GROW;
constexpr int FB_LDS = FB_ABYTES + 2 * FB_GBYTES + 2 * FB_C * 4;

template <typename T>
__global__ __launch_bounds__(512, 1) void ff_block_kernel(const FFBlockArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    char* sA = smem;
    char* sGb = smem + FB_ABYTES;
    float* sG = (float*)(smem + FB_ABYTES + 2 * FB_GBYTES);
    float* sBt = sG + FB_C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const long m0 = (long)blockIdx.x * FB_BM;
    T* __restrict__ Hp = (T*)p.H;
    const T* __restrict__ W1 = (const T*)p.W1;
    const T* __restrict__ W2 = (const T*)p.W2;

    // ---- LayerNorm of 128 rows -> A (4 lanes per row, row in registers, two-pass statistics) ----
    for (int i = tid; i < FB_C; i += 512) { sG[i] = p.ln_g[i]; sBt[i] = p.ln_b[i]; }
    {
        const int row = tid >> 2, part = tid & 3;
        const long m = m0 + row;
        float x[80];
        if (m < p.M) {
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const uint4 u = *(const uint4*)(Hp + m * p.ldh + (part + 4 * j) * 8);
                float f[8];
                unpack8<T>(u, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[8 * j + e] = f[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 80; ++e) x[e] = 0.f;
        }
        float su = 0.f;
#pragma unroll
        for (int e = 0; e < 80; ++e) su += x[e];
        su += dpp_f<0xB1>(su); su += dpp_f<0x4E>(su);
        const float mean = su * (1.0f / FB_C);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 80; ++e) { const float d = x[e] - mean; sq = fmaf(d, d, sq); }
        sq += dpp_f<0xB1>(sq); sq += dpp_f<0x4E>(sq);
        const float rstd = rsqrtf(sq * (1.0f / FB_C) + p.eps);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int c0 = (part + 4 * j) * 8;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf((x[8 * j + e] - mean) * rstd, sG[c0 + e], sBt[c0 + e]);
            *(uint4*)(sA + row * FB_AROW + c0 * 2) = pack8<T>(f);
        }
    }
    __syncthreads();

    f32x4 y[3][8];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) y[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Weight fragments are fetched a phase ahead of their use: the first PD k-steps of W1 for chunk c + 1 during b(c), the four W2 k-steps of
    // chunk c between a(c)'s MFMAs and its GEGLU arithmetic — otherwise every phase starts on an exposed L2 round trip (20 per workgroup).
    constexpr int NKS = FB_C / 32, PD = 2;
    uint4 w1[PD + 1][2];
    auto w1_rows = [&](int ch, const T*& wv, const T*& wg) __attribute__((always_inline)) {
        const int i0 = ch * FB_CH + wave * 16;                              // first inner feature of the wave in this chunk
        const int vrow = (i0 >> 5) * 64 + (i0 & 31) + l15;                  // its value row; the gate row is 32 further
        wv = W1 + (long)vrow * FB_C + g4 * 8;
        wg = wv + 32 * FB_C;
    };
    {
        const T *wv, *wg;
        w1_rows(0, wv, wg);
#pragma unroll
        for (int ks = 0; ks < PD; ++ks) { w1[ks][0] = *(const uint4*)(wv + ks * 32); w1[ks][1] = *(const uint4*)(wg + ks * 32); }
    }
    auto w2_load = [&](int ch, int ks, uint4 (&w2f)[4][3]) __attribute__((always_inline)) {
        const T* w2 = W2 + (long)(wave * 40) * FB_I + ch * FB_CH + g4 * 8;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int r = 16 * t + l15;
            w2f[ks][t] = (r < 40) ? *(const uint4*)(w2 + (long)r * FB_I + ks * 32) : make_uint4(0, 0, 0, 0);
        }
    };
    for (int ch = 0; ch < FB_NCH; ++ch) {
        char* sGc = sGb + (ch & 1) * FB_GBYTES;
        uint4 w2f[4][3];
        // ---- a) this wave's 16 inner features: value and gate rows of W1 (engine GEGLU layout), K = 320 ----
        {
            const T *wv, *wg;
            w1_rows(ch, wv, wg);
            f32x4 tv[8], tg[8];
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) { tv[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; tg[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + PD < NKS) { w1[(ks + PD) % (PD + 1)][0] = *(const uint4*)(wv + (ks + PD) * 32); w1[(ks + PD) % (PD + 1)][1] = *(const uint4*)(wg + (ks + PD) * 32); }
                V8 af[8];
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) af[qt] = as_v8<T>(*(const uint4*)(sA + (16 * qt + l15) * FB_AROW + (ks * 32 + g4 * 8) * 2));
                const V8 a8 = as_v8<T>(w1[ks % (PD + 1)][0]), g8 = as_v8<T>(w1[ks % (PD + 1)][1]);
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) { tv[qt] = mfma16(a8, af[qt], tv[qt]); tg[qt] = mfma16(g8, af[qt], tg[qt]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            w2_load(ch, 0, w2f);                            // first W2 k-step of this chunk: in flight during the GEGLU arithmetic and the barrier
            // GEGLU: lane holds q = 16 qt + l15, inner features 4 g4 + r of the wave's 16
            const int i0 = ch * FB_CH + wave * 16;
            const int brow = (i0 >> 5) * 64 + (i0 & 31) + 4 * g4;
            const float4 bv = *(const float4*)(p.b1 + brow), bg = *(const float4*)(p.b1 + brow + 32);
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) {
                const float v0 = (tv[qt][0] + bv.x) * gelu_erf_f(tg[qt][0] + bg.x), v1 = (tv[qt][1] + bv.y) * gelu_erf_f(tg[qt][1] + bg.y);
                const float v2 = (tv[qt][2] + bv.z) * gelu_erf_f(tg[qt][2] + bg.z), v3 = (tv[qt][3] + bv.w) * gelu_erf_f(tg[qt][3] + bg.w);
                *(uint2*)(sGc + (16 * qt + l15) * FB_GROW + (wave * 16 + 4 * g4) * 2) = pack4<T>(v0, v1, v2, v3);
            }
        }
        __syncthreads();
        w2_load(ch, 1, w2f); w2_load(ch, 2, w2f); w2_load(ch, 3, w2f);      // the others arrive under the first MFMAs of b) (more of them before the GEGLU: scratch)
        if (ch + 1 < FB_NCH) {       // first W1 k-steps of the next chunk: in flight during b)
            const T *wv, *wg;
            w1_rows(ch + 1, wv, wg);
#pragma unroll
            for (int ks = 0; ks < PD; ++ks) { w1[ks][0] = *(const uint4*)(wv + ks * 32); w1[ks][1] = *(const uint4*)(wg + ks * 32); }
        }
        // ---- b) Y^T[40 wave ..][q] += W2[:, chunk] . G^T, K = 128 ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            V8 gf[8];
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) gf[qt] = as_v8<T>(*(const uint4*)(sGc + (16 * qt + l15) * FB_GROW + (ks * 32 + g4 * 8) * 2));
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const V8 w8 = as_v8<T>(w2f[ks][t]);
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) y[t][qt] = mfma16(w8, gf[qt], y[t][qt]);
            }
            __builtin_amdgcn_sched_barrier(0);            // keep the next k-step's eight fragment reads from being hoisted over these MFMAs (registers)
        }
    }

    // ---- epilogue: + b2 + residual h, in place (all 24 residual pieces of the lane in one round trip) ----
    uint2 rr[3][8];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) {
            const long m = m0 + 16 * qt + l15;
            const int nl = 16 * t + 4 * g4;
            rr[t][qt] = (m < p.M && nl < 40) ? *(const uint2*)(Hp + m * p.ldh + wave * 40 + nl) : make_uint2(0u, 0u);
        }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nl = 16 * t + 4 * g4;
        const int n = wave * 40 + (nl < 40 ? nl : 0);
        const float4 bo = p.b2 ? *(const float4*)(p.b2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) {
            float r4[4];
            unpack4<T>(rr[t][qt], r4);
            rr[t][qt] = pack4<T>(y[t][qt][0] + bo.x + r4[0], y[t][qt][1] + bo.y + r4[1], y[t][qt][2] + bo.z + r4[2], y[t][qt][3] + bo.w + r4[3]);
        }
    }
    __syncthreads();                                     // every wave is done with the last G chunk: the output tile goes over the G buffers
    rb_store_rows<T, 8>(Hp, p.ldh, m0, p.M, 0, wave, l15, g4, tid, rr, sGb);
}

bool ff_block_ok(const FFBlockArgs& a) {
    static const bool off = getenv("LDX_FF_FUSE") && atoi(getenv("LDX_FF_FUSE")) == 0;
    return !off && a.C == FB_C && a.inner == FB_I && a.M > 0 && a.ldh % 8 == 0 && a.b1 != nullptr;
}
template <typename T>
static void launch_ff_t(const FFBlockArgs& a, hipStream_t s) {
    static DevOnce once;
    set_dyn_lds(once, (const void*)ff_block_kernel<T>, FB_LDS);
    hipLaunchKernelGGL((ff_block_kernel<T>), dim3((unsigned)((a.M + FB_BM - 1) / FB_BM)), dim3(512), FB_LDS, s, a);
}
void launch_ff_block(const FFBlockArgs& a, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_ff_t<__bf16>(a, s); else launch_ff_t<_Float16>(a, s);
}

}  // namespace ldx
