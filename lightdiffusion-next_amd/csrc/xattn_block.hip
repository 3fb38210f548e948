// Cross-attention sub-block of a BasicTransformerBlock as ONE kernel (SD1.5 level 0: C = 320, 8 heads of 40, <= 80 context keys):
//     h += to_out( softmax( to_q(LayerNorm(h)) . k^T ) . v ) + bias            (transformer.py:186-245 attn2, Attention.py:100-124,
//                                                                                AttentionMethods.py:107-150; k | v of the context are
//                                                                                projected once per forward by the engine)
// The unfused plan runs LayerNorm, the q projection (N = K = C GEMM), a 77-key attention and the out projection (N = K = C GEMM with bias and
// residual): four launches and three [M][C] round trips for 13.4 GFLOP at M = 32768.  Here a 512-thread workgroup owns 128 rows of h:
//   phase 0  LayerNorm of the 128 rows (4 lanes per row, row in registers, two-pass statistics) -> 16-bit A in LDS
//   phase 1  wave w = head w:  Q^T[d][q] = Wq[40w + d][:] . A^T   (16x16x32 MFMAs; Wq rows straight from L2 into A-operand registers, four
//            k-steps in flight; A^T fragments = 16-byte LDS reads of A rows)                       -> 3 x 8 accumulator tiles, kept in registers
//   phase 2  per 16-query tile: S^T = K_h Q^T (the accumulators ARE the B operand: a lane's 4 + 4 values of two d-tiles fill its 8 k-slots, and
//            the K fragments are gathered in the same slot order), one-pass softmax over the <= 80 keys (lane-local + v_permlane swaps),
//            O^T = V_h^T P^T (V_h row-major in a per-wave LDS region, transposed by ds_read_b64_tr_b16), O / l -> 16-bit A2 in LDS (over A)
//   phase 3  Y^T[n][q] = Wo[40w + n][:] . A2^T, + bias + residual h, stored in place
// Same roundings as the unfused path (q, P and the attention output are rounded to 16 bit where the separate kernels store them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ldx_device.h"
#include "ldx_kernels.h"
#include "rowblock_store.h"

namespace ldx {

namespace {
typedef __attribute__((ext_vector_type(4))) short xs16x4;
__device__ __forceinline__ uint2 x_lds_read_tr16(const char* p) {
    const xs16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xs16x4*)p);
    union { xs16x4 v; uint2 u; } x; x.v = v; return x.u;
}
__device__ __forceinline__ float x_quad_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float x_quad_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
}  // namespace

#ifdef XA_ABLATE
__device__ int xa_dbg = 0;       // timing ablations (wrong results): 1 no phase 1, 2 no phase 2, 4 no phase 3 MFMAs, 8 no epilogue, 16 no LN loads
#define XA_DBG(bit) (xa_dbg & (bit))
#else
#define XA_DBG(bit) false
#endif
constexpr int XA_C = 320, XA_H = 8, XA_D = 40, XA_BM = 128, XA_MK = 80;
constexpr int XA_AROW = XA_C * 2 + 16;            // 656 B: 16 consecutive rows start in 16 different 16-byte bank groups
constexpr int XA_VROW = 96;                       // bytes per key row of a wave's V_h region (48 d; == 32 mod 64, see AttnCfg)
constexpr int XA_ABYTES = XA_BM * XA_AROW, XA_VBYTES = XA_MK * XA_VROW;
constexpr int XA_LDS = XA_ABYTES + XA_H * XA_VBYTES + 2 * XA_C * 4;       // + gamma / beta

// Q^T / Y^T projection of one wave: acc[t][qt] (lane: q = 16 qt + l15, row 16 t + 4 g4 + r of the wave's 40 weight rows) = W[rows][:] . A^T
template <typename T>
__device__ __forceinline__ void xa_project(const T* __restrict__ W, const int wrow0, const char* sA, const int l15, const int g4, f32x4 (&acc)[3][8]) {
    using V8 = typename Vec<T>::v8;
    constexpr int NKS = XA_C / 32, PD = 3;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) acc[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 wf[PD + 1][3];
    auto wload = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int r = 16 * t + l15;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (r < XA_D) u = *(const uint4*)(W + (long)(wrow0 + r) * XA_C + ks * 32 + g4 * 8);
            wf[slot][t] = u;
        }
    };
#pragma unroll
    for (int ks = 0; ks < PD; ++ks) wload(ks, ks);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + PD < NKS) wload(ks + PD, (ks + PD) % (PD + 1));
        V8 af[8];
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) af[qt] = as_v8<T>(*(const uint4*)(sA + (16 * qt + l15) * XA_AROW + (ks * 32 + g4 * 8) * 2));
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const V8 w8 = as_v8<T>(wf[ks % (PD + 1)][t]);
#pragma unroll
            for (int qt = 0; qt < 8; ++qt) acc[t][qt] = mfma16(w8, af[qt], acc[t][qt]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(512, 1) void xattn_block_kernel(const XAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    char* sA = smem;
    float* sG = (float*)(smem + XA_ABYTES + XA_H * XA_VBYTES);
    float* sBt = sG + XA_C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    char* sV = smem + XA_ABYTES + wave * XA_VBYTES;
    const long m0 = (long)blockIdx.x * XA_BM;
    const int b = (int)(m0 / p.N);                      // N % 128 == 0: a workgroup never straddles two images
    T* __restrict__ Hp = (T*)p.H;

    // ---- phase 0: gamma / beta, this wave's V_h, LayerNorm of 128 rows ----
    for (int i = tid; i < XA_C; i += 512) { sG[i] = p.ln_g[i]; sBt[i] = p.ln_b[i]; }
    {
        const T* __restrict__ Vp = (const T*)p.V + (long)b * p.Mk * p.ldv + wave * XA_D;
        for (int i = lane; i < XA_VBYTES / 16; i += 64) {         // 80 rows x 6 chunks: 5 of data (d 0..39), 1 of zeros; rows >= Mk zero
            const int row = i / 6, ch = i - row * 6;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (row < p.Mk && ch < 5) u = *(const uint4*)(Vp + (long)row * p.ldv + ch * 8);
            *(uint4*)(sV + row * XA_VROW + ch * 16) = u;
        }
    }
    {
        const int row = tid >> 2, part = tid & 3;
        const long m = m0 + row;
        float x[80];
        if (m < p.M && !XA_DBG(16)) {
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
        const float mean = su * (1.0f / XA_C);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 80; ++e) { const float d = x[e] - mean; sq = fmaf(d, d, sq); }
        sq += dpp_f<0xB1>(sq); sq += dpp_f<0x4E>(sq);
        const float rstd = rsqrtf(sq * (1.0f / XA_C) + p.eps);
        __syncthreads();                                 // gamma / beta in LDS
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int c0 = (part + 4 * j) * 8;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf((x[8 * j + e] - mean) * rstd, sG[c0 + e], sBt[c0 + e]);
            *(uint4*)(sA + row * XA_AROW + c0 * 2) = pack8<T>(f);
        }
    }
    __syncthreads();

    // ---- phase 1: Q^T of head `wave` ----
    f32x4 qT[3][8];
    if (!XA_DBG(1)) xa_project<T>((const T*)p.Wq, wave * XA_D, sA, l15, g4, qT);
    else for (int t = 0; t < 3; ++t) for (int qt = 0; qt < 8; ++qt) qT[t][qt] = (f32x4){1.f, 1.f, 1.f, 1.f};

    // K fragments of the head, gathered in the k-slot order of the accumulator-as-B-operand trick:
    //   k-step 0: slots e < 4 -> d = 4 g4 + e (d-tile 0), e >= 4 -> d = 16 + 4 g4 + e - 4 (d-tile 1);  k-step 1: e < 4 -> d = 32 + 4 g4 + e (< 40), rest 0
    V8 kf[5][2];
    {
        const T* __restrict__ Kp = (const T*)p.K + (long)b * p.Mk * p.ldk + wave * XA_D;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
            const int key = 16 * kt + l15;
            uint2 a0 = make_uint2(0, 0), a1 = make_uint2(0, 0), a2 = make_uint2(0, 0);
            if (key < p.Mk) {
                const T* kr = Kp + (long)key * p.ldk;
                a0 = *(const uint2*)(kr + 4 * g4);
                a1 = *(const uint2*)(kr + 16 + 4 * g4);
                if (g4 < 2) a2 = *(const uint2*)(kr + 32 + 4 * g4);
            }
            kf[kt][0] = as_v8<T>(make_uint4(a0.x, a0.y, a1.x, a1.y));
            kf[kt][1] = as_v8<T>(make_uint4(a2.x, a2.y, 0u, 0u));
        }
    }
    // V^T fragments of the head (A operand of PV): k-step j covers keys 32 j .. 32 j + 31; the region holds 80 keys, the last half k-step is zero
    V8 vf[3][3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const char* vp = sV + (j * 32 + g4 * 4 + (l15 >> 2)) * XA_VROW + (dt * 16 + (l15 & 3) * 4) * 2;
            U128 u;
            u.d[0] = x_lds_read_tr16(vp);
            u.d[1] = (j < 2) ? x_lds_read_tr16(vp + 16 * XA_VROW) : make_uint2(0u, 0u);
            vf[dt][j] = as_v8<T>(u.u);
        }
    __syncthreads();                                     // every wave is done with A: the attention output goes over it

    // ---- phase 2: attention of head `wave`, one 16-query tile at a time ----
    const float c = p.scale * 1.44269504088896340736f;
#pragma unroll
    for (int qt = 0; qt < 8; ++qt) {
        if (XA_DBG(2)) break;
        V8 qb0, qb1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qb0[e] = (T)qT[0][qt][e]; qb0[4 + e] = (T)qT[1][qt][e];
            qb1[e] = (g4 < 2) ? (T)qT[2][qt][e] : (T)0.f; qb1[4 + e] = (T)0.f;
        }
        f32x4 s[5];
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
            s[kt] = mfma16(kf[kt][0], qb0, (f32x4){0.f, 0.f, 0.f, 0.f});
            s[kt] = mfma16(kf[kt][1], qb1, s[kt]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kt + 4 * g4 + r >= p.Mk) s[kt][r] = -INFINITY;
                mx = fmaxf(mx, s[kt][r]);
            }
        mx = x_quad_max(mx);
        const float mc = (mx == -INFINITY) ? 0.f : mx * c;
        float pv[6][4], l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = to_f32(from_f32<T>(__builtin_amdgcn_exp2f(fmaf(s[kt][r], c, -mc))));
                pv[kt][r] = e; l += e;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[5][r] = 0.f;
        l = x_quad_sum(l);
        f32x4 o[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            V8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pf[r] = (T)pv[2 * j][r]; pf[4 + r] = (T)pv[2 * j + 1][r]; }
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) o[dt] = mfma16(vf[dt][j], pf, o[dt]);
        }
        const float inv = (l > 0.f) ? 1.0f / l : 0.f;
        const int q = 16 * qt + l15;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int d = 16 * dt + 4 * g4;
            if (d < XA_D) *(uint2*)(sA + q * XA_AROW + (wave * XA_D + d) * 2) = pack4<T>(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
        }
    }
    __syncthreads();

    // ---- phase 3: out projection + bias + residual, in place ----
    f32x4 y[3][8];
    if (!XA_DBG(4)) xa_project<T>((const T*)p.Wo, wave * XA_D, sA, l15, g4, y);
    else for (int t = 0; t < 3; ++t) for (int qt = 0; qt < 8; ++qt) y[t][qt] = (f32x4){1.f, 1.f, 1.f, 1.f};
    if (XA_DBG(8)) return;
    // all 24 residual pieces of the lane in ONE round trip (as load -> add -> store per tile the round trips were serialised: the compiler cannot
    // move a load of h above a store to h).  Fetching them before the projection instead costs 48 registers there: 192 B of scratch, 45 -> 60 us.
    uint2 rr[3][8];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) {
            const long m = m0 + 16 * qt + l15;
            const int nl = 16 * t + 4 * g4;
            rr[t][qt] = (m < p.M && nl < XA_D && !XA_DBG(8)) ? *(const uint2*)(Hp + m * p.ldh + wave * XA_D + nl) : make_uint2(0u, 0u);
        }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nl = 16 * t + 4 * g4;
        const int n = wave * XA_D + (nl < XA_D ? nl : 0);
        const float4 bo = p.bo ? *(const float4*)(p.bo + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) {
            float r4[4];
            unpack4<T>(rr[t][qt], r4);
            rr[t][qt] = pack4<T>(y[t][qt][0] + bo.x + r4[0], y[t][qt][1] + bo.y + r4[1], y[t][qt][2] + bo.z + r4[2], y[t][qt][3] + bo.w + r4[3]);
        }
    }
    // whole 640-byte rows through an LDS tile (rowblock_store.h); the V regions are free: their fragments went to registers before phase 2
    rb_store_rows<T, 8>(Hp, p.ldh, m0, p.M, 0, wave, l15, g4, tid, rr, smem + XA_ABYTES);
}

bool xattn_block_ok(const XAttnArgs& a) {
    static const bool off = getenv("LDX_XATTN_FUSE") && atoi(getenv("LDX_XATTN_FUSE")) == 0;
    return !off && a.C == XA_C && a.heads == XA_H && a.Mk >= 1 && a.Mk <= XA_MK && a.N % XA_BM == 0 && a.M % a.N == 0 && a.ldh % 8 == 0 && a.ldk % 4 == 0 && a.ldv % 8 == 0;
}

template <typename T>
static void launch_xattn_t(const XAttnArgs& a, hipStream_t s) {
    static DevOnce once;
    set_dyn_lds(once, (const void*)xattn_block_kernel<T>, XA_LDS);
#ifdef XA_ABLATE
    static const int dbg_once = []() { const int v = getenv("LDX_XA_DBG") ? atoi(getenv("LDX_XA_DBG")) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(xa_dbg), &v, sizeof(int)); return v; }();
    (void)dbg_once;
#endif
    hipLaunchKernelGGL((xattn_block_kernel<T>), dim3((unsigned)((a.M + XA_BM - 1) / XA_BM)), dim3(512), XA_LDS, s, a);
}
void launch_xattn_block(const XAttnArgs& a, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_xattn_t<__bf16>(a, s); else launch_xattn_t<_Float16>(a, s);
}

}  // namespace ldx
