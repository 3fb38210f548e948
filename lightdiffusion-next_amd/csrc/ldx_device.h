// Device-side helpers shared by the gfx950 kernels (wave64, MFMA 16x16x32 16-bit inputs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldx {

typedef __attribute__((ext_vector_type(8))) __bf16   bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16   bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float    f32x4;
typedef __attribute__((ext_vector_type(16))) float   f32x16;

template <typename T> struct Vec;
template <> struct Vec<__bf16>   { using v8 = bf16x8; using v4 = bf16x4; };
template <> struct Vec<_Float16> { using v8 = f16x8;  using v4 = f16x4; };

// D(16x16 f32) += A(16x32) * B(32x16).  Lane l supplies A[i = l&15][k = 8*(l>>4) + 0..7] and
// B[k = 8*(l>>4) + 0..7][j = l&15]; result lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// D(32x32 f32) += A(32x16) * B(16x32).  Lane l supplies A[i = l&31][k = 8*(l>>5) + 0..7] and B[k = 8*(l>>5) + 0..7][j = l&31];
// result lane l holds D[i = 8*(r>>2) + 4*(l>>5) + (r&3)][j = l&31], r = 0..15.  Issues every 32.4 cycles = the full 2.5 PFLOP/s,
// whereas 16x16x32 issues every 21.5 cycles (75 %) — profiles/ubench/mfma_rate.hip.
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// MX block-scaled fp8 (OCP e4m3fn): D(16x16 f32) += A(16x128) * B(128x16).  Lane l (i = l&15, g = l>>4) supplies 32 bytes of
// row i: registers 0-3 = k 16g .. 16g+15, registers 4-7 = k 64+16g .. 64+16g+15 (same for B), and ONE E8M0 scale
// (2^(s-127), low byte of sa / sb) that the hardware applies to the 32-element block k = 32g .. 32g+31 of row i — i.e. the
// scale a lane supplies does NOT belong to the bytes it supplies (measured, profiles/ubench/mx_layout.hip).  D layout as mfma16.  Issues at twice the bf16 flop rate (the only way to the 5 PFLOP/s fp8 peak on gfx950).
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ f32x4 mfma16_mx(i32x8 a, i32x8 b, f32x4 c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}

// MX quantisation rule shared by every producer of MX operands (mx.hip header): E8M0 exponent for a block with the given
// amax, the matching inverse scale, and four fp32 -> four e4m3fn bytes (round to nearest even), packed little-endian.
__device__ __forceinline__ int mx_scale_e8m0(float amax) {
    const uint32_t b = __float_as_uint(amax * (1.0f / 448.0f));
    int e = (int)((b >> 23) & 0xff) + ((b & 0x7fffff) ? 1 : 0);
    return min(max(e, 1), 253);
}
__device__ __forceinline__ float mx_inv_scale(int e) { return __uint_as_float((uint32_t)(254 - e) << 23); }
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d) {
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}

union U128 {
    uint4 u;
    bf16x8 b;
    f16x8 h;
    uint2 d[2];
};

template <typename T> __device__ __forceinline__ typename Vec<T>::v8 as_v8(const uint4& u);
template <> __device__ __forceinline__ bf16x8 as_v8<__bf16>(const uint4& u) { U128 x; x.u = u; return x.b; }
template <> __device__ __forceinline__ f16x8 as_v8<_Float16>(const uint4& u) { U128 x; x.u = u; return x.h; }

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// unpack 8 16-bit values held in a uint4
template <typename T> __device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    typename Vec<T>::v8 v = as_v8<T>(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    typename Vec<T>::v8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)f[i];
    union { typename Vec<T>::v8 v; uint4 u; } x; x.v = v; return x.u;
}
template <typename T> __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    typename Vec<T>::v4 v; v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    union { typename Vec<T>::v4 v; uint2 u; } x; x.v = v; return x.u;
}
template <typename T> __device__ __forceinline__ void unpack4(const uint2& u, float (&f)[4]) {
    union { typename Vec<T>::v4 v; uint2 u; } x; x.u = u;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (float)x.v[i];
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf-GELU (cond/Activation.py:31, F.gelu default).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below
// the 16-bit output rounding): one rcp + one exp + 5 FMAs instead of the ~35-instruction branchy libm erff.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float r = fmaf(-poly * t, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// tanh-GELU (nn.GELU(approximate="tanh")): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x * sigmoid(2u)
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));      // v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division: the MX output stage of a Flux MLP tile evaluates 96 of these per lane
}

// Cross-lane reductions on the VALU (DPP within a 16-lane row, v_permlane16/32_swap across rows) instead of
// __shfl_xor's ds_bpermute round trips through the LDS pipe.  All lanes of the wave must be active.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror: the other quad of the 8-lane half
    v += dpp_f<0x140>(v);     // row_mirror: the other half of the 16-lane row
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ float xrow16_sum(float v) {    // + lane ^ 16
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xrow32_sum(float v) {    // + lane ^ 32
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
// sum over groups of LPR consecutive lanes (16, 32 or 64); every lane of the group gets the total
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
    v = row16_sum(v);
    if (LPR >= 32) v = xrow16_sum(v);
    if (LPR >= 64) v = xrow32_sum(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): give each XCD a contiguous
// range of logical tiles so neighbouring tiles (sharing A/W panels) hit the same private L2.
// Bijective for any grid size (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int nx = 8;
    if (nblk < 2 * nx) return bid;
    const int q = nblk / nx, r = nblk % nx;
    const int xcd = bid % nx, idx = bid / nx;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace ldx
