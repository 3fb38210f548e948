// Helpers shared by the software-pipelined attention kernels (attn_pipe.hip: D = 40, attn_pipe128.hip: D = 128): explicit-register-class MFMAs,
// the softmax asm pieces, compile-time loops.  See attn_pipe.hip for the design notes.
#pragma once
#include <type_traits>
#include <utility>
#include "ldx_device.h"

namespace ldx {

typedef __attribute__((ext_vector_type(4))) short ap_s16x4;
typedef __attribute__((ext_vector_type(4))) int ap_i32x4;
typedef __attribute__((ext_vector_type(2))) int ap_i32x2;
__device__ __forceinline__ ap_i32x2 ap_lds_read_tr16(const char* p) {
    return __builtin_bit_cast(ap_i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ap_s16x4*)p));
}
template <typename V> __device__ __forceinline__ ap_i32x4 ap_bits(const V& v) { return __builtin_bit_cast(ap_i32x4, v); }
template <typename T> struct ApT;
template <> struct ApT<__bf16>   { static constexpr int split = 256, expsh = 7, maxdl = 255; static constexpr float thr = 64.0f; };      // P <= 2^64: N * 2^64 * |v| stays far inside fp32
template <> struct ApT<_Float16> { static constexpr int split = 2048, expsh = 10, maxdl = 31; static constexpr float thr = 15.0f; };

// ---- QK^T MFMAs (32x32x16) with explicit register classes: S' in arch VGPRs, K / Q fragments in AGPRs.  First MFMA of a chain: C = 0.
#define AP_BF16 "v_mfma_f32_32x32x16_bf16"
#define AP_F16 "v_mfma_f32_32x32x16_f16"
template <typename T> __device__ __forceinline__ void ap_sacc0(f32x16& d, ap_i32x4 a, ap_i32x4 b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile(AP_BF16 " %0, %1, %2, 0" : "=&v"(d) : "a"(a), "a"(b));
    else asm volatile(AP_F16 " %0, %1, %2, 0" : "=&v"(d) : "a"(a), "a"(b));
}
template <typename T> __device__ __forceinline__ void ap_sacc(f32x16& d, ap_i32x4 a, ap_i32x4 b) {
    if constexpr (std::is_same<T, __bf16>::value) asm volatile(AP_BF16 " %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
    else asm volatile(AP_F16 " %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
}
// ---- softmax pieces.  Piece k: exp2 of element a, the pack of piece k - 2's results (px, py -> one dword of P), exp2 of element b.  A VALU read of a
// fresh transcendental result needs a wait state; hipcc cannot see the order inside an asm block and pads every block whose inputs were written by
// the block right in front of it with s_nop 0, so the pack trails by TWO pieces and the pieces rotate through three temporary pairs.  As separate C++ statements the exponentials drifted out of the gap they were written in.
template <typename T> __device__ __forceinline__ int ap_piece(float a, float b, float& x, float& y, float px, float py) {
    int r;
    if constexpr (std::is_same<T, __bf16>::value) asm("v_exp_f32 %1, %3\n\tv_cvt_pk_bf16_f32 %0, %5, %6\n\tv_exp_f32 %2, %4" : "=&v"(r), "=&v"(x), "=&v"(y) : "v"(a), "v"(b), "v"(px), "v"(py));
    else asm("v_exp_f32 %1, %3\n\tv_cvt_pk_f16_f32 %0, %5, %6\n\tv_exp_f32 %2, %4" : "=&v"(r), "=&v"(x), "=&v"(y) : "v"(a), "v"(b), "v"(px), "v"(py));
    return r;
}
__device__ __forceinline__ void ap_piece0(float a, float b, float& x, float& y) {
    asm("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(x), "=&v"(y) : "v"(a), "v"(b));
}
template <typename T> __device__ __forceinline__ int ap_pack(float x, float y) {
    int r;
    if constexpr (std::is_same<T, __bf16>::value) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    else asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// maximum of eight accumulator registers (asm: fmaxf on asm outputs would first canonicalise every input)
__device__ __forceinline__ float ap_max8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    float m;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8" : "=&v"(m) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
    return m;
}
__device__ __forceinline__ float ap_max4(float a, float b, float c, float d) {
    float m; asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(m) : "v"(a), "v"(b), "v"(c), "v"(d)); return m;
}
__device__ __forceinline__ float ap_max2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int... I, typename F> __device__ __forceinline__ void ap_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int LO, int... I> constexpr auto ap_range_impl(std::integer_sequence<int, I...>) { return std::integer_sequence<int, (LO + I)...>{}; }
template <int LO, int HI> constexpr auto ap_range() { return ap_range_impl<LO>(std::make_integer_sequence<int, HI - LO>{}); }
template <int N> using ap_ic = std::integral_constant<int, N>;

// O *= al for eight registers of an accumulator tile, as asm on the AGPRs themselves: written as o = o * al the rare path gives hipcc a VGPR use of O,
// and it then carries parts of O in VGPRs around the loop (24 v_accvgpr copies each way per block on the COMMON path)
template <int R0> __device__ __forceinline__ void ap_scale_acc8(f32x16& t, float al) {
    float tmp;
    asm volatile("v_accvgpr_read_b32 %8, %0\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %0, %8\n\t"
                 "v_accvgpr_read_b32 %8, %1\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %1, %8\n\t"
                 "v_accvgpr_read_b32 %8, %2\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %2, %8\n\t"
                 "v_accvgpr_read_b32 %8, %3\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %3, %8\n\t"
                 "v_accvgpr_read_b32 %8, %4\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %4, %8\n\t"
                 "v_accvgpr_read_b32 %8, %5\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %5, %8\n\t"
                 "v_accvgpr_read_b32 %8, %6\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %6, %8\n\t"
                 "v_accvgpr_read_b32 %8, %7\n\tv_mul_f32 %8, %8, %9\n\tv_accvgpr_write_b32 %7, %8\n\ts_nop 4"
                 : "+a"(t[R0]), "+a"(t[R0 + 1]), "+a"(t[R0 + 2]), "+a"(t[R0 + 3]), "+a"(t[R0 + 4]), "+a"(t[R0 + 5]), "+a"(t[R0 + 6]), "+a"(t[R0 + 7]), "=&v"(tmp) : "v"(al));
}

}  // namespace ldx
