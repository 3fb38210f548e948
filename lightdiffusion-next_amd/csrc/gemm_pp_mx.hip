// MX fp8 instantiations of the 256-row ping-pong GEMM (gemm_pp.inc): tile widths 128 / 160 / 192 / 224, single and two-problem launches.
#include "gemm_common.h"

namespace ldx {

#include "gemm_pp.inc"

template <typename T>
static void launch_pp_mx_t(const GemmArgs& a, int bn, int S, hipStream_t s) {
    // the scales ride in the W slot's tail (BN <= 192, or 224 with the scale DMA in the fourth W round); quantised output (C8) needs BN / 2 to
    // be a multiple of 32: 128 or 192
    if (bn == 224 && !a.C8) launch_pp_inst<T, 0, 224, false, true>(a, S, s);
    else if (bn == 192) launch_pp_inst<T, 0, 192, false, true>(a, S, s);
    else if (bn == 160 && !a.C8) launch_pp_inst<T, 0, 160, false, true>(a, S, s);
    else launch_pp_inst<T, 0, 128, false, true>(a, S, s);
}
void launch_gemm_pp_mx(const GemmArgs& a, int bn, int S, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_pp_mx_t<__bf16>(a, bn, S, s); else launch_pp_mx_t<_Float16>(a, bn, S, s);
}

template <typename T>
static void launch_pp2_mx_t(const GemmArgs& a, const GemmArgs& b, int bn, hipStream_t s) {
    const bool c8 = a.C8 || b.C8;
    if (bn == 224 && !c8) launch_pp2_inst<T, 224, true>(a, b, s);
    else if (bn == 192) launch_pp2_inst<T, 192, true>(a, b, s);
    else if (bn == 160 && !c8) launch_pp2_inst<T, 160, true>(a, b, s);
    else launch_pp2_inst<T, 128, true>(a, b, s);
}
void launch_gemm_pp2_mx(const GemmArgs& a, const GemmArgs& b, int bn, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_pp2_mx_t<__bf16>(a, b, bn, s); else launch_pp2_mx_t<_Float16>(a, b, bn, s);
}

}  // namespace ldx
