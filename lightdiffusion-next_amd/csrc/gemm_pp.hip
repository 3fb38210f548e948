// 256-row ping-pong GEMM / implicit-GEMM kernels (gemm_pp.inc) and their launchers, 16-bit operands (MX fp8: gemm_pp_mx.hip): translation
// units of their own so that they compile in parallel with gemm.hip (the three are most of the build time).
#include "gemm_common.h"

namespace ldx {

#include "gemm_pp.inc"

void launch_gemm_pp_mx(const GemmArgs& a, int bn, int S, DType dt, hipStream_t s);                       // gemm_pp_mx.hip
void launch_gemm_pp2_mx(const GemmArgs& a, const GemmArgs& b, int bn, DType dt, hipStream_t s);

template <typename T>
static void launch_pp_t(const GemmArgs& a, int bn, bool lnf, int S, hipStream_t s) {
    if (a.mode == 0 && a.f8) {            // MX fp8 operands: gemm_pp_mx.hip
        launch_gemm_pp_mx(a, bn, S, DTypeOf<T>::v, s);
    } else if (a.mode == 0 && lnf) {      // folded LayerNorm (GemmArgs::ln_c1), no split-K
        if (bn == 160 && !a.geglu) launch_pp_inst<T, 0, 160, true>(a, 1, s); else launch_pp_inst<T, 0, 128, true>(a, 1, s);
    } else if (a.mode == 0) {
        if (bn == 256) launch_pp_inst<T, 0, 256>(a, S, s);             // also GEGLU (two 64-column slabs per wave)
        else if (bn == 224 && !a.geglu) launch_pp_inst<T, 0, 224>(a, S, s);
        else if (bn == 192 && !a.geglu) launch_pp_inst<T, 0, 192>(a, S, s);
        else if (bn == 160 && !a.geglu) launch_pp_inst<T, 0, 160>(a, S, s);
        else launch_pp_inst<T, 0, 128>(a, S, s);
    } else {
        if (bn == 256) launch_pp_inst<T, 1, 256>(a, S, s);
        else if (bn == 160) launch_pp_inst<T, 1, 160>(a, S, s);
        else launch_pp_inst<T, 1, 128>(a, S, s);
    }
}
void launch_gemm_pp(const GemmArgs& a, int bn, bool lnf, int S, DType dt, hipStream_t s) {
    gemm_gn_tile_check(a, 256, (lnf || (a.geglu && bn != 256)) && bn != 160 ? 128 : bn, S);
    if (dt == DT_BF16) launch_pp_t<__bf16>(a, bn, lnf, S, s); else launch_pp_t<_Float16>(a, bn, lnf, S, s);
}

template <typename T>
static void launch_pp2_t(const GemmArgs& a, const GemmArgs& b, int bn, hipStream_t s) {
    if (a.f8) launch_gemm_pp2_mx(a, b, bn, DTypeOf<T>::v, s);
    else if (bn == 256) launch_pp2_inst<T, 256>(a, b, s);
    else if (bn == 224) launch_pp2_inst<T, 224>(a, b, s);
    else if (bn == 192) launch_pp2_inst<T, 192>(a, b, s);
    else if (bn == 160) launch_pp2_inst<T, 160>(a, b, s);
    else launch_pp2_inst<T, 128>(a, b, s);
}
void launch_gemm_pp2(const GemmArgs& a, const GemmArgs& b, int bn, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_pp2_t<__bf16>(a, b, bn, s); else launch_pp2_t<_Float16>(a, b, bn, s);
}

}  // namespace ldx
