// 256-row ping-pong GEMM / implicit-GEMM kernels (gemm_pp.inc) and their launchers: a translation unit of its own so that it compiles
// in parallel with gemm.hip (together they were 2.6 of the build's 2.7 minutes).
#include "gemm_common.h"

namespace ldx {

#include "gemm_pp.inc"

template <typename T, int MODE, int BN, bool LNF = false, bool F8 = false>
static void launch_pp_inst(const GemmArgs& a, int S, hipStream_t s) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + BN - 1) / BN) * S;
    static DevOnce once;
    set_dyn_lds(once, (const void*)gemm_pp_kernel<T, MODE, BN, LNF, F8>, PP_LDS);
    hipLaunchKernelGGL((gemm_pp_kernel<T, MODE, BN, LNF, F8>), dim3(tiles), dim3(512), PP_LDS, s, a);
}
template <typename T>
static void launch_pp_t(const GemmArgs& a, int bn, bool lnf, int S, hipStream_t s) {
    if (a.mode == 0 && a.f8) {            // MX fp8 operands: BN <= 160 (the scales ride in the W slot's tail)
        if (bn == 160 && !a.C8) launch_pp_inst<T, 0, 160, false, true>(a, S, s); else launch_pp_inst<T, 0, 128, false, true>(a, S, s);
    } else if (a.mode == 0 && lnf) {      // folded LayerNorm (GemmArgs::ln_c1), no split-K
        if (bn == 160 && !a.geglu) launch_pp_inst<T, 0, 160, true>(a, 1, s); else launch_pp_inst<T, 0, 128, true>(a, 1, s);
    } else if (a.mode == 0) {
        if (bn == 256 && !a.geglu) launch_pp_inst<T, 0, 256>(a, S, s);
        else if (bn == 160 && !a.geglu) launch_pp_inst<T, 0, 160>(a, S, s);
        else launch_pp_inst<T, 0, 128>(a, S, s);
    } else {
        if (bn == 256) launch_pp_inst<T, 1, 256>(a, S, s);
        else if (bn == 160) launch_pp_inst<T, 1, 160>(a, S, s);
        else launch_pp_inst<T, 1, 128>(a, S, s);
    }
}
void launch_gemm_pp(const GemmArgs& a, int bn, bool lnf, int S, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_pp_t<__bf16>(a, bn, lnf, S, s); else launch_pp_t<_Float16>(a, bn, lnf, S, s);
}

template <typename T, int BN, bool F8 = false>
static void launch_pp2_inst(const GemmArgs& a, const GemmArgs& b, hipStream_t s) {
    const int ta = ((a.M + 255) / 256) * ((a.N + BN - 1) / BN), tb = ((b.M + 255) / 256) * ((b.N + BN - 1) / BN);
    static DevOnce once;
    set_dyn_lds(once, (const void*)gemm_pp2_kernel<T, BN, F8>, PP_LDS);
    hipLaunchKernelGGL((gemm_pp2_kernel<T, BN, F8>), dim3(ta + tb), dim3(512), PP_LDS, s, a, b, ta);
}
template <typename T>
static void launch_pp2_t(const GemmArgs& a, const GemmArgs& b, int bn, hipStream_t s) {
    if (a.f8) { if (bn == 160) launch_pp2_inst<T, 160, true>(a, b, s); else launch_pp2_inst<T, 128, true>(a, b, s); }
    else if (bn == 256) launch_pp2_inst<T, 256>(a, b, s);
    else if (bn == 160) launch_pp2_inst<T, 160>(a, b, s);
    else launch_pp2_inst<T, 128>(a, b, s);
}
void launch_gemm_pp2(const GemmArgs& a, const GemmArgs& b, int bn, DType dt, hipStream_t s) {
    if (dt == DT_BF16) launch_pp2_t<__bf16>(a, b, bn, s); else launch_pp2_t<_Float16>(a, b, bn, s);
}

}  // namespace ldx
