// 256-row ping-pong GEMM / implicit-GEMM kernels (gemm_pp.inc), 16-bit operands: the dispatchers.  The kernels themselves are instantiated in six translation
// units (gemm_pp_{bf16,f16}_{0,1,2}.hip via gemm_pp_parts.inc) and the MX fp8 ones in gemm_pp_mx.hip, so that they compile in parallel with gemm.hip.
#include "gemm_common.h"

namespace ldx {

void launch_gemm_pp_mx(const GemmArgs& a, int bn, int S, DType dt, hipStream_t s);                       // gemm_pp_mx.hip
void launch_gemm_pp2_mx(const GemmArgs& a, const GemmArgs& b, int bn, DType dt, hipStream_t s);
#define LDX_PP_DECL(sfx) \
    void launch_pp_plain_##sfx(const GemmArgs& a, int bn, bool lnf, int S, hipStream_t s); \
    void launch_pp_conv_##sfx(const GemmArgs& a, int bn, int S, hipStream_t s); \
    void launch_pp2_narrow_##sfx(const GemmArgs& a, const GemmArgs& b, int bn, hipStream_t s); \
    void launch_pp2_wide_##sfx(const GemmArgs& a, const GemmArgs& b, int bn, hipStream_t s);
LDX_PP_DECL(bf16)
LDX_PP_DECL(f16)
#undef LDX_PP_DECL

void launch_gemm_pp(const GemmArgs& a, int bn, bool lnf, int S, DType dt, hipStream_t s) {
    gemm_gn_tile_check(a, 256, (lnf || (a.geglu && bn != 256)) && bn != 160 ? 128 : bn, S);
    if (a.mode == 0 && a.f8) { launch_gemm_pp_mx(a, bn, S, dt, s); return; }      // MX fp8 operands
    if (a.mode == 0) { if (dt == DT_BF16) launch_pp_plain_bf16(a, bn, lnf, S, s); else launch_pp_plain_f16(a, bn, lnf, S, s); }
    else { if (dt == DT_BF16) launch_pp_conv_bf16(a, bn, S, s); else launch_pp_conv_f16(a, bn, S, s); }
}

void launch_gemm_pp2(const GemmArgs& a, const GemmArgs& b, int bn, DType dt, hipStream_t s) {
    if (a.f8) { launch_gemm_pp2_mx(a, b, bn, dt, s); return; }
    const bool wide = bn == 256 || bn == 224 || bn == 192;
    if (dt == DT_BF16) { if (wide) launch_pp2_wide_bf16(a, b, bn, s); else launch_pp2_narrow_bf16(a, b, bn, s); }
    else { if (wide) launch_pp2_wide_f16(a, b, bn, s); else launch_pp2_narrow_f16(a, b, bn, s); }
}

}  // namespace ldx
