// see gemm_pp_parts.inc
#define LDX_PP_T _Float16
#define LDX_PP_SFX f16
#define LDX_PP_PART 2
#include "gemm_pp_parts.inc"
