// see gemm_pp_parts.inc
#define LDX_PP_T __bf16
#define LDX_PP_SFX bf16
#define LDX_PP_PART 2
#include "gemm_pp_parts.inc"
