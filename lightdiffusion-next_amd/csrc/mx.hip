// MX fp8 quantisation (OCP microscaling: 32-element blocks, E8M0 shared scale, e4m3fn elements) for the block-scaled
// MFMA path of gemm.hip (GemmArgs::f8).  BASELINE config 4 names "fp8 MFMA" for the Flux DiT; the reference itself runs
// Flux from Q8_0 weights (32-element blocks with one scale each, dequantised to 16-bit before every Linear), so this is an
// opt-in approximate mode of its own parity class, like the first-block cache.
//
// Scale rule (shared by every producer of MX operands — this kernel, and the fused epilogues):
//   r = amax * (1/448) in fp32;  e = biased exponent of r, + 1 if its mantissa is non-zero (ceil log2), clamped to [1, 253];
//   scale = 2^(e - 127) (the E8M0 byte is e);  y = e4m3fn_rne(x * 2^(127 - e)).  No element clips: |x / scale| <= 448.
#include "ldx_device.h"
#include "ldx_kernels.h"

namespace ldx {

// one thread per 32-element block: 4 x 16-B loads, 2 x 16-B stores, one scale byte
template <typename T>
__global__ __launch_bounds__(256) void mx_quant_kernel(const MxQuantArgs p) {
    const int nkb = p.K >> 5;
    const long total = (long)p.rows * nkb;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / nkb), kb = (int)(idx - (long)row * nkb);
        const T* src = (const T*)p.X + (long)row * p.ldx + kb * 32;
        float f[32];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t[8];
            unpack8<T>(*(const uint4*)(src + c * 8), t);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[c * 8 + i] = t[i];
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(f[i]));
        const int e = mx_scale_e8m0(amax);
        const float inv = mx_inv_scale(e);
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = mx_pack4(f[4 * i] * inv, f[4 * i + 1] * inv, f[4 * i + 2] * inv, f[4 * i + 3] * inv);
        uint4* dst = (uint4*)((char*)p.Y + (long)row * p.ldy + kb * 32);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        ((uint8_t*)p.S)[((long)(kb >> 2) * p.s_ld + row) * 4 + (kb & 3)] = (uint8_t)e;
    }
}

void launch_mx_quant(const MxQuantArgs& a, DType dt, hipStream_t s) {
    if (a.rows <= 0 || a.K <= 0) return;
    const long total = (long)a.rows * (a.K >> 5);
    long grid = (total + 255) / 256; if (grid > 65536) grid = 65536;
    if (dt == DT_BF16) hipLaunchKernelGGL((mx_quant_kernel<__bf16>), dim3((int)grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mx_quant_kernel<_Float16>), dim3((int)grid), dim3(256), 0, s, a);
}

}  // namespace ldx
