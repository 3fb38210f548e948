// Which E8M0 scale does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which operand bytes?
// For every (lane group g = lane>>4, VGPR r of the 8 A registers): A = 1.0 (e4m3 0x38) in that register of the 16 lanes of
// group g only, B = 1.0 everywhere, scale_b = 2^0, scale_a of lane group g' = 2^(g'+1).  D[0][0] = 4 bytes * scale applied.
// Result on MI355X: data in lane group g, registers 0-3 takes the scale of lane group g>>1, registers 4-7 that of lane group
// 2 + (g>>1): registers 0-3 hold k = 16g..16g+15, registers 4-7 hold k = 64+16g.., and lane group b's scale covers k = 32b..32b+31.
// op_sel picks the byte of the scale register.  (Confirmed end to end by tests/test_mx_gpu.py.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(float* out, int g_sel, int r_sel, int opsel_mode, float c0) {
    const int lane = threadIdx.x, g = lane >> 4;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
    for (int r = 0; r < 8; ++r) b[r] = 0x38383838;
    if (g == g_sel) a[r_sel] = 0x38383838;
    const int sa = 128 + g;                 // 2^(g+1)
    const int sa_packed = (130 << 24) | (129 << 16) | (128 << 8) | 127;   // bytes 0..3 = 2^0, 2^1, 2^2, 2^3
    f32x4 c = {c0, c0, c0, c0};     // a run-time accumulator: with a literal 0 hipcc lets the result registers overlap A / B
    if (opsel_mode == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, 127);
    else if (opsel_mode == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa_packed, 0, 127);
    else if (opsel_mode == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, sa_packed, 0, 127);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, sa_packed, 0, 127);
    if (lane == 0) out[0] = c[0];
}

int main() {
    float* d; hipMalloc(&d, 4); float h;
    for (int mode = 0; mode < 4; ++mode) {
        printf("opsel_mode %d (0: per-lane scale 2^(g+1) in byte 0, opsel 0; 1..3: packed bytes {1,2,4,8}, opsel = mode)\n", mode);
        for (int g = 0; g < 4; ++g) {
            printf("  data in lane group %d:", g);
            for (int r = 0; r < 8; ++r) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, g, r, mode, 0.f);
                hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
                printf(" r%d->x%g", r, h / 4.f);
            }
            printf("\n");
        }
    }
    return 0;
}
