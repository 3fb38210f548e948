// Operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3), found by trying hypotheses against a host reference:
// random small e4m3 values (exact sums), random per-(row, 32-block) E8M0 scales, D = sum_k sa[i][k / 32] A[i][k] * sb[j][k / 32] B[k][j].
//   data hypotheses (lane l: row / column l & 31, half h = l >> 5; byte p = 0 .. 31 of the lane's 8 registers):
//     D1: k = 32 h + p                       D2: k = 16 h + p (p < 16), 32 + 16 h + (p - 16) (p >= 16)
//   scale hypotheses (which 32-block of its row does the scale a lane supplies apply to):  S1: block h     S2: block 1 - h
//   C / D layout assumed = v_mfma_f32_32x32x16_bf16's: lane l holds D[8 (r >> 2) + 4 h + (r & 3)][l & 31], r = 0 .. 15.
// Prints the maximum |error| of each (D, S) combination: exactly one should be 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void run(const unsigned char* A, const unsigned char* B, const int* SA, const int* SB, float* D) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = ((const int*)A)[lane * 8 + r]; b[r] = ((const int*)B)[lane * 8 + r]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = D[lane * 16 + r];        // run-time accumulator (zeros)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, SA[lane], 0, SB[lane]);
    for (int r = 0; r < 16; ++r) D[lane * 16 + r] = c[r];
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

int main() {
    const unsigned char vals[8] = {0x00, 0x30, 0x38, 0x3c, 0x40, 0xb8, 0xb0, 0x44};      // 0, .5, 1, 1.5, 2, -1, -.5, 3
    unsigned char hA[64 * 32], hB[64 * 32]; int hSA[64], hSB[64]; float hD[64 * 16];
    srand(7);
    for (int i = 0; i < 64 * 32; ++i) { hA[i] = vals[rand() & 7]; hB[i] = vals[rand() & 7]; }
    for (int l = 0; l < 64; ++l) { hSA[l] = 125 + (rand() % 5); hSB[l] = 125 + (rand() % 5); }      // 2^-2 .. 2^2
    unsigned char *dA, *dB; int *dSA, *dSB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dSA, sizeof(hSA)); hipMalloc(&dSB, sizeof(hSB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dSA, hSA, sizeof(hSA), hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, sizeof(hSB), hipMemcpyHostToDevice);
    hipMemset(dD, 0, sizeof(hD));
    hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    for (int dh = 1; dh <= 2; ++dh)
        for (int sh = 1; sh <= 2; ++sh) {
            double worst = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int i = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3), j = l & 31;
                    double acc = 0;
                    for (int h = 0; h < 2; ++h)
                        for (int p = 0; p < 32; ++p) {
                            const int k = dh == 1 ? 32 * h + p : (p < 16 ? 16 * h + p : 32 + 16 * h + (p - 16));
                            const int blk = k >> 5;
                            // the lane that SUPPLIES the scale of block blk of row i: half (sh == 1 ? blk : 1 - blk)
                            const int hs = sh == 1 ? blk : 1 - blk;
                            const double sa = ldexp(1.0, hSA[hs * 32 + i] - 127), sb = ldexp(1.0, hSB[hs * 32 + j] - 127);
                            acc += sa * e4m3(hA[(h * 32 + i) * 32 + p]) * sb * e4m3(hB[(h * 32 + j) * 32 + p]);
                        }
                    worst = fmax(worst, fabs(acc - hD[l * 16 + r]));
                }
            printf("data hypothesis D%d, scale hypothesis S%d: max |err| = %g\n", dh, sh, worst);
        }
    printf("D[0][0..3] on the GPU: %g %g %g %g\n", hD[0], hD[16], hD[32], hD[48]);
    return 0;
}
