// Patch-resident 3x3 convolution for narrow outputs (round 5): N = 32 / 64 output channels, stride 1, pad 1, optional nearest-resize gather.
//
// Why: ESRGAN's RRDBNet (UltimateSDUpscale/RDRB.py:80-205) is 345 convs of 64..192 -> 32 / 64 channels over a 512^2 tile.  As an implicit GEMM
// with 128 x 32 tiles every K-tile re-fetches the A rows of ONE tap: 9 x the input through the L2 -> LDS path, 32 flop per operand byte, and the
// launch sits at the ~24 B/clk/CU that path delivers (375 TFLOP/s, 26.9 ms per tile; DESIGN r4 item 6).  Holding a whole 5-pixel-halo RDB in LDS
// does not fit — but the 9 taps of ONE conv share one input patch, and that does:
//
//   workgroup = 16 x 16 output pixels of one image, 4 waves (wave w owns image rows 4w .. 4w+3 of the tile), two workgroups per CU;
//   the input is streamed in 32-channel chunks: chunk kc of the 18 x 18 patch (324 pixels x 64 B = 20.25 KiB, LDS-DMA, double buffered)
//   serves all 9 taps; the weights of (chunk, kx) — 3 ky x N x 32 channels = 3 N / 16 KiB — ride a three-slot LDS ring one (kc, kx) step ahead;
//   per step a wave reads 6 patch-row fragments (rows 4w .. 4w+5 at column offset kx: each serves up to 3 (row, ky) pairs) and 3 N / 16 weight
//   fragments, and issues 12 N / 16 MFMA 16x16x32: 1.27 x the input + the weights once per tile instead of 9 x the input —
//   78 (N = 32) / 114 (N = 64) KB per CU per chunk pair against 2304 / 4608 MFMA cycles.
//
// LDS layouts are chosen through the DMA's per-lane SOURCE address (the destination of a piece is always 64 consecutive 16-B slots):
//   patch slot (P = 18 Y + X, pos) holds channel octet  pos ^ 2 ((X >> 2) & 1)  — the 16 lanes of a ds_read_b128 group (pixels X = l15 + kx, octet
//   g4) then cover 16 distinct bank quads for kx = 0, 1, 2 (brute-forced); a weight piece (ky, j) holds octet g4 of row 16 j + n at slot 16 g4 + n.
// One s_barrier per step; DMA completion by counted vmcnt (pieces retire in order; the counts per wave are wave-uniform).
// Epilogue: bias, LeakyReLU(0.2), output scale, residual, second scaled residual (RDRB.py:76, 205), 16-bit and / or fp32 stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"

namespace ldx {

typedef __attribute__((ext_vector_type(4))) int cp_i32x4;
static __device__ __forceinline__ cp_i32x4 cp_srd(const void* base, long bytes) {
    const unsigned long long q = (unsigned long long)base;
    const int n = (int)(bytes > 0x7fffffffL ? 0x7fffffffL : (bytes > 0 ? bytes : 0));
    return (cp_i32x4){(int)(unsigned)q, (int)((unsigned)(q >> 32) & 0xffffu), n, 0x00020000};
}
// M0 is written without being declared (gemm_pp.inc explains why that is safe in these kernels)
static __device__ __forceinline__ void cp_dma16(const cp_i32x4 rsrc, int voff, int soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> static __device__ __forceinline__ void cp_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
static __device__ __forceinline__ void cp_wait_n(int n) {      // n is wave-uniform
    switch (n) {
        case 0: cp_wait<0>(); break;  case 1: cp_wait<1>(); break;  case 2: cp_wait<2>(); break;  case 3: cp_wait<3>(); break;
        case 4: cp_wait<4>(); break;  case 5: cp_wait<5>(); break;  case 6: cp_wait<6>(); break;  case 7: cp_wait<7>(); break;
        case 8: cp_wait<8>(); break;  default: cp_wait<9>(); break;
    }
}

constexpr int CP_TH = 16, CP_TW = 16, CP_PW = CP_TW + 2, CP_PIX = (CP_TH + 2) * CP_PW;     // 324 patch pixels
constexpr int CP_PP = (CP_PIX * 4 + 63) / 64;                                             // 21 pieces of 64 slots per channel chunk
constexpr int CP_PATCH = CP_PP * 1024;
constexpr int cp_lds_bytes(int NJ) { return 2 * CP_PATCH + 3 * (3 * NJ) * 1024; }

template <typename T, int NJ>
__global__ __launch_bounds__(256, 2) void conv_patch_kernel(const GemmArgs p) {
    constexpr int OOB = (int)0x80000000;
    constexpr int WP = 3 * NJ, WST = WP * 1024, NWP = (WP + 3) / 4, NPP = (CP_PP + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V8 = typename Vec<T>::v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int tiles_x = p.Wout / CP_TW, tiles_y = p.Hout / CP_TH, per_img = tiles_x * tiles_y;
    const int nimg = p.M / (p.Hout * p.Wout);
    const int lin = xcd_remap(blockIdx.x, nimg * per_img);
    const int b = lin / per_img, rem = lin - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = ty * CP_TH, x0 = tx * CP_TW;
    const int Cin = p.Cin, nkc = Cin >> 5, nsteps = 3 * nkc;

    const cp_i32x4 rA = cp_srd(p.A, (((long)nimg * p.Hin * p.Win - 1) * p.lda + Cin) * 2);
    const cp_i32x4 rW = cp_srd(p.W, (long)p.N * p.K * 2);
    const float rs_y = p.resize ? (float)p.Hin / (float)p.Hv : 1.f, rs_x = p.resize ? (float)p.Win / (float)p.Wv : 1.f;

    // this wave's patch pieces: piece = wave + 4 i; slot q = 64 piece + lane -> pixel P = q >> 2 = 18 Y + X, position q & 3
    int pv[NPP];
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int q = (wave + 4 * i) * 64 + lane, P = q >> 2;
        const int Y = P / CP_PW, X = P - Y * CP_PW;
        const int oct = (q & 3) ^ (((X >> 2) & 1) << 1);
        const int vy = y0 - 1 + Y, vx = x0 - 1 + X;
        int sy = vy, sx = vx;
        if (p.resize) {      // nearest: src = min(floor(dst * in / out), in - 1)  (torch upsample_nearest; as gemm.hip)
            sy = min((int)floorf((float)vy * rs_y), p.Hin - 1);
            sx = min((int)floorf((float)vx * rs_x), p.Win - 1);
        }
        const bool in = P < CP_PIX && vy >= 0 && vy < p.Hv && vx >= 0 && vx < p.Wv;
        pv[i] = in ? (int)((((long)b * p.Hin + sy) * p.Win + sx) * p.lda * 2 + oct * 16) : OOB;
    }
    // this wave's weight pieces of a (chunk, kx) stage: piece pw = wave + 4 i = ky * NJ + j; lane -> row 16 j + (lane & 15), octet lane >> 4
    int wv[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int pw = wave + 4 * i, ky = pw / NJ, j = pw - ky * NJ;
        wv[i] = pw < WP ? ((j * 16 + l15) * p.K + ky * 3 * Cin + g4 * 8) * 2 : OOB;
    }
    const int nP = (CP_PP - wave + 3) / 4, nW = (WP - wave + 3) / 4;      // pieces this wave issues per patch chunk / weight stage
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_w = lds_base + 2 * CP_PATCH;
    auto issue_patch = [&](int kc) __attribute__((always_inline)) {
        const unsigned dst = lds_base + (kc & 1) * CP_PATCH;
#pragma unroll
        for (int i = 0; i < NPP; ++i)
            if (wave + 4 * i < CP_PP) cp_dma16(rA, pv[i], kc * 64, dst + (wave + 4 * i) * 1024);
    };
    auto issue_w = [&](int s) __attribute__((always_inline)) {      // step s = 3 kc + kx
        const int kc = s / 3, kx = s - 3 * kc;
        const unsigned dst = lds_w + (s % 3) * WST;
        const int soff = (kx * Cin + kc * 32) * 2;
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            if (wave + 4 * i < WP) cp_dma16(rW, wv[i], soff, dst + (wave + 4 * i) * 1024);
    };

    f32x4 acc[4][NJ];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_patch(0); issue_w(0);
    if (nsteps > 1) issue_w(1);

    // fragment offsets: patch pixel (Y, X = l15 + kx), octet g4 at slot position g4 ^ 2 ((X >> 2) & 1); weights at piece * 1024 + (16 g4 + l15) * 16
    int a_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int X = l15 + kx;
        a_off[kx] = ((4 * wave) * CP_PW + X) * 64 + ((g4 ^ (((X >> 2) & 1) << 1)) << 4);
    }
    const int w_off = (g4 * 16 + l15) * 16;

    int s = 0, wslot = 0;
    for (int kc = 0; kc < nkc; ++kc) {
        const char* pa = smem + (kc & 1) * CP_PATCH;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx, ++s) {
            // pieces issued after the ones this step reads may stay in flight (see the schedule below)
            const int allowed = (s + 1 < nsteps) ? nW + ((kx != 0 && kc + 1 < nkc) ? nP : 0) : 0;
            cp_wait_n(allowed);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // every wave is past step s - 1: its weight slot and (at kx = 0) the patch slot of chunk kc - 1 are free
            if (s + 2 < nsteps) issue_w(s + 2);
            if (kx == 0 && kc + 1 < nkc) issue_patch(kc + 1);
            const char* pw = smem + 2 * CP_PATCH + wslot * WST + w_off;
            V8 af[6], wf[3][NJ];
#pragma unroll
            for (int i = 0; i < 6; ++i) af[i] = as_v8<T>(*(const uint4*)(pa + a_off[kx] + i * (CP_PW * 64)));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[ky][j] = as_v8<T>(*(const uint4*)(pw + (ky * NJ + j) * 1024));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[r][j] = mfma16(wf[ky][j], af[r + ky], acc[r][j]);
            wslot = wslot == 2 ? 0 : wslot + 1;
        }
    }

    // output stage: lane (l15, g4) holds pixel x0 + l15 of row y0 + 4 wave + r, channels 16 j + 4 g4 .. + 3
    T* Cp = (T*)p.C;
    const T* Rp = (const T*)p.R;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long m = ((long)b * p.Hout + y0 + 4 * wave + r) * p.Wout + x0 + l15;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = j * 16 + 4 * g4;
            float v[4] = {acc[r][j][0], acc[r][j][1], acc[r][j][2], acc[r][j][3]};
            if (p.bias) { const float4 bv = *(const float4*)(p.bias + n); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
            if (p.act == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.2f * v[q];
            }
            if (p.oscale != 0.f) { v[0] *= p.oscale; v[1] *= p.oscale; v[2] *= p.oscale; v[3] *= p.oscale; }
            if (Rp) { float rr[4]; unpack4<T>(*(const uint2*)(Rp + m * p.ldr + n), rr); v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3]; }
            if (p.R2) { float rr[4]; unpack4<T>(*(const uint2*)((const T*)p.R2 + m * p.ldr2 + n), rr);
                        v[0] = fmaf(v[0], p.oscale2, rr[0]); v[1] = fmaf(v[1], p.oscale2, rr[1]); v[2] = fmaf(v[2], p.oscale2, rr[2]); v[3] = fmaf(v[3], p.oscale2, rr[3]); }
            if (Cp) *(uint2*)(Cp + m * p.ldc + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            if (p.Cf) *(float4*)(p.Cf + m * p.ldcf + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// DMA schedule (per wave; pieces retire in issue order): prologue patch(0), W(0), W(1); step s = (kc, kx) issues W(s + 2), then at kx = 0 patch(kc + 1).
// Step s reads W(s) (issued during step s - 2) and patch(kc) (issued during step (kc - 1, 0)); issued after W(s): the patch pieces of step s - 2 if that
// was a kx = 0 step, and everything of step s - 1 — nW at kx = 0, nW + nP at kx = 1 and 2 (fewer near the end: the kernel then waits for everything).

bool conv_patch_ok(const GemmArgs& a) {
    static const bool off = getenv("LDX_CONV_PATCH") && atoi(getenv("LDX_CONV_PATCH")) == 0;
    if (off || a.mode != 1 || a.stride != 1 || a.A2 || a.pad0 || a.f8 || a.C8 || a.ln_c1 || a.geglu || a.rowvec || a.gate || a.gn_partial || a.splitk > 1) return false;
    if (a.N != 32 && a.N != 64) return false;
    if (a.act != 0 && a.act != 3) return false;
    if (a.Cin % 32 || a.K != 9 * a.Cin || a.Hout % CP_TH || a.Wout % CP_TW || a.Hv != a.Hout || a.Wv != a.Wout) return false;
    if (a.lda % 8 || (a.C && a.ldc % 4) || (a.R && a.ldr % 4) || (a.R2 && a.ldr2 % 4) || (a.Cf && a.ldcf % 4)) return false;
    const long nimg = a.M / ((long)a.Hout * a.Wout);
    if (nimg * a.Hin * a.Win * a.lda * 2 >= 0x7fffffffL) return false;      // 32-bit buffer offsets
    return nimg * (a.Hout / CP_TH) * (a.Wout / CP_TW) >= 256;               // at least one workgroup per CU
}

template <typename T, int NJ>
static void launch_conv_patch_t(const GemmArgs& a, hipStream_t s) {
    static DevOnce once;
    constexpr int lds = cp_lds_bytes(NJ);
    set_dyn_lds(once, (const void*)conv_patch_kernel<T, NJ>, lds);
    const unsigned tiles = (unsigned)((a.M / (a.Hout * a.Wout)) * (a.Hout / CP_TH) * (a.Wout / CP_TW));
    hipLaunchKernelGGL((conv_patch_kernel<T, NJ>), dim3(tiles), dim3(256), lds, s, a);
}

void launch_conv_patch(const GemmArgs& a, DType dt, hipStream_t s) {
    if (dt == DT_BF16) { if (a.N == 32) launch_conv_patch_t<__bf16, 2>(a, s); else launch_conv_patch_t<__bf16, 4>(a, s); }
    else               { if (a.N == 32) launch_conv_patch_t<_Float16, 2>(a, s); else launch_conv_patch_t<_Float16, 4>(a, s); }
}

}  // namespace ldx
